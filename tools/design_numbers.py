"""DESIGN.md section 7's table of current numbers, generated from the committed files under profiles/ (VERDICT r04 "next" 9: one
table, every figure from a named file).   python tools/design_numbers.py [--write] [tag]      (--write: rewrite the block between the
numbers:begin / numbers:end markers of DESIGN.md)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r06"
    txt = open(os.path.join(P, f"{tag}_bench_full.json")).read().strip()
    full = json.loads(txt) if txt.startswith("{\n") or "\n" in txt else json.loads(txt.splitlines()[-1])     # round 6: the extras file (indented); before: one line
    stats = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv")))}
    pmc = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(P, f"{tag}_pmc_summary.csv")))}
    k2t = json.load(open(os.path.join(P, "k2_traffic.json")))
    k2ct = json.load(open(os.path.join(P, "k2c_traffic.json")))
    rows = []

    def row(what, value, src):
        rows.append(f"| {what} | {value} | `{src}` |")
    B = f"profiles/{tag}_bench_full.json"
    row("full chain, 2^29 samples per pass, one GPU (`value`, two passes in flight)", f"**{full['value'] / 1e3:.1f} Gsample/s**, {full['config']['ms_per_pass']:.4f} ms per pass", B)
    if full.get("one_pass_at_a_time"):
        row("… one pass at a time", f"{full['one_pass_at_a_time']['value'] / 1e3:.1f} Gsample/s, {full['one_pass_at_a_time']['ms_per_pass']:.4f} ms", B)
    sm = full["stage_ms"]
    row("stages (HIP events, ms): decimate + fix-up / fmDemod+resample + fix-up / filter + fix-up", f"{sm['decimate']:.4f} / {sm['resample']:.4f} / {sm['filter']:.4f}", B)
    rf = full["roofline"]
    row("K2 roofline (VALU, unfused, 78.6 TFLOP/s)", f"{rf['achieved']} TFLOP/s = **{rf['frac']:.3f}**; HBM view {rf['hbm']['achieved']:.0f} GB/s = {rf['hbm']['frac']:.3f}", B)
    row("K2 HBM traffic per launch (FETCH x 2 + WRITE) / algorithmic", f"{k2t['hbm_bytes_per_launch'] / 1e9:.4f} GB / {k2t['algorithmic_bytes_per_launch'] / 1e9:.4f} GB = {k2t['hbm_bytes_per_launch'] / k2t['algorithmic_bytes_per_launch']:.4f}", "profiles/k2_traffic.json")
    for k, label in (("k_decimate_systolic<true, 1, false>", "K2 u8 kernel alone"), ("k_decimate_systolic<true, 1, true, false>", "K2 u8 kernel alone"), ("k_decimate_systolic<true, 1, true, false, 16>", "K2 u8 kernel alone"), ("k_resample3_fast<3, 64, 4, 3, 3, 256, true, 8, true>", "fmDemod + resampler kernel alone"),
                     ("k_fir_real8_fast<true, 4, 256, 8>", "filter kernel alone"), ("k_decimate_c_crossfix<true, 8, 128, 16, 16, false>", "decimator seam fix-up"),
                     ("k_resample_real_crossfix<20, 135, 32>", "resampler seam fix-up"), ("k_resample3_stragglers<20, 135, 8>", "resampler seam fix-up + lead-in + tail (one launch)"), ("k_filter_real_crossfix_lds<128>", "filter seam fix-up")):
        if k in stats:
            extra = ""
            if k in pmc and pmc[k].get("SQ_INSTS_VALU") and pmc[k].get("GRBM_GUI_ACTIVE") and float(stats[k]['avg_ns']) > 5e4:   # (a counter pass of a 10 us kernel says little)
                try:
                    extra = f"; {float(pmc[k]['SQ_INSTS_VALU']) / 1e6:.0f} M VALU instructions, GRBM {float(pmc[k]['GRBM_GUI_ACTIVE']) / 8 / (float(stats[k]['avg_ns']) * 1e-9) / 1e9:.2f} GHz"
                except (ValueError, ZeroDivisionError):
                    extra = ""
            row(f"{label} (rocprofv3 kernel trace, avg of {stats[k]['calls']})", f"{float(stats[k]['avg_ns']) / 1e3:.1f} µs ({float(stats[k]['pct']):.1f} % of the pass){extra}", f"profiles/{tag}_kernel_stats.csv, {tag}_pmc_summary.csv")
    ex = full.get("example_taps_chain")
    if isinstance(ex, dict):
        row("the same pass with the reference example's OWN taps (51 / 31 / 64: `examples/fm/Coeffs.hs` as data)", f"{ex['value'] / 1e3:.1f} Gsample/s, {ex['ms_per_pass']:.4f} ms per pass "
            f"(stages {' / '.join(f'{k} {v:.3f}' for k, v in ex['stage_ms'].items())} ms: the 52-tap tile decimator, stand-alone fmDemod + 16-float-group resampler, 32-half-tap filter -- parity path, not tuned)", B)
    c1 = full["roofline_config1_cfloat_decimate"]
    row("BASELINE configs[1]: cfloat ÷8, 2^27 samples, 8192-sample seams", f"{c1['avg_launch_ms']:.4f} ms per launch: read-only **{c1['read_only_frac']:.3f}** of 8 TB/s ({c1['frac']:.3f} incl. writes); "
        f"{c1['ceilings_same_process']['kernel_over_nt_stream']:.3f} of the best no-arithmetic stream of that shape in the same process", B)
    row("… its HBM traffic / algorithmic", f"{k2ct['ratio']:.4f}", "profiles/k2c_traffic.json")
    sh = full.get("shard_1M_samples_per_gpu") or {}
    if sh:
        row("BASELINE configs[4] shard (2^20 samples per pass, one-kernel chain)", f"{sh['us_per_pass']} µs per pass = {sh['value'] / 1e3:.1f} Gsample/s; two passes in flight {sh.get('two_passes_in_flight', {}).get('us_per_pass', '-')} µs", B)
    h = full.get("host_streamed") or {}
    if h.get("link"):
        ln = h["link"]
        row("host link of the bench process: pinned H2D / D2H / kernel reading pinned host memory", f"{ln.get('pinned_h2d_GBps')} / {ln.get('pinned_d2h_GBps')} / {ln.get('kernel_reads_pinned_host_GBps')} GB/s", B)
        for name, d in (h.get("link_roofline") or {}).items():
            if "Msamples_per_s" in d:
                row(f"host-streamed `{name}`", f"{d['Msamples_per_s'] / 1e3:.2f} Gsample/s = {d['link_GBps']} GB/s on the link = **{d.get('frac')}** of its ceiling", B)
            else:       # BASELINE configs[3]: a Pipe of float elements
                row(f"host-streamed `{name}` (BASELINE configs[3])", f"{d['Melements_per_s'] / 1e3:.2f} G elements/s ({d.get('us_per_push')} µs per 65 536-float push) = {d['link_GBps']} GB/s on the link = **{d.get('frac')}** of its ceiling", B)
        oe = h.get("overlap_efficiency_4096_block_pushes")
        if isinstance(oe, dict):
            row("overlap efficiency of the double-buffered path (4096-block copying pushes)", f"{oe['value']} (copy {oe['copy_ms']} ms, compute {oe['compute_ms']} ms, wall {oe['wall_ms_per_push']} ms)", B)
    sw = full.get("launch_size_sweep")
    if isinstance(sw, dict):
        worst = max(sw["full_chain_u8"], key=lambda r: r["auto_over_best"])
        row("launch-size sweep, full chain: worst `auto_over_best` over the sizes", f"{worst['auto_over_best']} at B = {worst['blocks_per_launch']} blocks ({worst['auto']['route']})", B + " (launch_size_sweep)")
        pts = ", ".join(f"{r['blocks_per_launch']}: {r['auto']['us_per_launch']} µs" for r in sw["full_chain_u8"])
        row("… µs per launch on the library's own route, by blocks per launch", pts, B)
        w1 = max(sw["config1_cfloat_decimator"], key=lambda r: r["auto_over_best"])
        row("launch-size sweep, configs[1] (cfloat decimator): worst `auto_over_best`", f"{w1['auto_over_best']} at B = {w1['blocks_per_launch']} blocks", B + " (launch_size_sweep)")
        pts1 = ", ".join(f"{r['blocks_per_launch']}: {r['auto']['us_per_launch']} µs" + (f" (round 5's systolic form {r['systolic_kernel']['us_per_launch']}, tile kernel {r['tile_kernel']['us_per_launch']})" if r['blocks_per_launch'] in (512, 2048) and 'systolic_kernel' in r else "") for r in sw["config1_cfloat_decimator"])
        row("… µs per launch, by blocks per launch", pts1, B)
    cpu = full.get("cpu_baseline") or {}
    if cpu:
        row("CPU baseline on the GPU box's host (reference's C kernels, compiled caller)", f"{cpu['value'] / 1e3:.2f} Gsample/s on {cpu['cores']} threads ({cpu.get('physical_cores')} cores); single thread {cpu['single_thread_value']:.0f} Msample/s", B)
    pw = full.get("power") or {}
    if pw.get("rows"):
        r = pw["rows"]
        row("socket power / sclk while running (cap %s W)" % pw.get("cap_w"), "; ".join(f"{k.split('_2^')[0]}: {v.get('mean_w')} W at {v.get('mean_sclk_mhz')} MHz" for k, v in r.items()), B)
    out = "### Current numbers (MI355X, one GPU; generated by `tools/design_numbers.py` from the files named)\n\n| quantity | value | source |\n|---|---|---|\n" + "\n".join(rows) + "\n"
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = s.index("<!-- numbers:begin -->") + len("<!-- numbers:begin -->"), s.index("<!-- numbers:end -->")
        open(p, "w").write(s[:a] + "\n" + out + s[b:])
        print("DESIGN.md updated")
    else:
        print(out)


if __name__ == "__main__":
    main()
