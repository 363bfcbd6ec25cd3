"""Turn gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the committed summaries under profiles/."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)


    def short(name):
        n = name.replace("void ", "").replace("sdrhip::(anonymous namespace)::", "")
        return n.split("(")[0]


    # 1. kernel-trace stats
    rows = list(csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns", "stddev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    stats_avg = {short(r["Name"]): float(r["AverageNs"]) for r in rows}


    # 2. PMC passes: per kernel, mean over dispatches of the per-dispatch sum over dimensions
    def pmc(sub):
        path = os.path.join(src, sub, "bench_counter_collection.csv")
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = {}
        for r in csv.DictReader(open(path)):
            k = (short(r["Kernel_Name"]), r["Dispatch_Id"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        out = collections.defaultdict(lambda: collections.defaultdict(list))
        for (k, d), cs in per.items():
            for c, v in cs.items():
                out[k][c].append(v)
            out[k]["_dur_ns"].append(dur[(k, d)])
        return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


    allc = collections.defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
        for k, cs in pmc(sub).items():
            for c, v in cs.items():
                allc[k][c if c != "_dur_ns" else f"dur_ns[{sub}]"] = v
    ours = {k: v for k, v in allc.items() if k.startswith("k_")}
    cols = sorted({c for v in ours.values() for c in v})
    with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + cols)
        for k, v in sorted(ours.items()):
            w.writerow([k] + [f"{v.get(c, float('nan')):.1f}" for c in cols])

    # 3. K2 HBM traffic, per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
    #    FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced
    #    streaming read on gfx950 -> doubled; WRITE_SIZE is uncalibrated -> reported as measured.
    bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
    # since round 3 a K2 launch is two kernels: the FULL-tile instantiation (all but at most 64 tiles) and the general one for
    # the remainder; the traffic of a launch is their sum, the kernel named is the one that moves (nearly) all of it
    # since round 4 the launch is ONE kernel again, the systolic walk (whole and ragged strips in the same launch)
    k2s = [k for k in ours if k.startswith(("k_decimate_systolic<true", "k_decimate_c4<")) and "FETCH_SIZE" in ours[k]]
    k2 = max(k2s, key=lambda k: ours[k].get("FETCH_SIZE", 0.0))
    fetch_kib = sum(ours[k].get("FETCH_SIZE", 0.0) for k in k2s)
    write_kib = sum(ours[k].get("WRITE_SIZE", 0.0) for k in k2s)
    samples = bench["roofline"]["algorithmic_bytes_per_launch"] / 3.0
    sys.path.insert(0, ROOT)
    from bench import k2_source_sha256
    traffic = {
        "kernel": k2,
        "kernels_fast_sha256": k2_source_sha256(),      # decimate_tile.hpp + kernels_fast.hip
        "samples_per_launch": int(round(samples)),
        "FETCH_SIZE_KiB_raw": fetch_kib,
        "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_bytes_corrected": 2.0 * fetch_kib * 1024.0,
        "write_bytes": write_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 5 --warmup 2 "
               "--no-cpu-baseline`, mean over the kernel's dispatches of the sum over counter dimensions; FETCH_SIZE x2 "
               "(gfx950 wide-coalesced-read under-count, MI355X_MICROARCH.md HBM section), x1024 (KiB -> B)",
    }
    json.dump(traffic, open(os.path.join(dst, "k2_traffic.json"), "w"), indent=1)
    json.dump(bench, open(os.path.join(dst, f"{tag}_bench_unprofiled.json"), "w"), indent=1)

    # 3b. the cfloat-in instantiation (BASELINE configs[1]), from the separate passes over tools/prof_k2.py
    try:
        c_f = pmc("pmc_fetch_k2c")
        c_w = pmc("pmc_write_k2c")
        # template arguments <D, P, R, NT, U8, TC, GUARD, NP, ORD>: the cfloat-in instantiation has U8 = false
        kcs = [k for k in c_f if k.startswith("k_decimate_systolic<false")
               or (k.startswith("k_decimate_c4<") and k[k.index("<") + 1:].split(",")[4].strip() == "false")]
        kc = max(kcs, key=lambda k: c_f[k].get("FETCH_SIZE", 0.0))
        n_c = 1 << 27
        fk = sum(c_f[k].get("FETCH_SIZE", 0.0) for k in kcs)
        wk = sum(c_w[k].get("WRITE_SIZE", 0.0) for k in kcs if k in c_w)
        json.dump({"kernel": kc, "samples_per_launch": n_c, "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB_raw": wk,
                   "fetch_bytes_corrected": 2.0 * fk * 1024.0, "write_bytes": wk * 1024.0,
                   "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0, "algorithmic_bytes_per_launch": 9.0 * n_c,
                   "ratio": (2.0 * fk * 1024.0 + wk * 1024.0) / (9.0 * n_c),
                   "kernels_fast_sha256": traffic["kernels_fast_sha256"],
                   "how": "as k2_traffic.json, over `python tools/prof_k2.py 27 f32 8192 400 200` (cfloat IQ in, 2^27 samples per launch, 8192-sample seams: the "
                          "seam fix-up kernel's traffic is a row of its own in the pmc csv)"},
                  open(os.path.join(dst, "k2c_traffic.json"), "w"), indent=1)
    except Exception as e:          # noqa: BLE001
        print("no k2c passes:", e)

    # 3c. round 3: kernel-trace rows of the cfloat-in kernel on its own and of the one-kernel chain at the 2^20-sample shard
    for sub, name in (("stats_k2c", f"{tag}_k2c_kernel_stats.csv"), ("stats_shard", f"{tag}_shard_kernel_stats.csv")):
        path = os.path.join(src, sub, "bench_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        with open(os.path.join(dst, name), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns", "stddev"])
            for r in csv.DictReader(open(path)):
                if "sdrhip" in r["Name"]:
                    w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
            # round 4: the traced run warms up with the same kernel (tools/prof_k2.py: 400 of 600 launches); the sustained figure is the
            # mean over each kernel's LAST 200 dispatches, and what bench.py times is the span from one launch's start to the next one's
            tpath = os.path.join(src, sub, "bench_kernel_trace.csv")
            if sub == "stats_k2c" and os.path.exists(tpath):
                per = collections.defaultdict(list)
                for r in csv.DictReader(open(tpath)):
                    if "sdrhip" in r["Kernel_Name"]:
                        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
                w.writerow([])
                w.writerow(["kernel", "last_200_dispatches_avg_ns", "last_200_period_ns (start to start, = the launch time bench.py sees)"])
                for k, v in sorted(per.items()):
                    v.sort()
                    last = v[-200:]
                    avg = sum(e - b for b, e in last) / len(last)
                    period = (last[-1][0] - last[0][0]) / max(1, len(last) - 1)
                    w.writerow([k, f"{avg:.0f}", f"{period:.0f}"])
    for sub, name in (("pmc_sq_k2c", f"{tag}_k2c_pmc.csv"), ("pmc_shard", f"{tag}_shard_pmc.csv")):
        if not os.path.exists(os.path.join(src, sub, "bench_counter_collection.csv")):
            continue
        res = {k: v for k, v in pmc(sub).items() if k.startswith("k_")}
        cols2 = sorted({c for v in res.values() for c in v})
        with open(os.path.join(dst, name), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel"] + cols2 + ["clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / dur_ns"])
            for k, v in sorted(res.items()):
                clk = v.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0 / v.get("_dur_ns", float("nan"))
                w.writerow([k] + [f"{v.get(c, float('nan')):.1f}" for c in cols2] + [f"{clk:.3f}"])

    # 4. derived table for the README
    print("kernel                                   avg_us(stats)  clock_GHz  valu_quad_busy  lds_conflict/idx")
    for k, v in sorted(ours.items()):
        d = v.get("dur_ns[pmc_lds]", float("nan"))
        clock = v.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0 / d if d == d else float("nan")
        dsq = v.get("dur_ns[pmc_sq]", float("nan"))
        busy = v.get("SQ_ACTIVE_INST_VALU", float("nan")) * 4.0 / 1024.0 / (dsq * clock) if dsq == dsq else float("nan")
        lds = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
        print(f"{k[:40]:40s} {stats_avg.get(k, float('nan'))/1e3:12.1f} {clock:10.2f} {busy:14.2f} {lds:12.3f}")
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
