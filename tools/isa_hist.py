"""Instruction histograms of the pass's three stage kernels, from the compiler's own assembly (no GPU needed):
    python tools/isa_hist.py > profiles/r04_isa_histograms.txt
Compiles kernels_systolic.hip and kernels_chain.hip to gfx950 assembly with the library's flags and counts mnemonics per kernel
(whole kernel, every path: the systolic decimator holds a whole-strip and a ragged-strip body, the fused resampler an interior and
an edge loader plus the full fmDemod form behind the vote)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdr_amd import build as B

KERNELS = [("kernels_systolic.hip", "k_decimate_systolicILb1ELi1ELb1ELb0ELi16EE", "k_decimate_systolic<u8, zero tap skipped> (chain K2, fix-up as a second launch: launches past 2^26 samples)"),
           ("kernels_systolic.hip", "k_decimate_systolicILb1ELi1ELb1ELb1ELi16EE", "k_decimate_systolic<u8, zero tap skipped, seam fix-up workgroups inside> (launches up to 2^26 samples)"),
           ("kernels_systolic.hip", "k_decimate_systolicILb0ELi0ELb1ELb0ELi16EE", "k_decimate_systolic<cfloat, non-temporal loads> (BASELINE configs[1] at 2^27 samples)"),
           ("kernels_systolic.hip", "k_decimate_systolicILb0ELi0ELb0ELb1ELi16EE", "k_decimate_systolic<cfloat, plain loads, seam fix-up workgroups inside> (64 ... 285 MB of input)"),
           ("kernels_systolic.hip", "k_decimate_systolicILb1ELi12ELb1ELb0ELi8EE", "k_decimate_systolic<u8, 64-tap instantiation, 12 padding taps skipped> (the reference example's 52-tap RF filter)"),
           ("kernels_chain.hip", "k_resample3_fastILi3ELi64ELi4ELi3ELi3ELi256ELb1ELi8ELb1EE", "k_resample3_fast<.., DEMOD, 8 lanes, packed pairs> (fmDemod + 3/10 resampler)"),
           ("kernels_chain.hip", "k_fir_real8_fastILb1ELi4ELi256ELi8EE", "k_fir_real8_fast<symmetric, 8 lanes> (audio filter + gain)")]


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = {}
        for f in sorted({k[0] for k in KERNELS}):
            out = os.path.join(td, f + ".s")
            cmd = [B.HIPCC] + B.FLAGS + B.FILE_FLAGS.get(f, []) + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(B.CSRC, f), "-o", out]
            subprocess.run(cmd, check=True, capture_output=True)
            asm[f] = open(out).read().splitlines()
        for f, key, title in KERNELS:
            lines = asm[f]
            start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\w+:", l))
            end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
            hist = collections.Counter()
            for l in lines[start + 1:end + 1]:
                l = l.strip()
                if not l or l.startswith((";", ".")) or l.endswith(":"):
                    continue
                hist[l.split()[0]] += 1
            meta = {}
            for i, l in enumerate(lines):
                if ".name:" in l and key in l:
                    for m in lines[i:i + 14]:
                        for k in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size"):
                            if k + ":" in m:
                                meta[k] = m.split(":")[1].strip()
            for i, l in enumerate(lines):
                if ".name:" in l and key in l:
                    for m in lines[max(0, i - 12):i]:
                        if ".group_segment_fixed_size:" in m:
                            meta[".group_segment_fixed_size"] = m.split(":")[1].strip()
            print(f"== {title}")
            print("   " + "  ".join(f"{k[1:]} {v}" for k, v in sorted(meta.items())))
            total = sum(hist.values())
            valu = sum(v for k, v in hist.items() if k.startswith("v_"))
            print(f"   {total} instructions, {valu} VALU, {sum(v for k, v in hist.items() if k.startswith('ds_'))} LDS, "
                  f"{sum(v for k, v in hist.items() if k.startswith(('global_', 'buffer_')))} global, "
                  f"{sum(v for k, v in hist.items() if k.startswith('v_fma') or k.startswith('v_pk_fma') or k.startswith('v_mfma'))} fma/mfma "
                  f"(the divisions' own v_fma_f32 / v_fmac_f32 are the IEEE division sequence)")
            for k, v in hist.most_common(28):
                print(f"   {v:6d}  {k}")
            print()


if __name__ == "__main__":
    main()
