"""Run the K2 decimate kernel for rocprofv3: `python tools/prof_k2.py <log2 samples> <f32|u8|both> [seam] [warm] [reps]`.

Round 4 (VERDICT r03 "next" #3): the configuration bench.py measures -- 8192-sample seams (kernel + fix-up rows), `warm` launches
before the `reps` traced-and-counted ones so that the rows are taken in the sustained power state, not on a cold chip.  All
launches are traced; the summary's average therefore includes the warm-up launches (400 of 600 by default: the first ~100 of a
fresh process run 15-25 % slow), which is why warm-up and measurement use the SAME kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    seam = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    K = (n - 128) // 8 + 1
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    out = torch.empty(2 * K, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if mode in ("both", "f32"):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        for _ in range(warm):
            dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e0.record()
        for _ in range(reps):
            dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e1.record()
        torch.cuda.synchronize()
        print(f"f32 in, seam {seam}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch (HIP events over the {reps} launches after {warm} warm-up launches)")
    if mode in ("both", "u8"):
        u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
        for _ in range(warm):
            dec.run_u8(u8.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e0.record()
        for _ in range(reps):
            dec.run_u8(u8.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e1.record()
        torch.cuda.synchronize()
        print(f"u8 in, seam {seam}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch (HIP events over the {reps} launches after {warm} warm-up launches)")


if __name__ == "__main__":
    main()
