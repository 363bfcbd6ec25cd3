"""BASELINE configs[1]'s kernel (cfloat IQ in, 127 -> 128 taps, decimate by 8) on 2^27 samples: launch time with and without
the reference Pipes' 8192-sample seams (what the seam fix-up launch costs), then the sustained launch time over 3000
back-to-back launches in windows of 100 (the first ~100 launches of a fresh process run slow while the clocks ramp).
Run on a GPU box:  python tools/k2c_seam_cost.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import sdr_amd.lib as L
    import signals as S
    n = 1 << 27
    K = (n - 128) // 8 + 1
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    out = torch.empty(2 * K + 64, device="cuda")
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(seam, reps=20, warm=10):
        for _ in range(warm):
            dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e0.record(stream)
        for _ in range(reps):
            dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(3):
        a, b = t(0), t(8192)
        print(f"cfloat /8 128 taps 2^27 samples: no seams {a:.4f} ms ({8 * n / a / 1e9 / 8000 * 1e3:.3f} of read roof), "
              f"8192-sample seams {b:.4f} ms ({8 * n / b / 1e9 / 8000 * 1e3:.3f}); fix-up costs {1e3 * (b - a):.1f} us")
    print("sustained, 8192-sample seams, ms per launch in windows of 100 launches:")
    print(" ".join(f"{t(8192, reps=100, warm=0):.4f}" for _ in range(30)))


if __name__ == "__main__":
    main()
