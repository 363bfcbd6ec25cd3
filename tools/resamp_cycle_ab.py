"""Real resamplers with an odd decimation: the thread-per-cycle kernel (kernels_resample_cycle.hip) against the lane-split one
(SDRHIP_RESAMP_CYCLE=0 in the environment switches the former off; run the tool once each way)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
from family_bench import timeit


def main():
    st = torch.cuda.current_stream().cuda_stream
    m = 1 << 24
    xr = torch.rand(m + 4096, device="cuda") * 2 - 1
    out = torch.empty(m + 4096, device="cuda")
    t191 = S.taps_resamp191()
    print("SDRHIP_RESAMP_CYCLE =", os.environ.get("SDRHIP_RESAMP_CYCLE", "(unset: on)"))
    for I, D, taps, order in ((2, 3, t191, L.ORDER_AVX), (5, 7, t191, L.ORDER_AVX), (3, 5, t191, L.ORDER_AVX), (1, 3, t191, L.ORDER_AVX),
                              (4, 5, S.gauss_taps(150, 9), L.ORDER_AVX), (6, 7, S.gauss_taps(700, 13), L.ORDER_AVX),
                              (2, 3, t191, L.ORDER_SSE), (5, 7, t191, L.ORDER_SSE)):
        r = L.Resampler(I, D, taps, order)
        K = (m * I - len(taps) - I) // D
        for seam in (8192, 0):
            c0 = L.lib.sdrhip_debug_resample_cycle_launches()
            t = timeit(lambda: r.run(xr.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            took = L.lib.sdrhip_debug_resample_cycle_launches() > c0
            print(f"resample {I}/{D} {len(taps)} taps [{'AVX' if order == L.ORDER_AVX else 'SSE'}] seam {seam:5d}: "
                  f"{m / t / 1e9:7.1f} G inputs/s  {K * ((len(taps) + I - 1) // I) / t / 1e12:6.2f} T MAC/s  {'cycle kernel' if took else 'other kernel'}")


def cplx():
    st = torch.cuda.current_stream().cuda_stream
    m = 1 << 24
    xc = torch.rand(2 * (m + 4096), device="cuda") * 2 - 1
    out = torch.empty(2 * (m + 4096), device="cuda")
    t191 = S.taps_resamp191()
    for I, D, taps, order in ((2, 3, t191, L.ORDER_AVX), (5, 7, t191, L.ORDER_AVX), (3, 5, t191, L.ORDER_AVX), (2, 3, t191, L.ORDER_SSE), (5, 7, t191, L.ORDER_SSE)):
        r = L.Resampler(I, D, taps, order, complex_=True)
        K = (m * I - len(taps) - I) // D
        for seam in (8192, 0):
            c0 = L.lib.sdrhip_debug_resample_cycle_launches()
            t = timeit(lambda: r.run(xc.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            took = L.lib.sdrhip_debug_resample_cycle_launches() > c0
            print(f"resample {I}/{D} {len(taps)} taps complex [{'AVX' if order == L.ORDER_AVX else 'SSE'}] seam {seam:5d}: "
                  f"{m / t / 1e9:7.1f} G samples/s  {2 * K * ((len(taps) + I - 1) // I) / t / 1e12:6.2f} T MAC/s  {'cycle kernel' if took else 'other kernel'}")


if __name__ == "__main__":
    cplx()
    main()
