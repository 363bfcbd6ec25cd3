"""Real decimators by 2 / 4 / 8 / 16: their own kernel (kernels_decimate_real.hip) against the lane-split one
(SDRHIP_DECIM_REAL16=0 in the environment switches the former off; run the tool once each way)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
from family_bench import timeit


def main():
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 26
    xr = torch.rand(n + 4096, device="cuda") * 2 - 1
    out = torch.empty(n // 2 + 4096, device="cuda")
    print("SDRHIP_DECIM_REAL16 =", os.environ.get("SDRHIP_DECIM_REAL16", "(unset: on)"))
    for D, nt, order in ((8, 128, L.ORDER_AVX), (8, 128, L.ORDER_SSE), (4, 128, L.ORDER_AVX), (4, 128, L.ORDER_SSE), (2, 128, L.ORDER_AVX),
                         (2, 128, L.ORDER_SSE), (16, 128, L.ORDER_AVX), (16, 256, L.ORDER_AVX), (8, 40, L.ORDER_AVX), (8, 500, L.ORDER_AVX),
                         (4, 60, L.ORDER_SSE)):
        m = n if D >= 8 else n // 4
        d = L.Decimator(D, S.gauss_taps(nt, nt + D), order)
        K = (m - d.num_coeffs) // D + 1
        for seam in (8192, 0):
            c0 = L.lib.sdrhip_debug_decimate_real16_launches()
            t = timeit(lambda: d.run(xr.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            took = L.lib.sdrhip_debug_decimate_real16_launches() > c0
            print(f"decimate /{D} {d.num_coeffs} taps real [{'AVX' if order == L.ORDER_AVX else 'SSE'}] seam {seam:5d}: "
                  f"{m / t / 1e9:7.1f} G inputs/s  {K * d.num_coeffs / t / 1e12:6.2f} T MAC/s  {'own kernel' if took else 'other kernel'}")


def sym():
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 26
    xr = torch.rand(n + 4096, device="cuda") * 2 - 1
    out = torch.empty(n // 2 + 4096, device="cuda")
    for D, nh, order in ((2, 64, L.ORDER_AVX), (2, 64, L.ORDER_SSE), (4, 64, L.ORDER_AVX), (8, 64, L.ORDER_AVX), (8, 200, L.ORDER_AVX), (16, 64, L.ORDER_AVX)):
        m = n if D >= 8 else n // 4
        d = L.Decimator(D, S.gauss_taps(nh, nh + D), order, sym=True)
        K = (m - 2 * nh) // D + 1
        for seam in (8192, 0):
            c0 = L.lib.sdrhip_debug_decimate_real16_launches()
            t = timeit(lambda: d.run(xr.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            took = L.lib.sdrhip_debug_decimate_real16_launches() > c0
            print(f"decimate /{D} {nh} half-taps symmetric real [{'AVX' if order == L.ORDER_AVX else 'SSE'}] seam {seam:5d}: "
                  f"{m / t / 1e9:7.1f} G inputs/s  {K * nh / t / 1e12:6.2f} T sym-MAC/s  {'own kernel' if took else 'other kernel'}")


if __name__ == "__main__":
    sym()
    main()
