"""Device-resident throughput of every kernel family behind the device stream API:
which ones have an LDS-tiled fast path and which run on the generic one-thread-per-output kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def timeit(fn, iters=10, warm=5):
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    print(L.device_name())
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 26
    xr = torch.rand(n, device="cuda") * 2 - 1
    xc = torch.rand(2 * n, device="cuda") * 2 - 1
    out = torch.empty(2 * n, device="cuda")
    t127, t191, h64 = S.taps_decim127(), S.taps_resamp191(), S.taps_audio_half64()
    rows = []

    def fir(name, desc, inp, width, K, unit_in):
        t = timeit(lambda: desc.run(inp.data_ptr(), 0, out.data_ptr(), 0, K, 8192, stream=st))
        rows.append((name, unit_in / t / 1e9))

    for order, oname in ((L.ORDER_AVX, "AVX"), (L.ORDER_SSE, "SSE"), (L.ORDER_SCALAR, "scalar")):
        d = L.Decimator(8, t127, order, complex_=True)
        fir(f"decimate /8 128 taps complex [{oname}]", d, xc, 2, (n - 128) // 8 + 1, n)
    d = L.Decimator(8, t127, L.ORDER_AVX)
    fir("decimate /8 128 taps real [AVX]", d, xr, 1, (n - 128) // 8 + 1, n)
    f = L.Filter(t127, L.ORDER_AVX, complex_=True)
    m = 1 << 24
    fir("filter 128 taps complex [AVX]", f, xc, 2, m - 127, m)
    f = L.Filter(t127, L.ORDER_AVX)
    fir("filter 128 taps real [AVX]", f, xr, 1, m - 127, m)
    f = L.Filter(h64, L.ORDER_AVX, sym=True)
    fir("filter 64 half-taps symmetric real [AVX]", f, xr, 1, n - 127, n)
    f = L.Filter(h64, L.ORDER_SSE, sym=True)
    fir("filter 64 half-taps symmetric real [SSE]", f, xr, 1, m - 127, m)
    r = L.Resampler(3, 10, t191, L.ORDER_AVX)
    fir("resample 3/10 191 taps real [AVX]", r, xr, 1, (n * 3 - 192) // 10 + 1, n)
    r = L.Resampler(3, 10, t191, L.ORDER_AVX, complex_=True)
    fir("resample 3/10 191 taps complex [AVX]", r, xc, 2, (m * 3 - 192) // 10 + 1, m)
    r = L.Resampler(5, 7, t191, L.ORDER_AVX)
    fir("resample 5/7 191 taps real [AVX]", r, xr, 1, (m * 5 - 200) // 7 + 1, m)
    t = timeit(lambda: L.check(L.lib.sdrhip_fm_demod_run(st, xc.data_ptr(), 0, out.data_ptr(), 0, n, 0.0, 0.0)))
    rows.append(("fmDemod", n / t / 1e9))
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: L.check(L.lib.sdrhip_convert_u8_run(st, u8.data_ptr(), out.data_ptr(), 2 * n)))
    rows.append(("convert u8 -> cfloat (standalone)", n / t / 1e9))
    for name, g in rows:
        print(f"{name:48s} {g:9.1f} G input elements/s")


if __name__ == "__main__":
    main()
