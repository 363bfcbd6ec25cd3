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
    m = m0 = 1 << 24

    def fir(name, desc, inp, width, K, unit_in, macs_per_out=None):
        before = L.lib.sdrhip_debug_tiled_launches()
        t = timeit(lambda: desc.run(inp.data_ptr(), 0, out.data_ptr(), 0, K, 8192, stream=st))
        tiled = L.lib.sdrhip_debug_tiled_launches() > before
        if macs_per_out is None:
            macs_per_out = desc.num_coeffs * width if not hasattr(desc, "in_offset") else None
        rows.append((name + (" *" if tiled else ""), unit_in / t / 1e9, (K * macs_per_out / t / 1e12) if macs_per_out else None))

    for order, oname in ((L.ORDER_AVX, "AVX"), (L.ORDER_SSE, "SSE"), (L.ORDER_SCALAR, "scalar")):
        d = L.Decimator(8, t127, order, complex_=True)
        fir(f"decimate /8 128 taps complex [{oname}]", d, xc, 2, (n - 128) // 8 + 1, n)
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    for D, nt in ((8, 31), (8, 63), (8, 95), (8, 100), (4, 63), (4, 127), (16, 63), (16, 127)):
        nn = n if D >= 8 else m0 * 2
        d = L.Decimator(D, S.gauss_taps(nt, nt), L.ORDER_AVX, complex_=True)
        K = (nn - d.num_coeffs) // D + 1
        fir(f"decimate /{D} {d.num_coeffs} taps complex [AVX] (guarded k_decimate_c4)", d, xc, 2, K, nn)
        t = timeit(lambda: d.run_u8(u8.data_ptr(), 0, out.data_ptr(), 0, K, 8192, stream=st))
        rows.append((f"decimate /{D} {d.num_coeffs} taps complex [AVX], u8 IQ in", nn / t / 1e9, K * d.num_coeffs * 2 / t / 1e12))
    del u8
    d = L.Decimator(8, t127, L.ORDER_AVX)
    fir("decimate /8 128 taps real [AVX]", d, xr, 1, (n - 128) // 8 + 1, n)
    d = L.Decimator(8, t127, L.ORDER_SSE)
    fir("decimate /8 128 taps real [SSE]", d, xr, 1, (n - 128) // 8 + 1, n)
    d = L.Decimator(4, t127, L.ORDER_AVX)
    fir("decimate /4 128 taps real [AVX]", d, xr, 1, (n - 128) // 4 + 1, n)
    d = L.Decimator(2, h64, L.ORDER_AVX, sym=True)
    fir("decimate /2 64 half-taps symmetric real [AVX]", d, xr, 1, (n - 128) // 2 + 1, n, macs_per_out=64)
    d = L.Decimator(4, t127, L.ORDER_AVX, complex_=True)
    fir("decimate /4 128 taps complex [AVX]", d, xc, 2, (m0 - 128) // 4 + 1, m0)
    d = L.Decimator(10, S.taps_decim51(), L.ORDER_AVX, complex_=True)
    fir("decimate /10 52 taps complex [AVX]", d, xc, 2, (n - 52) // 10 + 1, n)
    f = L.Filter(t127, L.ORDER_AVX, complex_=True)
    fir("filter 128 taps complex [AVX]", f, xc, 2, m - 127, m)
    f = L.Filter(t127, L.ORDER_AVX)
    fir("filter 128 taps real [AVX]", f, xr, 1, m - 127, m)
    f = L.Filter(h64, L.ORDER_AVX, sym=True)
    fir("filter 64 half-taps symmetric real [AVX]", f, xr, 1, n - 127, n)
    f = L.Filter(h64, L.ORDER_SSE, sym=True)
    fir("filter 64 half-taps symmetric real [SSE]", f, xr, 1, m - 127, m)
    r = L.Resampler(3, 10, t191, L.ORDER_AVX)
    fir("resample 3/10 191 taps real [AVX]", r, xr, 1, (n * 3 - 192) // 10 + 1, n)
    r = L.Resampler(3, 10, t191, L.ORDER_SSE)
    fir("resample 3/10 191 taps real [SSE]", r, xr, 1, (n * 3 - 192) // 10 + 1, n, macs_per_out=64)
    r = L.Resampler(3, 10, t191, L.ORDER_AVX, complex_=True)
    fir("resample 3/10 191 taps complex [AVX]", r, xc, 2, (m * 3 - 192) // 10 + 1, m, macs_per_out=128)
    r = L.Resampler(5, 7, t191, L.ORDER_AVX)
    fir("resample 5/7 191 taps real [AVX]", r, xr, 1, (m * 5 - 200) // 7 + 1, m, macs_per_out=40)
    r = L.Resampler(2, 3, t191, L.ORDER_AVX)
    fir("resample 2/3 191 taps real [AVX]", r, xr, 1, (m * 2 - 192) // 3 + 1, m, macs_per_out=96)
    r = L.Resampler(1, 4, t127, L.ORDER_AVX)
    fir("resample 1/4 127 taps real [AVX]", r, xr, 1, (m - 128) // 4 + 1, m, macs_per_out=128)
    t = timeit(lambda: L.check(L.lib.sdrhip_fm_demod_run(st, xc.data_ptr(), 0, out.data_ptr(), 0, n, 0.0, 0.0)))
    rows.append(("fmDemod", n / t / 1e9, None))
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: L.check(L.lib.sdrhip_convert_u8_run(st, u8.data_ptr(), out.data_ptr(), 2 * n)))
    rows.append(("convert u8 -> cfloat (standalone)", n / t / 1e9, None))
    print("(* = served by the general LDS-tiled kernel, kernels_split.hip; T MAC/s = unfused multiply-add pairs, peak ~36-39)")
    for name, g, macs in rows:
        print(f"{name:52s} {g:9.1f} G input elements/s" + (f"  {macs:6.2f} T MAC/s" if macs else ""))


if __name__ == "__main__":
    main()
