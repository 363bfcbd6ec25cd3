"""The low-rate stages of the chain as stand-alone launches at the sizes of the 2^29-sample pass (2^26 decimator outputs):
fmDemod, the 3/10 resampler, the symmetric filter -- microseconds per launch, TB/s of algorithmic traffic.  Environment
knobs of the library (e.g. SDRHIP_RESAMP_STREAM_WGS) apply; results are checked against a first run's CRC."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def timeit(fn, iters=200, warm=100):
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
    log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    nk = 1 << log2k
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(5)
    d = torch.rand(2 * nk, device="cuda") * 2 - 1
    y = torch.empty(nk, device="cuda")
    t = timeit(lambda: L.check(L.lib.sdrhip_fm_demod_run(st, d.data_ptr(), 0, y.data_ptr(), 0, nk, 0.0, 0.0)))
    print(f"fmDemod    {nk} samples: {t*1e6:8.1f} us  {12*nk/t/1e12:6.3f} TB/s")
    res = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
    m = (nk * 3 - 192) // 10 + 1
    z = torch.empty(m + 16, device="cuda")
    for seam in (8192, 0):
        t = timeit(lambda: res.run(y.data_ptr(), 0, z.data_ptr(), 0, m, seam, stream=st, out_block=seam))
        torch.cuda.synchronize()
        crc = zlib.crc32(z[:m].cpu().numpy().tobytes())
        print(f"resample   {nk} inputs seam {seam}: {t*1e6:8.1f} us  {(4*nk+4*m)/t/1e12:6.3f} TB/s  crc {crc:08x}")
    flt = L.Filter(S.taps_audio_half64(), L.ORDER_AVX, sym=True)
    q = m - 127
    a = torch.empty(q + 16, device="cuda")
    for seam in (8192, 0):
        t = timeit(lambda: flt.run(z.data_ptr(), 0, a.data_ptr(), 0, q, seam, stream=st))
        crc = zlib.crc32(a[:q].cpu().numpy().tobytes())
        print(f"filter     {m} inputs seam {seam}: {t*1e6:8.1f} us  {8*q/t/1e12:6.3f} TB/s  crc {crc:08x}")


if __name__ == "__main__":
    main()
