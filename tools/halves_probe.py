"""One stage kernel over the whole 2^26-element stream on one HIP stream against its two halves launched concurrently on two
streams (fork / join by events): fmDemod, the 3/10 resampler, the symmetric filter.  Wall time per whole stage, alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

nk = 1 << 26
s0 = torch.cuda.current_stream()
s1 = torch.cuda.Stream()
torch.manual_seed(5)
d = torch.rand(2 * nk, device="cuda") * 2 - 1
y = torch.empty(nk, device="cuda")
res = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
m = (nk * 3 - 192) // 10 + 1
z = torch.empty(m + 16, device="cuda")
flt = L.Filter(S.taps_audio_half64(), L.ORDER_AVX, sym=True)
q = m - 127
a = torch.empty(q + 16, device="cuda")
SEAM = 8192


def demod(lo, hi, st):
    L.check(L.lib.sdrhip_fm_demod_run(st.cuda_stream, d.data_ptr(), 0, y.data_ptr() + 4 * lo, lo, hi, 0.0, 0.0))


def resample(lo, hi, st):
    res.run(y.data_ptr(), 0, z.data_ptr() + 4 * lo, lo, hi, SEAM, stream=st.cuda_stream, out_block=SEAM)


def filt(lo, hi, st):
    flt.run(z.data_ptr(), 0, a.data_ptr() + 4 * lo, lo, hi, SEAM, stream=st.cuda_stream)


def whole(fn, n):
    fn(0, n, s0)


def halves(fn, n):
    h = (n // 2) // 3072 * 3072
    ev = torch.cuda.Event()
    ev.record(s0)
    s1.wait_event(ev)
    fn(h, n, s1)
    fn(0, h, s0)
    ev2 = torch.cuda.Event()
    ev2.record(s1)
    s0.wait_event(ev2)


def timeit(run, iters=100, warm=30):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    for _ in range(iters):
        run()
    e1.record(s0)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, fn, n in (("fmDemod", demod, nk), ("resample", resample, m), ("filter", filt, q)):
    whole(fn, n)
    torch.cuda.synchronize()
    ref = {"fmDemod": y, "resample": z, "filter": a}[name][:n].clone()
    halves(fn, n)
    torch.cuda.synchronize()
    same = bool(torch.equal({"fmDemod": y, "resample": z, "filter": a}[name][:n].view(torch.int32), ref.view(torch.int32)))
    rows = []
    for rnd in range(3):
        rows.append((timeit(lambda: whole(fn, n)), timeit(lambda: halves(fn, n))))
    print(f"{name:9s} whole / two halves on two streams (us): " + "  ".join(f"{w:6.1f}/{h:6.1f}" for w, h in rows) + f"   same bits: {same}")
