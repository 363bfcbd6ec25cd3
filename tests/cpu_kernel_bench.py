"""Single-thread rates of the reference's own C kernels (oracle/_ref) on this host, at the block sizes of
BASELINE.json's configs: what one pipeline thread of the reference sustains per stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # lives under tests/: only tests, smoke() and bench.py's cpu_baseline may touch oracle/
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle.oracle import Oracle, Ref, have_ref
import signals as S

o = Oracle()
ref = Ref() if have_ref() else None
B = 8192


def rate(fn, units, secs=1.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn(); n += 1
    return n * units / (time.perf_counter() - t0) / 1e6


print("host cores:", os.cpu_count(), "| kernels from", "the reference's compiled C (oracle/_ref)" if ref else "the restatement (oracle)")
u8 = S.iq_u8(B)
x = S.cfloat_block(B)
xr = S.real_block(B)
h127 = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
hd = np.repeat(h127, 2)
half = S.taps_audio_half64()
prep = o.prepare_coeffs(8, 3, 10, S.taps_resamp191())
x64k = S.real_block(65536)
rows = []
if ref:
    rows.append(("convertCAVX, 8192-sample u8 IQ block", rate(lambda: ref.convert("convertCAVX", u8), B)))
    rows.append(("decimateAVXRC /8, 128 taps, 8192-sample block (configs[1])", rate(lambda: ref.decim("decimateAVXRC", 1009, 8, hd, x, True), B)))
    rows.append(("filterAVXSymmetricRR 64 half-taps, 8192 floats (configs[0])", rate(lambda: ref.filt("filterAVXSymmetricRR", 8065, half, xr), B)))
    rows.append(("resampleAVXRR 3/10 191 taps, 65536 floats (configs[3])", rate(lambda: ref.resample("resampleAVXRR", 19642, prep, 0, x64k), 65536)))
rows.append(("fmDemod (restated GHC formula, C), 8192 samples", rate(lambda: o.fm_demod(x), B)))
for name, r in rows:
    print(f"{name:66s} {r:9.1f} M input elements/s")
