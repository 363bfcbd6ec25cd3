"""Per-call cost of the drop-in symbols (layer 1 of include/sdr_hip.h: host pointers in and out, synchronous) at the
block sizes of BASELINE.json's configs, beside the reference's own compiled C (oracle/_ref) on one host thread --
what a Level-0 relink of the reference pays per FFI call.  Timed through the same ctypes path for both
libraries.  Lives under tests/: only tests, smoke() and bench.py's cpu_baseline may touch oracle/.  Run on a GPU box:
    python tests/dropin_call_bench.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def per_call_us(fn, secs=0.7):
    for _ in range(20):
        fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn(); n += 1
    return (time.perf_counter() - t0) / n * 1e6


def main():
    import sdr_amd.lib as L
    from oracle.oracle import Oracle, Ref, have_ref, duplicate
    import signals as S
    o = Oracle()
    ref = Ref() if have_ref() else None
    B = 8192
    u8 = S.iq_u8(B)
    x = S.cfloat_block(B)
    xr = S.real_block(B)
    hd = duplicate(np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)]))
    half = S.taps_audio_half64()
    prep = o.prepare_coeffs(8, 3, 10, S.taps_resamp191())
    x64k = S.real_block(65536)
    D = L.DropIn
    cases = [
        ("convertCAVX, 8192-sample u8 IQ block", B, lambda lib: lib.convert("convertCAVX", u8)),
        ("decimateAVXRC /8 128 taps, 8192-sample block (configs[1])", B, lambda lib: lib.decim("decimateAVXRC", 1009, 8, hd, x, True)),
        ("filterAVXSymmetricRR 64 half-taps, 8192 floats (configs[0])", B, lambda lib: lib.filt("filterAVXSymmetricRR", 8065, half, xr)),
    ]
    print(f"{'call':62s} {'GPU us/call':>12s} {'M el/s':>9s} {'CPU us/call':>12s} {'M el/s':>9s}")
    for name, units, fn in cases:
        g = per_call_us(lambda: fn(D))
        c = per_call_us(lambda: fn(ref)) if ref else float("nan")
        print(f"{name:62s} {g:12.1f} {units / g:9.1f} {c:12.1f} {units / c:9.1f}")
    # the resampler's argument list differs between the two wrappers
    incs, groups, nc = prep["increments"], prep["groups"], prep["num_coeffs"]
    g = per_call_us(lambda: D.resample("resampleAVXRR", 19642, nc, 0, incs, groups, x64k))
    c = per_call_us(lambda: ref.resample("resampleAVXRR", 19642, prep, 0, x64k)) if ref else float("nan")
    print(f"{'resampleAVXRR 3/10 191 taps, 65536 floats (configs[3])':62s} {g:12.1f} {65536 / g:9.1f} {c:12.1f} {65536 / c:9.1f}")


if __name__ == "__main__":
    main()
