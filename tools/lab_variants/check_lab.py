"""Bit-equality of the lab variants with the product library (GPU box):

    python tools/lab_variants/build_lab.py && python tools/lab_variants/check_lab.py

Every variant here was built in rounds 4-5, found to give the same bits as the production kernels and to be slower, and left the
product in round 6 (VERDICT r05 "weak" 8).  This script is what their tests in tests/ used to assert, against the product's own
stage kernels through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdr_amd.lib as L      # noqa: E402
import signals as S          # noqa: E402

lab = C.CDLL(os.path.join(HERE, "liblab.so"))
vp, i64 = C.c_void_p, C.c_longlong
lab.sdrlab_resample3_systolic.argtypes = [vp, i64, C.c_int, i64, vp, C.c_int, vp]
lab.sdrlab_resample3_demod_stream.argtypes = [C.c_int, vp, i64, C.c_int, C.c_int, i64, vp, C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int]
lab.sdrlab_decimate_demod_systolic.argtypes = [vp, i64, i64, i64, i64, vp, vp, C.c_int, C.c_int, i64, vp]
lab.sdrlab_fm_demod_form.argtypes = [C.c_int, vp, vp, i64, C.c_int, C.c_float, C.c_float]
lab.sdrlab_fm_demod_form.restype = None
B = 8192


def same(a, b, what):
    a, b = a.view(torch.int32), b.view(torch.int32)
    bad = int((a != b).sum())
    print(("ok   " if bad == 0 else "FAIL ") + what + ("" if bad == 0 else f": {bad} of {a.numel()} differ"))
    return bad == 0


def groups_table():
    """the 3/10 resampler's polyphase groups as resamp_create lays them out: group g = taps off_g, off_g + 3, ... padded to 64"""
    h = S.taps_resamp191()
    offs = [0, 2, 1]                                       # prepareCoeffs (FilterInternal.hs:297-319) for I = 3, D = 10
    t = np.zeros((3, 64), np.float32)
    for g, o in enumerate(offs):
        v = h[o::3]
        t[g, :v.size] = v
    return torch.from_numpy(t).cuda()


def main():
    ok = True
    torch.manual_seed(7)
    # ---- fmDemod forms against the product's stand-alone kernel
    n = 1 << 22
    bits = torch.randint(0, 1 << 32, (2 * n,), dtype=torch.int64, device="cuda").to(torch.int32)
    x = bits.view(torch.float32).clone()
    x[: 2 * (n // 2)] = torch.rand(2 * (n // 2), device="cuda") * 2 - 1      # half ordinary samples, half arbitrary bit patterns
    ref = torch.empty(n, device="cuda")
    L.check(L.lib.sdrhip_fm_demod_run(None, x.data_ptr(), 0, ref.data_ptr(), 0, n, 0.0, 0.0))
    torch.cuda.synchronize()
    for form in range(5):
        out = torch.empty(n, device="cuda")
        lab.sdrlab_fm_demod_form(form, x.data_ptr(), out.data_ptr(), n, 0, 0.0, 0.0)
        torch.cuda.synchronize()
        nan = torch.isnan(ref)
        ok &= bool(torch.equal(torch.isnan(out), nan)) and same(out[~nan], ref[~nan], f"fmDemod form {form} on 2^22 samples (half arbitrary bit patterns)")

    # ---- the resampler variants: whole polyphase cycles from the stream start
    gt = groups_table()
    ncyc = 1 << 20
    avail = (ncyc - 1) * 10 + 7 + 64
    d = torch.rand(2 * (avail + 64), device="cuda") * 2 - 1                    # decimator output (cfloat)
    y = torch.empty(avail + 64, device="cuda")
    L.check(L.lib.sdrhip_fm_demod_run(None, d.data_ptr(), 0, y.data_ptr(), 0, avail + 64, 0.0, 0.0))
    res = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
    zref = torch.empty(3 * ncyc, device="cuda")
    res.run(y.data_ptr(), 0, zref.data_ptr(), 0, 3 * ncyc, 0)
    torch.cuda.synchronize()
    z = torch.zeros(3 * ncyc, device="cuda")
    took = lab.sdrlab_resample3_systolic(y.data_ptr(), 0, ncyc, avail, gt.data_ptr(), 64, z.data_ptr())
    torch.cuda.synchronize()
    ok &= took == 1 and same(z, zref, "3/10 resampler, register-resident systolic walk (round 4), 2^20 cycles")
    for mode, name in ((2, "streaming fmDemod + resampler (round 5)"), (37, "... cut for 37 workgroups"), (1002, "... LDS-DMA prefetch"), (1037, "... LDS-DMA, 37 workgroups")):
        z = torch.zeros(3 * ncyc, device="cuda")
        ykeep = torch.zeros(avail + 64, device="cuda")
        took = lab.sdrlab_resample3_demod_stream(mode, d.data_ptr(), 0, ncyc, 0, avail, gt.data_ptr(), 64, z.data_ptr(), ykeep.data_ptr(), 0, 0, 160, 256)
        torch.cuda.synchronize()
        ok &= took == 1 and same(z, zref, f"{name}, 2^20 cycles")
        ok &= same(ykeep[:256], y[:256], "    ... the launch's leading edge of y")

    # ---- K2 + K3 in one launch against decimator + fmDemod kernels, 8192-sample seams
    K = 64 * 240 * 4 * 3 + 77
    nin = 8 * (K - 1) + 128
    u8 = torch.randint(0, 256, (2 * nin + 64,), dtype=torch.uint8, device="cuda")
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    dd = torch.empty(2 * K, device="cuda")
    dec.run_u8(u8.data_ptr(), 0, dd.data_ptr(), 0, K, B)
    yref = torch.empty(K, device="cuda")
    L.check(L.lib.sdrhip_fm_demod_run(None, dd.data_ptr(), 0, yref.data_ptr(), 0, K, 0.0, 0.0))
    torch.cuda.synchronize()
    h = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    hs = torch.from_numpy((h * np.float32(1.0 / 128.0)).astype(np.float32)).cuda()
    hp = torch.from_numpy(h).cuda()
    yy = torch.zeros(K, device="cuda")
    took = lab.sdrlab_decimate_demod_systolic(u8.data_ptr(), 0, 0, K, 0, hs.data_ptr(), hp.data_ptr(), 128, 1, B, yy.data_ptr())
    torch.cuda.synchronize()
    ok &= took == 1 and same(yy, yref, f"fmDemod in the systolic decimator's epilogue (round 4), {K} outputs, 8192-sample seams")
    print("ALL OK" if ok else "SOMETHING DIFFERS")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
