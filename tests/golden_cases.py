"""The golden cases: one table, three providers.

    provider "ref"    -> the reference's own compiled C (oracle/_ref)      [generates the fixtures]
    provider "oracle" -> our CPU restatement                                [CPU test]
    provider "hip"    -> the product's drop-in symbols on the GPU           [GPU test]

Inputs are regenerated from seeds; tests/golden/reference_outputs.npz stores the
reference's outputs (as uint32 bit patterns) and a CRC of each case's inputs.
"""
import zlib

import numpy as np

import signals as S
from oracle.oracle import duplicate


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


class Providers:
    def __init__(self, kind, orc, ref=None, hip=None):
        self.kind, self.orc, self.ref, self.hip = kind, orc, ref, hip

    # each method: reference symbol name + the arguments the FFI wrapper would pass
    def convert(self, sym, u8):
        if self.kind == "ref":
            return self.ref.convert(sym, u8)
        if self.kind == "hip":
            return self.hip.DropIn.convert(sym, u8)
        return self.orc.convert_u8(u8)

    def filt(self, sym, num, c, x, cplx=False):
        if self.kind == "ref":
            return self.ref.filt(sym, num, c, x, cplx)
        if self.kind == "hip":
            return self.hip.DropIn.filt(sym, num, c, x, cplx)
        o = self.orc
        return {
            "filterRR": lambda: o.filter_rr(1, num, c, x),
            "filterSSERR": lambda: o.filter_rr(4, num, c, x),
            "filterAVXRR": lambda: o.filter_rr(8, num, c, x),
            "filterSSESymmetricRR": lambda: o.filter_sym_rr(4, num, c, x),
            "filterAVXSymmetricRR": lambda: o.filter_sym_rr(8, num, c, x),
            "filterAVXSymmetricRC": lambda: o.decimate_sym_rc(4, num, 1, c, x),
        }[sym]()

    def decim(self, sym, num, factor, c, x, cplx=False):
        if self.kind == "ref":
            return self.ref.decim(sym, num, factor, c, x, cplx)
        if self.kind == "hip":
            return self.hip.DropIn.decim(sym, num, factor, c, x, cplx)
        o = self.orc
        return {
            "decimateRC": lambda: o.decimate_rc(1, num, factor, c, x),
            "decimateSSERC": lambda: o.decimate_rc(2, num, factor, c, x),
            "decimateAVXRC": lambda: o.decimate_rc(4, num, factor, c, x),
            "decimateAVXRC2": lambda: o.decimate_rc2(4, num, factor, c, x),
            "decimateAVXRR": lambda: o.decimate_rr(8, num, factor, c, x),
        }[sym]()

    def resample(self, sym, num, prep, start, x, cplx=False):
        if self.kind == "ref":
            return self.ref.resample(sym, num, prep, start, x, cplx)
        if self.kind == "hip":
            return self.hip.DropIn.resample(sym, num, prep["num_coeffs"], start, prep["increments"], prep["groups"], x, cplx)
        o = self.orc
        return {
            "resample2RR": lambda: o.resample_rr(1, num, prep, start, x),
            "resampleAVXRR": lambda: o.resample_rr(8, num, prep, start, x),
            "resampleAVXRC": lambda: o.resample_rc(4, num, prep, start, x),
        }[sym]()

    def resample_legacy(self, num, I, D, fo, c, x):
        if self.kind == "ref":
            return self.ref.resample_legacy(num, I, D, fo, c, x)
        if self.kind == "hip":
            return self.hip.DropIn.resample_legacy(num, I, D, fo, c, x)
        return self.orc.resample_legacy_rr(num, I, D, fo, c, x)

    def scale(self, sym, f, x):
        if self.kind == "ref":
            return self.ref.scale(sym, f, x)
        if self.kind == "hip":
            return self.hip.DropIn.scale(sym, f, x)
        return self.orc.scale(f, x)


def cases(p):
    """-> dict name -> (output float32 array, input crc)."""
    orc = p.orc
    out = {}
    # A1 convert: all byte values + a random block (convert.c:15-50)
    u8 = np.concatenate([np.arange(256, dtype=np.uint8), S.iq_u8(4096)[:4088]])
    for sym in ("convertC", "convertCSSE", "convertCAVX"):
        out[sym] = (p.convert(sym, u8), crc(u8))
    # A2 BASELINE configs[1]: decimateAVXRC, 127 -> 128 taps, /8, one 8192-sample block (decimate.c:105-113)
    x = S.cfloat_block(8192)
    h = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    hd = duplicate(h)
    out["decimateAVXRC_cfg2"] = (p.decim("decimateAVXRC", 1009, 8, hd, x, True), crc(x, hd))
    out["decimateSSERC_cfg2"] = (p.decim("decimateSSERC", 1009, 8, hd, x, True), crc(x, hd))
    out["decimateRC_cfg2"] = (p.decim("decimateRC", 1009, 8, h, x, True), crc(x, h))
    xu = orc.convert_u8(S.iq_u8(8192))
    out["decimateAVXRC_u8iq"] = (p.decim("decimateAVXRC", 1009, 8, hd, xu, True), crc(xu, hd))
    # A4 BASELINE configs[0]: filterAVXSymmetricRR, 64 half-taps, one 8192-float block (filter.c:60-68)
    xr = S.real_block(8192)
    half = S.taps_audio_half64()
    out["filterAVXSymmetricRR_cfg1"] = (p.filt("filterAVXSymmetricRR", 8065, half, xr), crc(xr, half))
    out["filterSSESymmetricRR_cfg1"] = (p.filt("filterSSESymmetricRR", 8065, half, xr), crc(xr, half))
    full = np.concatenate([half, half[::-1]])
    out["filterAVXRR_128"] = (p.filt("filterAVXRR", 8065, full, xr), crc(xr, full))
    out["filterRR_128"] = (p.filt("filterRR", 8065, full, xr), crc(xr, full))
    # A3 BASELINE configs[3]: resampleAVXRR 3/10, 191 taps, one 65536-float block (resample.c:70-87)
    x64 = S.real_block(65536)
    h191 = S.taps_resamp191()
    prep = orc.prepare_coeffs(8, 3, 10, h191)
    r, g = p.resample("resampleAVXRR", 19642, prep, 0, x64)
    out["resampleAVXRR_cfg4"] = (r, crc(x64, h191))
    out["resampleAVXRR_cfg4_endgroup"] = (np.array([g], np.float32), crc(x64))
    out["resampleAVXRR_start2"] = (p.resample("resampleAVXRR", 3000, prep, 2, x64)[0], crc(x64, h191))
    prep1 = orc.prepare_coeffs(1, 3, 10, h191)
    out["resample2RR_cfg4"] = (p.resample("resample2RR", 19642, prep1, 0, x64)[0], crc(x64, h191))
    out["resampleRR_legacy"] = (p.resample_legacy(19000, 3, 10, 0, h191, x64), crc(x64, h191))
    # stress set in the style of the reference's QuickCheck generators (TestSuite.hs:62-64): [-10,10]
    xs = S.real_block(4096, seed=7, lo=-10, hi=10)
    hs = np.random.default_rng(8).uniform(-10, 10, 64).astype(np.float32)
    out["decimateAVXRR_stress_f7"] = (p.decim("decimateAVXRR", (4096 - 64) // 7 + 1, 7, hs, xs), crc(xs, hs))
    xcs = S.cfloat_block(4096, seed=9, lo=-10, hi=10)
    out["decimateAVXRC2_stress_f3"] = (p.decim("decimateAVXRC2", (4096 - 64) // 3 + 1, 3, hs, xcs, True), crc(xcs, hs))
    out["filterAVXSymmetricRC_stress"] = (p.filt("filterAVXSymmetricRC", 4096 - 127, hs, xcs, True), crc(xcs, hs))
    prepc = orc.prepare_coeffs(8, 5, 7, hs)
    out["resampleAVXRC_5_7"] = (p.resample("resampleAVXRC", 2000, prepc, 1, xcs, True)[0], crc(xcs, hs))
    out["scaleAVX"] = (p.scale("scaleAVX", 0.2, xr), crc(xr))
    # more than 64 polyphase groups (the drop-in symbols used to abort): 97/100, 1500 taps -> 97 groups of 16
    h1500 = np.random.default_rng(18).uniform(-1, 1, 1500).astype(np.float32)
    x8k = S.real_block(8192, seed=19)
    prep97 = orc.prepare_coeffs(8, 97, 100, h1500)
    r97, g97 = p.resample("resampleAVXRR", 7000, prep97, 0, x8k)
    out["resampleAVXRR_97_100"] = (r97, crc(x8k, h1500))
    out["resampleAVXRR_97_100_endgroup"] = (np.array([g97], np.float32), crc(x8k))
    out["resampleAVXRR_97_100_start40"] = (p.resample("resampleAVXRR", 5000, prep97, 40, x8k)[0], crc(x8k, h1500))
    out["resampleRR_legacy_97_100"] = (p.resample_legacy(6000, 97, 100, 0, h1500, x8k), crc(x8k, h1500))
    out["resampleRR_legacy_97_100_off33"] = (p.resample_legacy(6000, 97, 100, 33, h1500, x8k), crc(x8k, h1500))
    xc8k = S.cfloat_block(8192, seed=20)
    out["resampleAVXRC_97_100"] = (p.resample("resampleAVXRC", 7000, prep97, 5, xc8k, True)[0], crc(xc8k, h1500))
    # The reference example's own filters (examples/fm/Coeffs.hs, tests/golden/example_taps.npz) on the three hot kernels, as
    # examples/fm/fm.hs:30-41 uses them: 51 -> 52 taps duplicated, /8; 3/10 with 31 taps (groups of 11/10/10 -> 16); 32 half-taps
    he = S.taps_example_rf_decim()
    hep = np.concatenate([he, np.zeros((-he.size) % 4, np.float32)])
    out["decimateAVXRC_example_taps"] = (p.decim("decimateAVXRC", (8192 - hep.size) // 8 + 1, 8, duplicate(hep), xu, True), crc(xu, hep))
    hr = S.taps_example_audio_resampler()
    prepe = orc.prepare_coeffs(8, 3, 10, hr)
    xe = S.real_block(8192, seed=21, lo=-3.2, hi=3.2)            # fmDemod's range
    ne = (8192 * 3 - 3 * prepe["groups"].shape[1]) // 10
    re_, ge = p.resample("resampleAVXRR", ne, prepe, 0, xe)
    out["resampleAVXRR_example_taps"] = (re_, crc(xe, hr))
    out["resampleAVXRR_example_taps_endgroup"] = (np.array([ge], np.float32), crc(xe))
    ha = S.taps_example_audio_filter_half()
    out["filterAVXSymmetricRR_example_taps"] = (p.filt("filterAVXSymmetricRR", 8192 - 63, ha, xr), crc(xr, ha))
    return out
