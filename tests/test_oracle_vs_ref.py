"""The CPU restatement (oracle/sdr_oracle.c) against the reference's own compiled C
(oracle/_ref, built by oracle/Makefile from /root/reference/c_sources): every exported
kernel, bit for bit.  Mirrors the structure of the reference's differential QuickCheck
properties (tests/TestSuite.hs:32-50) -- all variants on the same random input -- with
its generators' ranges (sizes, tap counts, factors, [-10,10] data: TestSuite.hs:55-64)."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle.oracle import duplicate
import signals as S

SIZES = [1024, 4096, 16384]
NCOEFF = [32, 64, 128, 256, 512]
FACTORS = [1, 2, 3, 5, 7, 11, 13, 17, 23]


def test_convert(oracle, ref):
    u8 = np.arange(256, dtype=np.uint8).repeat(2)
    for sym in ("convertC", "convertCSSE", "convertCAVX"):
        assert_bit_equal(oracle.convert_u8(u8), ref.convert(sym, u8), sym)
    i16 = np.random.default_rng(1).integers(-2048, 2048, 4096, dtype=np.int16)
    import ctypes as C
    for sym in ("convertCBladeRF", "convertCSSEBladeRF", "convertCAVXBladeRF"):
        out = np.empty(4096 + 8, np.float32)
        buf = np.zeros(4096 + 16, np.int16)
        buf[:4096] = i16
        getattr(ref.lib, sym)(C.c_int(4096), buf.ctypes.data_as(C.POINTER(C.c_int16)), out.ctypes.data_as(C.POINTER(C.c_float)))
        assert_bit_equal(oracle.convert_i16(i16), out[:4096], sym)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("ntaps", NCOEFF)
def test_filters(oracle, ref, n, ntaps):
    rng = np.random.default_rng(n + ntaps)
    x = rng.uniform(-10, 10, n).astype(np.float32)
    xc = rng.uniform(-10, 10, 2 * n).astype(np.float32)
    h = rng.uniform(-10, 10, ntaps).astype(np.float32)
    half, hd = h[: ntaps // 2], duplicate(h)
    num = n - ntaps + 1
    for L, sym in ((1, "filterRR"), (4, "filterSSERR"), (8, "filterAVXRR")):
        assert_bit_equal(oracle.filter_rr(L, num, h, x), ref.filt(sym, num, h, x), sym)
    for L, sym in ((4, "filterSSESymmetricRR"), (8, "filterAVXSymmetricRR")):
        assert_bit_equal(oracle.filter_sym_rr(L, num, half, x), ref.filt(sym, num, half, x), sym)
    assert_bit_equal(oracle.filter_rc(1, num, h, xc), ref.filt("filterRC", num, h, xc, True), "filterRC")
    for CL, sym in ((2, "filterSSERC"), (4, "filterAVXRC")):
        assert_bit_equal(oracle.filter_rc(CL, num, hd, xc), ref.filt(sym, num, hd, xc, True), sym)
    for CL, sym in ((2, "filterSSERC2"), (4, "filterAVXRC2")):
        assert_bit_equal(oracle.decimate_rc2(CL, num, 1, h, xc), ref.filt(sym, num, h, xc, True), sym)
    for CL, sym in ((2, "filterSSESymmetricRC"), (4, "filterAVXSymmetricRC")):
        assert_bit_equal(oracle.decimate_sym_rc(CL, num, 1, half, xc), ref.filt(sym, num, half, xc, True), sym)


@pytest.mark.parametrize("factor", FACTORS)
@pytest.mark.parametrize("ntaps", [32, 128])
def test_decimators(oracle, ref, factor, ntaps):
    n = 8192
    rng = np.random.default_rng(factor * 31 + ntaps)
    x = rng.uniform(-10, 10, n).astype(np.float32)
    xc = rng.uniform(-10, 10, 2 * n).astype(np.float32)
    h = rng.uniform(-10, 10, ntaps).astype(np.float32)
    half, hd = h[: ntaps // 2], duplicate(h)
    num = (n - ntaps) // factor + 1
    for L, sym in ((1, "decimateRR"), (4, "decimateSSERR"), (8, "decimateAVXRR")):
        assert_bit_equal(oracle.decimate_rr(L, num, factor, h, x), ref.decim(sym, num, factor, h, x), sym)
    for L, sym in ((4, "decimateSSESymmetricRR"), (8, "decimateAVXSymmetricRR")):
        assert_bit_equal(oracle.decimate_sym_rr(L, num, factor, half, x), ref.decim(sym, num, factor, half, x), sym)
    assert_bit_equal(oracle.decimate_rc(1, num, factor, h, xc), ref.decim("decimateRC", num, factor, h, xc, True), "decimateRC")
    for CL, sym in ((2, "decimateSSERC"), (4, "decimateAVXRC")):
        assert_bit_equal(oracle.decimate_rc(CL, num, factor, hd, xc), ref.decim(sym, num, factor, hd, xc, True), sym)
    for CL, sym in ((2, "decimateSSERC2"), (4, "decimateAVXRC2")):
        assert_bit_equal(oracle.decimate_rc2(CL, num, factor, h, xc), ref.decim(sym, num, factor, h, xc, True), sym)
    for CL, sym in ((2, "decimateSSESymmetricRC"), (4, "decimateAVXSymmetricRC")):
        assert_bit_equal(oracle.decimate_sym_rc(CL, num, factor, half, xc), ref.decim(sym, num, factor, half, xc, True), sym)


@pytest.mark.parametrize("I,D", [(1, 2), (2, 3), (3, 10), (3, 5), (5, 7), (7, 11), (11, 13), (13, 17), (17, 23), (2, 23)])
@pytest.mark.parametrize("ntaps", [32, 191, 512])
def test_resamplers(oracle, ref, I, D, ntaps):
    """TestSuite.hs:170-194: interpolation < decimation from the factor list, random starting group."""
    n = 8192
    rng = np.random.default_rng(I * 101 + D + ntaps)
    x = rng.uniform(-10, 10, n).astype(np.float32)
    xc = rng.uniform(-10, 10, 2 * n).astype(np.float32)
    h = rng.uniform(-10, 10, ntaps).astype(np.float32)
    for L, sym, CL, csym in ((1, "resample2RR", 1, "resample2RC"), (4, "resampleSSERR", 2, "resampleSSERC"),
                             (8, "resampleAVXRR", 4, "resampleAVXRC")):
        prep = oracle.prepare_coeffs(L, I, D, h)
        ng, period = prep["num_groups"], int(prep["increments"].sum())
        count = ((n - prep["padded_len"] - period) // period) * ng
        start = int(rng.integers(0, ng))
        a, ga = oracle.resample_rr(L, count, prep, start, x)
        b, gb = ref.resample(sym, count, prep, start, x)
        assert_bit_equal(a, b, sym)
        assert ga == gb
        a, ga = oracle.resample_rc(CL, count, prep, start, xc)
        b, gb = ref.resample(csym, count, prep, start, xc, True)
        assert_bit_equal(a, b, csym)
        assert ga == gb
    cnt = (n * I - ntaps) // D - 2
    assert_bit_equal(oracle.resample_legacy_rr(cnt, I, D, 0, h, x), ref.resample_legacy(cnt, I, D, 0, h, x), "resampleRR")


def test_sequential_order_is_scalar_c(oracle, ref):
    """The Haskell cross-buffer kernels sum left to right (VG.sum); so do the scalar C
    kernels -- which makes the scalar C symbols executable oracles for the Cross
    arithmetic on a concatenated window."""
    rng = np.random.default_rng(3)
    last = rng.uniform(-1, 1, 2 * 120).astype(np.float32)
    nxt = rng.uniform(-1, 1, 2 * 8192).astype(np.float32)
    h = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    got = oracle.decimate_cross_c(8, h, 15, last, nxt)
    cat = np.concatenate([last, nxt])
    assert_bit_equal(got, ref.decim("decimateRC", 15, 8, h, cat, True), "decimateCross == decimateRC on the concatenation")
    lr, nr = last[::2].copy(), nxt[::2].copy()
    got = oracle.decimate_cross_r(1, h, 120, lr, nr)
    assert_bit_equal(got, ref.filt("filterRR", 120, h, np.concatenate([lr, nr])), "filterCross == filterRR")
    h191 = S.taps_resamp191()
    got, off = oracle.resample_cross_r(3, 10, h191, 2, 19, lr[:57], nr)
    exp = ref.resample_legacy(19, 3, 10, 2, h191, np.concatenate([lr[:57], nr]))
    assert_bit_equal(got, exp, "resampleCross == legacy resampleRR")


def test_avx_order_differs_from_sequential(oracle):
    """Why 1 ULP needs the lane order: the orders really give different bits."""
    x = S.cfloat_block(8192)
    h = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    a = oracle.decimate_rc(4, 1009, 8, duplicate(h), x)
    b = oracle.decimate_rc(1, 1009, 8, h, x)
    assert (a.view(np.uint32) != b.view(np.uint32)).mean() > 0.3
    assert np.abs(a - b).max() < 1e-5
