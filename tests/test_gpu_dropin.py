"""GPU parity, layer (1): every drop-in symbol against the CPU oracle, bit for bit.

Mirrors the reference's QuickCheck differential properties (tests/TestSuite.hs:32-50:
every variant on the same random input must agree) with the oracle as the first
element -- but with bit equality instead of the reference's 0.01 tolerance, because
each variant keeps its own summation order.
"""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle.oracle import duplicate
import signals as S

pytestmark = pytest.mark.gpu

SIZES = [1024, 8192]
FACTORS = [1, 2, 3, 5, 7, 8, 11, 13, 17, 23]   # TestSuite.hs:57 (+ the FM chain's 8)


def test_convert_u8_all_bytes(hip, oracle):
    u8 = np.arange(256, dtype=np.uint8).repeat(3)
    for sym in ("convertC", "convertCSSE", "convertCAVX"):
        assert_bit_equal(hip.DropIn.convert(sym, u8), oracle.convert_u8(u8), sym)


@pytest.mark.parametrize("n", [1, 7, 15, 16, 17, 255, 4096, 16384, 16391])
def test_convert_u8_ragged(hip, oracle, n):
    u8 = S.iq_u8(n)[:n]
    assert_bit_equal(hip.DropIn.convert("convertCAVX", u8), oracle.convert_u8(u8), f"convertCAVX n={n}")


def test_convert_bladerf_and_scale(hip, oracle):
    rng = np.random.default_rng(5)
    i16 = rng.integers(-2048, 2048, 5000, dtype=np.int16)
    for sym in ("convertCBladeRF", "convertCSSEBladeRF", "convertCAVXBladeRF"):
        assert_bit_equal(hip.DropIn.convert_i16(sym, i16), oracle.convert_i16(i16), sym)
    x = S.real_block(5001)
    for sym in ("scale", "scaleSSE", "scaleAVX"):
        assert_bit_equal(hip.DropIn.scale(sym, 0.2, x), oracle.scale(0.2, x), sym)
    tx = hip.DropIn.convert_tx(x)
    exp = (np.floor((x + np.float32(1)) * np.float32(2048)).astype(np.int32) - 2048).clip(-2048, 2047).astype(np.int16)
    assert np.array_equal(tx, exp)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("ntaps", [32, 128, 512])
def test_filters_real(hip, oracle, n, ntaps):
    x = S.real_block(n, lo=-10, hi=10)
    h = np.random.default_rng(ntaps).uniform(-10, 10, ntaps).astype(np.float32)
    num = n - ntaps + 1
    for L, sym in ((1, "filterRR"), (4, "filterSSERR"), (8, "filterAVXRR")):
        assert_bit_equal(hip.DropIn.filt(sym, num, h, x), oracle.filter_rr(L, num, h, x), sym)
    half = h[: ntaps // 2]
    for L, sym in ((4, "filterSSESymmetricRR"), (8, "filterAVXSymmetricRR")):
        assert_bit_equal(hip.DropIn.filt(sym, num, half, x), oracle.filter_sym_rr(L, num, half, x), sym)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("ntaps", [32, 128])
def test_filters_complex(hip, oracle, n, ntaps):
    x = S.cfloat_block(n, lo=-10, hi=10)
    h = np.random.default_rng(ntaps + 1).uniform(-10, 10, ntaps).astype(np.float32)
    num = n - ntaps + 1
    hd = duplicate(h)
    assert_bit_equal(hip.DropIn.filt("filterRC", num, h, x, True), oracle.filter_rc(1, num, h, x), "filterRC")
    for CL, sym in ((2, "filterSSERC"), (4, "filterAVXRC")):
        assert_bit_equal(hip.DropIn.filt(sym, num, hd, x, True), oracle.filter_rc(CL, num, hd, x), sym)
    for CL, sym in ((2, "filterSSERC2"), (4, "filterAVXRC2")):
        assert_bit_equal(hip.DropIn.filt(sym, num, h, x, True), oracle.decimate_rc2(CL, num, 1, h, x), sym)
    half = h[: ntaps // 2]
    for CL, sym in ((2, "filterSSESymmetricRC"), (4, "filterAVXSymmetricRC")):
        assert_bit_equal(hip.DropIn.filt(sym, num, half, x, True), oracle.decimate_sym_rc(CL, num, 1, half, x), sym)


@pytest.mark.parametrize("factor", FACTORS)
def test_decimators_real(hip, oracle, factor):
    n, ntaps = 8192, 128
    x = S.real_block(n, lo=-10, hi=10)
    h = np.random.default_rng(factor).uniform(-10, 10, ntaps).astype(np.float32)
    num = (n - ntaps) // factor + 1
    for L, sym in ((1, "decimateRR"), (4, "decimateSSERR"), (8, "decimateAVXRR")):
        assert_bit_equal(hip.DropIn.decim(sym, num, factor, h, x), oracle.decimate_rr(L, num, factor, h, x), sym)
    half = h[:64]
    for L, sym in ((4, "decimateSSESymmetricRR"), (8, "decimateAVXSymmetricRR")):
        assert_bit_equal(hip.DropIn.decim(sym, num, factor, half, x), oracle.decimate_sym_rr(L, num, factor, half, x), sym)


@pytest.mark.parametrize("factor", FACTORS)
@pytest.mark.parametrize("ntaps", [64, 128])
def test_decimators_complex(hip, oracle, factor, ntaps):
    n = 8192
    x = S.cfloat_block(n, lo=-10, hi=10)
    h = np.random.default_rng(factor * 7 + ntaps).uniform(-10, 10, ntaps).astype(np.float32)
    num = (n - ntaps) // factor + 1
    hd = duplicate(h)
    assert_bit_equal(hip.DropIn.decim("decimateRC", num, factor, h, x, True), oracle.decimate_rc(1, num, factor, h, x), "decimateRC")
    for CL, sym in ((2, "decimateSSERC"), (4, "decimateAVXRC")):
        assert_bit_equal(hip.DropIn.decim(sym, num, factor, hd, x, True), oracle.decimate_rc(CL, num, factor, hd, x), sym)
    for CL, sym in ((2, "decimateSSERC2"), (4, "decimateAVXRC2")):
        assert_bit_equal(hip.DropIn.decim(sym, num, factor, h, x, True), oracle.decimate_rc2(CL, num, factor, h, x), sym)
    half = h[: ntaps // 2]
    for CL, sym in ((2, "decimateSSESymmetricRC"), (4, "decimateAVXSymmetricRC")):
        assert_bit_equal(hip.DropIn.decim(sym, num, factor, half, x, True), oracle.decimate_sym_rc(CL, num, factor, half, x), sym)


def test_decimate_avxrc_config2(hip, oracle, ref):
    """BASELINE configs[1]: 127 taps -> 128, /8, one 8192-sample block -> 1009 outputs;
    also against the reference's own compiled decimateAVXRC when oracle/_ref exists."""
    x = S.cfloat_block(8192)
    h = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    hd = duplicate(h)
    got = hip.DropIn.decim("decimateAVXRC", 1009, 8, hd, x, True)
    assert_bit_equal(got, oracle.decimate_rc(4, 1009, 8, hd, x), "vs oracle")
    assert_bit_equal(got, ref.decim("decimateAVXRC", 1009, 8, hd, x, True), "vs reference build")


def test_decimate_avxrc_non_duplicated_taps(hip, oracle):
    """The 'duplicated' array is honoured as passed (re uses even, im uses odd entries)."""
    x = S.cfloat_block(2048)
    c = S.gauss_taps(256, 9)  # 128 complex taps with DIFFERENT re/im coefficients
    num = (2048 - 128) // 8 + 1
    assert_bit_equal(hip.DropIn.decim("decimateAVXRC", num, 8, c, x, True), oracle.decimate_rc(4, num, 8, c, x), "odd taps")


@pytest.mark.parametrize("I,D", [(3, 10), (2, 3), (1, 2), (5, 7), (7, 11), (3, 23), (13, 17), (2, 4)])
@pytest.mark.parametrize("ntaps", [32, 191, 512])
def test_resamplers(hip, oracle, I, D, ntaps):
    n = 8192
    h = np.random.default_rng(I * 100 + D + ntaps).uniform(-10, 10, ntaps).astype(np.float32)
    x = S.real_block(n, lo=-10, hi=10)
    xc = S.cfloat_block(n, lo=-10, hi=10)
    for L, sym, csym, CL in ((1, "resample2RR", "resample2RC", 1), (4, "resampleSSERR", "resampleSSERC", 2),
                             (8, "resampleAVXRR", "resampleAVXRC", 4)):
        prep = oracle.prepare_coeffs(L, I, D, h)
        ng = prep["num_groups"]
        period = int(prep["increments"].sum())
        count = ((n - prep["padded_len"] - period) // period) * ng
        for start in sorted({0, ng - 1}):
            a, ga = oracle.resample_rr(L, count, prep, start, x)
            b, gb = hip.DropIn.resample(sym, count, prep["num_coeffs"], start, prep["increments"], prep["groups"], x)
            assert_bit_equal(b, a, f"{sym} start={start}")
            assert ga == gb
            a, ga = oracle.resample_rc(CL, count, prep, start, xc)
            b, gb = hip.DropIn.resample(csym, count, prep["num_coeffs"], start, prep["increments"], prep["groups"], xc, True)
            assert_bit_equal(b, a, f"{csym} start={start}")
            assert ga == gb
    cnt = (n * I - ntaps) // D - 2
    for fo in range(min(I, 3)):
        assert_bit_equal(hip.DropIn.resample_legacy(cnt, I, D, fo, h, x), oracle.resample_legacy_rr(cnt, I, D, fo, h, x), "resampleRR")


def test_resample_config4(hip, oracle, ref):
    """BASELINE configs[3]: 3/10, 191 taps, one 65536-float block -> 19642 outputs, end group 1."""
    h = S.taps_resamp191()
    x = S.real_block(65536)
    prep = oracle.prepare_coeffs(8, 3, 10, h)
    assert list(prep["increments"]) == [4, 3, 3] and list(prep["offsets"]) == [0, 2, 1]
    got, g = hip.DropIn.resample("resampleAVXRR", 19642, prep["num_coeffs"], 0, prep["increments"], prep["groups"], x)
    exp, ge = ref.resample("resampleAVXRR", 19642, prep, 0, x)
    assert g == ge == 1
    assert_bit_equal(got, exp, "resampleAVXRR vs reference build")


def test_fm_demod(hip, oracle):
    rng = np.random.default_rng(77 + SWEEP_SEED)
    x = rng.uniform(-1, 1, 2 * 100000).astype(np.float32)
    # special values: zeros, negative zeros, axes, repeated samples, denormals, huge ratios
    sp = np.array([0, 0, -0.0, 0, 0, -0.0, -0.0, -0.0, 1, 0, -1, 0, 0, 1, 0, -1, -1, -0.0, 1e-40, 1e-40,
                   1e-30, 1, 1, 1e-30, 1e30, 1e-8, -1e-8, 1e30, 0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 3, -4],
                  np.float32)
    x[: sp.size] = sp
    assert_bit_equal(hip.DropIn.fm_demod(x), oracle.fm_demod(x), "fmDemodF")
    assert_bit_equal(hip.DropIn.fm_demod(x, (0.3, -0.7)), oracle.fm_demod(x, (0.3, -0.7)), "fmDemodF carried")
    iq = oracle.convert_u8(S.iq_u8_fm(50000))
    assert_bit_equal(hip.DropIn.fm_demod(iq), oracle.fm_demod(iq), "fmDemodF fm signal")


def test_dc_blocker(hip, ref):
    x = S.real_block(4096)
    out, fs, fo = hip.DropIn.dc_blocker(x, 0.25, -0.5)
    import ctypes as C
    exp = np.empty_like(x)
    efs, efo = C.c_float(), C.c_float()
    ref.lib.dcBlocker.argtypes = [C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]
    ref.lib.dcBlocker(x.size, 0.25, -0.5, C.byref(efs), C.byref(efo), x.ctypes.data_as(C.POINTER(C.c_float)),
                      exp.ctypes.data_as(C.POINTER(C.c_float)))
    assert_bit_equal(out, exp, "dcBlocker")
    assert fs == efs.value and fo == efo.value


def test_empty_and_tiny_calls(hip, oracle):
    """num = 0 is a no-op for every family (the reference's loops simply do not execute); the smallest
    non-empty calls work."""
    x = S.real_block(256)
    h = S.gauss_taps(128, 4)
    assert hip.DropIn.filt("filterAVXRR", 0, h, x).size == 0
    assert hip.DropIn.decim("decimateAVXRC", 0, 8, duplicate(h), S.cfloat_block(256), True).size == 0
    assert hip.DropIn.convert("convertCAVX", np.zeros(0, np.uint8)).size == 0
    assert hip.DropIn.fm_demod(np.zeros(0, np.float32)).size == 0
    prep = oracle.prepare_coeffs(8, 3, 10, S.taps_resamp191())
    out, g = hip.DropIn.resample("resampleAVXRR", 0, prep["num_coeffs"], 2, prep["increments"], prep["groups"], x)
    assert out.size == 0 and g == 2
    assert_bit_equal(hip.DropIn.filt("filterAVXRR", 1, h, x[:128]), oracle.filter_rr(8, 1, h, x[:128]), "one output")
    xc = S.cfloat_block(128)
    assert_bit_equal(hip.DropIn.decim("decimateAVXRC", 1, 8, duplicate(h), xc, True), oracle.decimate_rc(4, 1, 8, duplicate(h), xc), "one output")


def test_reference_test_suite_maximum_size(hip, oracle):
    """The largest size of the reference's generators (65536, TestSuite.hs:55) with its largest taps (1024)."""
    n, ntaps = 65536, 1024
    rng = np.random.default_rng(99)
    x = rng.uniform(-10, 10, n).astype(np.float32)
    xc = rng.uniform(-10, 10, 2 * n).astype(np.float32)
    h = rng.uniform(-10, 10, ntaps).astype(np.float32)
    num = n - ntaps + 1
    assert_bit_equal(hip.DropIn.filt("filterAVXRR", num, h, x), oracle.filter_rr(8, num, h, x), "filterAVXRR 65536/1024")
    assert_bit_equal(hip.DropIn.filt("filterAVXSymmetricRR", num, h[:512], x), oracle.filter_sym_rr(8, num, h[:512], x), "sym")
    nd = (n - ntaps) // 23 + 1
    assert_bit_equal(hip.DropIn.decim("decimateAVXRC", nd, 23, duplicate(h), xc, True), oracle.decimate_rc(4, nd, 23, duplicate(h), xc), "decimateAVXRC /23")


SWEEP_SCALE = max(1, int(__import__("os").environ.get("SDRHIP_SWEEP_SCALE", "1")))
SWEEP_SEED = int(__import__("os").environ.get("SDRHIP_SWEEP_SEED", "0"))        # other seeds for soak runs


def test_dropin_random_sweep_against_the_reference_build(hip, oracle, ref):
    """Every drop-in filter / decimator / resampler symbol with seeded random shapes, straight against the reference's
    own compiled C (oracle/_ref) -- no restatement in between."""
    rng = np.random.default_rng(31337 + SWEEP_SEED)
    fir = [  # (symbol, complex, taps multiple, duplicated, symmetric)
        ("RR", False, 1, False, False), ("SSERR", False, 4, False, False), ("AVXRR", False, 8, False, False),
        ("SSESymmetricRR", False, 4, False, True), ("AVXSymmetricRR", False, 8, False, True),
        ("RC", True, 1, False, False), ("SSERC", True, 2, True, False), ("AVXRC", True, 4, True, False),
        ("SSERC2", True, 4, False, False), ("AVXRC2", True, 8, False, False),
        ("SSESymmetricRC", True, 4, False, True), ("AVXSymmetricRC", True, 8, False, True)]
    res = [("resample2RR", False, 1), ("resampleSSERR", False, 4), ("resampleAVXRR", False, 8),
           ("resample2RC", True, 1), ("resampleSSERC", True, 4), ("resampleAVXRC", True, 8)]
    done = 0
    for trial in range(120 * SWEEP_SCALE):
        if rng.integers(0, 4) < 3:
            name, cplx, mult, dup, sym = fir[rng.integers(0, len(fir))]
            decimate = bool(rng.integers(0, 2))
            factor = int(rng.integers(1, 13)) if decimate else 1
            nt = mult * int(rng.integers(1, 1 + 256 // mult))          # taps as the kernel walks them (half taps when sym)
            full = 2 * nt if sym else nt
            num = int(rng.integers(1, 3000))
            n_in = (num - 1) * factor + full
            w = 2 if cplx else 1
            x = rng.uniform(-10, 10, w * n_in).astype(np.float32)
            h = rng.uniform(-10, 10, nt).astype(np.float32)
            passed = duplicate(h) if dup else h
            symbol = ("decimate" if decimate else "filter") + name
            if decimate:
                got = hip.DropIn.decim(symbol, num, factor, passed, x, cplx)
                exp = ref.decim(symbol, num, factor, passed, x, cplx)
            else:
                got = hip.DropIn.filt(symbol, num, passed, x, cplx)
                exp = ref.filt(symbol, num, passed, x, cplx)
            assert_bit_equal(got, exp, f"trial {trial}: {symbol} num={num} factor={factor} taps={nt}")
        else:
            symbol, cplx, simd = res[rng.integers(0, len(res))]
            while True:
                I, D = int(rng.integers(1, 9)), int(rng.integers(2, 30))
                if D > I and np.gcd(I, D) == 1:
                    break
            ntaps = int(rng.integers(I, 60 * I))
            h = rng.uniform(-10, 10, ntaps).astype(np.float32)
            prep = oracle.prepare_coeffs(simd, I, D, h)
            num = int(rng.integers(1, 2000))
            g0 = int(rng.integers(0, prep["num_groups"]))
            n_in = num * (D // I + 2) + prep["groups"].shape[1] + 16
            w = 2 if cplx else 1
            x = rng.uniform(-10, 10, w * n_in).astype(np.float32)
            got, g = hip.DropIn.resample(symbol, num, prep["num_coeffs"], g0, prep["increments"], prep["groups"], x, cplx)
            exp, ge = ref.resample(symbol, num, prep, g0, x, cplx)
            assert g == ge, f"trial {trial}: {symbol} end group {g} vs {ge}"
            assert_bit_equal(got, exp, f"trial {trial}: {symbol} {I}/{D} taps={ntaps} num={num} g0={g0}")
        done += 1
    assert done == 120 * SWEEP_SCALE


def test_dropin_calls_from_several_threads(hip, oracle):
    """Each call leases its own scratch context (stream, staging, tap cache: abi_dropin.cpp), so pipeline threads do not
    share state: four threads hammer different symbols with different taps at once and every result is still exact."""
    import threading
    x = S.cfloat_block(8192)
    xr = S.real_block(8192)
    jobs = []
    for t in range(4):
        taps = S.gauss_taps(64 + 16 * t, 40 + t)
        hd = duplicate(taps)
        num_c = (8192 - taps.size) // 8 + 1
        num_r = 8192 - taps.size + 1
        jobs.append((("decimateAVXRC", num_c, 8, hd, x, True), oracle.decimate_rc(4, num_c, 8, hd, x),
                     ("filterAVXRR", num_r, taps, xr), oracle.filter_rr(8, num_r, taps, xr)))
    errors = []

    def work(job):
        dargs, dexp, fargs, fexp = job
        try:
            for _ in range(25):
                assert_bit_equal(hip.DropIn.decim(*dargs), dexp, "decimateAVXRC from a thread")
                assert_bit_equal(hip.DropIn.filt(*fargs), fexp, "filterAVXRR from a thread")
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]
