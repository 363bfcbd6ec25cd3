"""The record seam of the C ABI (sdrhip_{filter,decimator,resampler}_{one,cross}, abi_records.cpp): the closures a Haskell
constructor puts into the reference's Filter / Decimator / Resampler records (Filter.hs:116-144), on host vectors --
against the restated C kernels (One) and the restated Haskell cross kernels (Cross, FilterInternal.hs:397-423)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle.oracle import duplicate
import signals as S

pytestmark = pytest.mark.gpu


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pad(taps, m):
    t = np.asarray(taps, np.float32)
    return np.concatenate([t, np.zeros((-t.size) % m, np.float32)])


def test_decimator_record_complex(hip, oracle):
    L = hip
    taps = S.taps_decim127()
    dec = L.Decimator(8, taps, L.ORDER_AVX, complex_=True)
    Lp = dec.num_coeffs
    assert Lp == 128
    rng = np.random.default_rng(5)
    B = 8192
    last = (rng.random(2 * B, dtype=np.float32) * 2 - 1)
    nxt = (rng.random(2 * B, dtype=np.float32) * 2 - 1)
    # decimateOne: (B - Lp)/8 + 1 outputs of one buffer, the AVX order of decimateAVXRC
    num = (B - Lp) // 8 + 1
    out = np.empty(2 * num, np.float32)
    L.check(L.lib.sdrhip_decimator_one(dec.h, num, _fp(last), _fp(out)), "sdrhip_decimator_one")
    assert_bit_equal(out, oracle.decimate_rc(4, num, 8, duplicate(_pad(taps, 4)), last), "decimateOne")
    # decimateCross: the 15 outputs straddling last | next, starting at the first window that no longer fits `last`
    used = num * 8
    tail = last[2 * used:]
    ncross = (B - used + 7) // 8
    assert ncross == 15
    outc = np.empty(2 * ncross, np.float32)
    L.check(L.lib.sdrhip_decimator_cross(dec.h, ncross, _fp(tail), tail.size // 2, _fp(nxt), nxt.size // 2, _fp(outc)), "sdrhip_decimator_cross")
    assert_bit_equal(outc, oracle.decimate_cross_c(8, _pad(taps, 4), ncross, tail, nxt), "decimateCross")
    # too little data for the requested outputs is refused, not padded
    assert L.lib.sdrhip_decimator_cross(dec.h, ncross, _fp(tail), tail.size // 2, _fp(nxt), 10, _fp(outc)) < 0


@pytest.mark.parametrize("sym", [True, False])
def test_filter_record_real(hip, oracle, sym):
    L = hip
    half = S.taps_audio_half64()
    full = np.concatenate([half, half[::-1]]).astype(np.float32)
    f = L.Filter(half, L.ORDER_AVX, sym=True) if sym else L.Filter(full, L.ORDER_AVX)
    Lp = f.num_coeffs
    assert Lp == 128
    rng = np.random.default_rng(6)
    B = 8192
    last = rng.standard_normal(B).astype(np.float32)
    nxt = rng.standard_normal(B).astype(np.float32)
    num = B - Lp + 1
    out = np.empty(num, np.float32)
    L.check(L.lib.sdrhip_filter_one(f.h, num, _fp(last), _fp(out)), "sdrhip_filter_one")
    exp = oracle.filter_sym_rr(8, num, half, last) if sym else oracle.filter_rr(8, num, full, last)
    assert_bit_equal(out, exp, "filterOne")
    tail = np.ascontiguousarray(last[num:])
    ncross = Lp - 1
    outc = np.empty(ncross, np.float32)
    L.check(L.lib.sdrhip_filter_cross(f.h, ncross, _fp(tail), tail.size, _fp(nxt), nxt.size, _fp(outc)), "sdrhip_filter_cross")
    assert_bit_equal(outc, oracle.decimate_cross_r(1, full, ncross, tail, nxt), "filterCross")


def test_resampler_record_walks_like_the_reference_pipe(hip, oracle):
    """resampleOne / resampleCross chained exactly as firResampler chains them (Filter.hs:683-727) over three buffers,
    carrying (group, offset) the way mkResampler does (Filter.hs:408-425)."""
    L = hip
    I, D = 3, 10
    taps = S.taps_resamp191()
    r = L.Resampler(I, D, taps, L.ORDER_AVX)
    Lp = r.num_coeffs
    assert Lp == 192
    prep = oracle.prepare_coeffs(8, I, D, taps)
    rng = np.random.default_rng(7)
    B = 8192
    bufs = [rng.standard_normal(B).astype(np.float32) for _ in range(3)]
    group, off = 0, 0
    buf = bufs[0]
    for nb in (1, 2):
        # simple: as many outputs as fit the current buffer
        count = (buf.size * I - Lp + off) // D + 1
        out = np.empty(count, np.float32)
        g2 = L.lib.sdrhip_resampler_one(r.h, group, count, _fp(buf), buf.size, _fp(out))
        assert g2 >= 0, L.lib.sdrhip_last_error()
        exp, eg = oracle.resample_rr(8, count, prep, group, buf)
        assert_bit_equal(out, exp, f"resampleOne, buffer {nb}")
        assert g2 == eg
        group = g2
        end_off = I - 1 - ((I + group * D - 1) % I)
        used = -(-(count * D - off) // I)
        rest = np.ascontiguousarray(buf[used:])
        off = end_off
        nxt = bufs[nb]
        # crossover: the outputs whose window starts in `rest`
        ccount = -(-(rest.size * I + off) // D)
        outc = np.empty(ccount, np.float32)
        o2 = L.lib.sdrhip_resampler_cross(r.h, off, ccount, _fp(rest), rest.size, _fp(nxt), nxt.size, _fp(outc))
        assert o2 >= 0, L.lib.sdrhip_last_error()
        expc, eo = oracle.resample_cross_r(I, D, taps, off, ccount, rest, nxt)
        assert_bit_equal(outc, expc, f"resampleCross, boundary {nb}")
        assert o2 == eo
        group = (group + ccount) % I
        usedc = -(-(ccount * D - off) // I)
        off = o2
        buf = np.ascontiguousarray(nxt[usedc - rest.size:])
