"""fmDemod restatement (Demod.hs:21-46 + GHC base atan2 + libm atanf).  The reference pins
nothing here (no test, no vector) and GHC cannot run in this image: these tests pin
the restatement to the image's libm and to mathematical properties.  'parity unpinned'
vs a real GHC run remains."""
import math

import numpy as np

import signals as S


SWEPT_LIBC = "2.35"        # the glibc whose atanf was compared with the model on all 2^32 inputs: 0 differences


def check_model_against_this_libm(oracle, stride=97):
    """The SPEC of fmDemod's `atan` is the fdlibm f32 model (oracle/sdr_oracle.c: orc_atanf_model; the device evaluates the same
    operations, demod.hpp).  Its agreement with the libm of the machine running the tests is a SEPARATE statement: exact on the
    glibc it was swept against, within 1 ULP on any other (a correctly-rounded atanf differs from fdlibm's in ~5 % of arguments) --
    so a different libm on the GPU box moves neither the test suite nor what `the reference` means."""
    exact = oracle.libc_version() == SWEPT_LIBC
    bad, first, worst = oracle.atanf_sweep(0, 0xFFFFFFFF, stride)
    # the count is also the number of arguments on which the DEVICE would differ from a GHC build over this host's libm: the device
    # equals the model on every float (tests/test_gpu_demod_exhaustive.py)
    print(f"atanf model vs this host's libm (glibc {oracle.libc_version()}), every {stride}th float: {bad} mismatches, worst {worst} ULP")
    if exact:
        assert bad == 0, f"glibc {SWEPT_LIBC}: first mismatch at {first:#x}, up to {worst} ULP"
    else:
        assert worst <= 1, f"glibc {oracle.libc_version()}: the model is {worst} ULP from atanf at {first:#x}"
    for edge in (0x31000000, 0x3ee00000, 0x3f300000, 0x3f980000, 0x401c0000, 0x4c000000, 0x7f800000):
        for sign in (0, 0x80000000):
            lo = (edge | sign) - 4096
            bad, first, worst = oracle.atanf_sweep(lo, lo + 8192, 1)
            assert (bad == 0) if exact else (worst <= 1), f"first mismatch at {first:#x} ({worst} ULP)"
    return exact


def test_atanf_model_matches_libm_on_a_dense_sweep(oracle):
    """(All 2^32 inputs were swept when this was written: 0 mismatches; here every 97th float plus the neighbourhoods of the
    range boundaries.  tests/test_gpu_demod_exhaustive.py runs the same statement, over every float, on the GPU box's host.)"""
    check_model_against_this_libm(oracle, 97)


def test_the_oracle_does_not_call_libm_for_fm_demod(oracle):
    """orc_fm_demod / orc_ghc_atan2f must give the model's bits whatever libm says: the arguments where a correctly-rounded atanf
    and fdlibm's differ are not rare (about one in twenty), so equality with the model on a few thousand ratios pins the wiring."""
    rng = np.random.default_rng(11)
    q = np.exp2(rng.uniform(-28, 24, 4000)).astype(np.float32) * rng.choice(np.array([-1.0, 1.0], np.float32), 4000)
    for v in q:
        a = oracle.lib.orc_ghc_atan2f(float(v), 1.0)
        m = oracle.lib.orc_atanf_model(float(v))
        assert np.float32(a).view(np.uint32) == np.float32(m).view(np.uint32)


def test_ghc_atan2_quadrants(oracle):
    f = oracle.lib.orc_ghc_atan2f
    pi = np.float32(np.pi)
    assert f(0.0, 1.0) == 0.0
    assert f(1.0, 0.0) == pi / 2
    assert f(-1.0, 0.0) == -pi / 2
    assert f(0.0, -1.0) == pi
    assert f(-0.0, -1.0) == -pi
    assert f(0.0, 0.0) == 0.0
    assert math.copysign(1, f(-0.0, 0.0)) == -1
    assert f(-0.0, -0.0) == -pi
    assert f(0.0, -0.0) == pi
    rng = np.random.default_rng(0)
    for y, x in rng.uniform(-1, 1, (2000, 2)).astype(np.float32):
        assert abs(f(float(y), float(x)) - math.atan2(float(y), float(x))) < 4e-7


def test_fm_demod_semantics(oracle):
    x = oracle.convert_u8(S.iq_u8_fm(4096))
    y = oracle.fm_demod(x)
    assert y[0] == 0.0                                     # last = 0 -> phase (0:+0) = 0 (Demod.hs:41)
    z = x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64)
    exp = np.angle(z[1:] * np.conj(z[:-1]))
    assert np.abs(y[1:] - exp).max() < 1e-6
    # carrying the last sample across buffers == one long buffer (Demod.hs:43-46)
    a = oracle.fm_demod(x[: 2 * 1000])
    b = oracle.fm_demod(x[2 * 1000:], (float(x[1998]), float(x[1999])))
    assert np.array_equal(np.concatenate([a, b]).view(np.uint32), y.view(np.uint32))
    # a 1 kHz tone at 25 kHz deviation, fs = 1.28 MS/s: |instantaneous phase step| ~ 2*pi*25e3/1.28e6 (+ noise)
    assert np.abs(y[1:]).max() < 2 * np.pi * 25e3 / 1.28e6 * 2.0
