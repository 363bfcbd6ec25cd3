"""fmDemod restatement (Demod.hs:21-46 + GHC base atan2 + libm atanf).  The reference pins
nothing here (no test, no vector) and GHC cannot run in this image: these tests pin
the restatement to the image's libm and to mathematical properties.  'parity unpinned'
vs a real GHC run remains."""
import math

import numpy as np

import signals as S


def test_atanf_model_matches_libm_on_a_dense_sweep(oracle):
    """The f32 fdlibm evaluation the GPU kernels use == the host libm atanf.
    (All 2^32 inputs were swept once when this was written: 0 mismatches; here every
    97th float plus the neighbourhoods of the range boundaries.)"""
    bad, first = oracle.atanf_sweep(0, 0xFFFFFFFF, 97)
    assert bad == 0, f"first mismatch at {first:#x}"
    for edge in (0x31000000, 0x3ee00000, 0x3f300000, 0x3f980000, 0x401c0000, 0x4c000000, 0x7f800000):
        for sign in (0, 0x80000000):
            lo = (edge | sign) - 4096
            bad, first = oracle.atanf_sweep(lo, lo + 8192, 1)
            assert bad == 0, f"first mismatch at {first:#x}"


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
