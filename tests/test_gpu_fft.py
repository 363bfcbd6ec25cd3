"""The spectrum path (sdr_amd/csrc/fft.cpp): SDR.FFT's fftw' / fftwReal' / fftwParallel (FFT.hs:44-168)
on hipFFT.  Floating point with a different summation tree from FFTW's, so the contract is a tolerance: every bin within
1e-11 of the largest bin magnitude of an f64 reference DFT in FFTW's order and sign (numpy.fft uses the same conventions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.mark.parametrize("n", [64, 1000, 1024, 8192, 65536])
def test_complex_dft_matches_f64_reference(hip, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    got = hip.Fft(n).run(x)
    ref = np.fft.fft(x)
    assert got.shape == (n,)
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))


@pytest.mark.parametrize("n", [256, 1000, 8192])
def test_real_dft_matches_f64_reference(hip, n):
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n)
    got = hip.Fft(n, real_input=True).run(x)
    ref = np.fft.rfft(x)
    assert got.shape == (n // 2 + 1,)
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))


def test_batched_plan_is_the_parallel_pipe(hip):
    """fftwParallel (FFT.hs:112-168) overlaps several buffers with a thread pool; here one batched plan transforms them in
    one call -- same outputs, same order."""
    n, batch = 4096, 8
    rng = np.random.default_rng(3)
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    got = hip.Fft(n, batch=batch).run(x)
    ref = np.fft.fft(x, axis=1)
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))
    # a tone lands in its bin with FFTW's sign convention (forward = exp(-2 pi i jk/n))
    k = 37
    tone = np.exp(2j * np.pi * k * np.arange(n) / n)
    spec = hip.Fft(n).run(tone)
    assert np.argmax(np.abs(spec)) == k and abs(spec[k] - n) < 1e-8 * n
