"""Seeded synthetic inputs shared by tests, golden generation and bench.py."""
import numpy as np

SEED_IQ, SEED_CF, SEED_RF = 1001, 1002, 1003


def windowed_sinc(size, cutoff, window="hamming"):
    """Odd-length windowed-sinc low-pass (the standard formulas SDR.FilterDesign also uses), f32."""
    assert size % 2 == 1
    idx = np.arange(size) - (size - 1) // 2
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(idx == 0, cutoff, np.sin(np.pi * cutoff * idx) / (idx * np.pi))
    n = np.arange(size)
    if window == "hamming":
        w = 0.54 - 0.46 * np.cos(2 * np.pi * n / (size - 1))
    elif window == "hanning":
        w = 0.5 * (1 - np.cos(2 * np.pi * n / (size - 1)))
    else:
        w = 0.42 - 0.5 * np.cos(2 * np.pi * n / (size - 1)) + 0.08 * np.cos(4 * np.pi * n / (size - 1))
    return (s * w).astype(np.float32)


def taps_decim127():
    """T127: 127-tap low-pass, cutoff 1/16 (-> padded to 128 by the AVX complex constructor)."""
    return windowed_sinc(127, 1.0 / 16)


def taps_resamp191():
    """T191: 191-tap low-pass for the 3/10 resampler (cutoff 1/10 of the 3x-upsampled rate)."""
    return (3.0 * windowed_sinc(191, 1.0 / 10)).astype(np.float32)


def taps_audio_half64():
    """T128s: first half (64 taps) of an even-length-128 symmetric low-pass, cutoff 0.3."""
    n = np.arange(128) - 63.5
    s = np.sin(np.pi * 0.3 * n) / (n * np.pi)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(128) / 127)
    return (s * w).astype(np.float32)[:64]


def iq_u8(n_samples, seed=SEED_IQ):
    """Uniform u8 IQ, interleaved (2*n bytes)."""
    return np.random.default_rng(seed).integers(0, 256, 2 * n_samples, dtype=np.uint8)


def iq_u8_fm(n_samples, fs=1.28e6, f_mod=1e3, dev=25e3, seed=SEED_IQ):
    """u8-quantised FM signal (1 kHz tone, 25 kHz deviation: inside the +-40 kHz passband of
    the 1/16-cutoff decimation filter) + a little noise."""
    t = np.arange(n_samples) / fs
    phase = 2 * np.pi * dev / (2 * np.pi * f_mod) * np.sin(2 * np.pi * f_mod * t)
    rng = np.random.default_rng(seed)
    z = 0.8 * np.exp(1j * phase) + 0.02 * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    out = np.empty(2 * n_samples, np.uint8)
    out[0::2] = np.clip(np.round(z.real * 127.5 + 127.5), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.round(z.imag * 127.5 + 127.5), 0, 255).astype(np.uint8)
    return out


def cfloat_block(n_samples, seed=SEED_CF, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, 2 * n_samples).astype(np.float32)


def real_block(n, seed=SEED_RF, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, n).astype(np.float32)


def gauss_taps(n, seed, sigma=0.05):
    return np.random.default_rng(seed).normal(0, sigma, n).astype(np.float32)


# "Example-shaped" tap sets: the lengths of the reference FM example's real filters
# (examples/fm/Coeffs.hs: 51-tap RF decimation filter, 31-tap audio resampler filter, 32 half-taps of
# a 64-tap symmetric audio filter) with our own windowed-sinc values.
def taps_decim51():
    return windowed_sinc(51, 1.0 / 16, "blackman")


def taps_resamp31():
    return (3.0 * windowed_sinc(31, 1.0 / 10)).astype(np.float32)


def taps_audio_half32():
    n = np.arange(64) - 31.5
    s = np.sin(np.pi * 0.3 * n) / (n * np.pi)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(64) / 63)
    return (s * w).astype(np.float32)[:32]


# The reference FM example's REAL tap tables (examples/fm/Coeffs.hs:11-154; Octave remez designs), as data:
# tests/golden/example_taps.npz, written by tests/golden/make_example_taps.py in the build container.
def _example_taps(key):
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_taps.npz"))
    return np.ascontiguousarray(d[key], dtype=np.float32)


def taps_example_rf_decim():        # 51 taps, fastDecimatorC 8 (fm.hs:30)
    return _example_taps("rf_decim")


def taps_example_audio_resampler():  # 31 taps, fastResamplerR 3 10 (fm.hs:31)
    return _example_taps("audio_resampler")


def taps_example_audio_filter_half():  # 32 = the first half of a 64-tap symmetric filter, fastFilterSymR (fm.hs:32)
    return _example_taps("audio_filter_half")
