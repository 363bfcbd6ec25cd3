"""The restated Pipes state machines (oracle/pipes_model.py; Filter.hs:504-727) and the
host-side tap preparation (A10).  The reference does not test these at all
(SURVEY.md 4); the checks here are its own `assert` invariants, the probed
per-block call pattern, and equivalences that must hold by construction."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import pipes_model as PM
from oracle.oracle import duplicate
import signals as S

B = 8192


def _blocks(x, w, n, size=B):
    return [x[i * size * w:(i + 1) * size * w] for i in range(n)]


def test_decimator_call_pattern(oracle):
    """8192-sample blocks, 128 padded taps, /8, blockSizeOut 8192:
    One(1009) then Cross(15) per input block; one output block per 8 input blocks."""
    x = S.cfloat_block(17 * B)
    m = PM.FilterModel(oracle, S.taps_decim127(), PM.ORDER_AVX, complex_=True, factor=8)
    assert m.num_coeffs == 128
    out, trace = PM.fir_decimator_pipe(m, _blocks(x, 2, 17), B)
    assert trace[:6] == [("one", 1009), ("cross", 15)] * 3
    assert len(out) == 2 and all(o.size == 2 * B for o in out)


def test_filter_and_resampler_call_pattern(oracle):
    x = S.real_block(5 * B)
    f = PM.FilterModel(oracle, S.taps_audio_half64(), PM.ORDER_AVX, sym=True)
    assert f.num_coeffs == 128
    out, trace = PM.fir_filter_pipe(f, _blocks(x, 1, 5), B)
    assert trace[:2] == [("one", 8065), ("cross", 127)]
    r = PM.ResamplerModel(oracle, 3, 10, S.taps_resamp191(), PM.ORDER_AVX)
    assert r.num_coeffs == 192 and list(r.prep["increments"]) == [4, 3, 3] and list(r.prep["offsets"]) == [0, 2, 1]
    out, trace = PM.fir_resampler_pipe(r, _blocks(x, 1, 5), B)
    ones = [c for k, c in trace if k == "one"]
    crosses = [c for k, c in trace if k == "cross"]
    assert set(crosses) == {19} and set(ones[:3]) <= {2439, 2438}


def test_stream_outputs_equal_contiguous_kernel_away_from_seams(oracle):
    """Global stream semantics: away from the seams the Pipe output k is
    the C kernel's output k on the whole contiguous stream."""
    x = S.cfloat_block(4 * B)
    taps = S.taps_decim127()
    m = PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=8)
    out, _ = PM.fir_decimator_pipe(m, _blocks(x, 2, 4), 1024)
    got = np.concatenate(out)
    h = np.concatenate([taps, np.zeros(1, np.float32)])
    K = got.size // 2
    whole = oracle.decimate_rc(4, K, 8, duplicate(h), x)
    k = np.arange(K)
    cross = ((k * 8) // B) != ((k * 8 + 127) // B)
    same = got.view(np.uint32).reshape(-1, 2) == whole.view(np.uint32).reshape(-1, 2)
    assert same[~cross].all()
    assert cross.sum() == 15 * 3 and not same[cross].all()     # the seam outputs use the sequential order


def test_pipe_is_invariant_to_output_block_size(oracle):
    x = S.real_block(6 * B)
    r = PM.ResamplerModel(oracle, 3, 10, S.taps_resamp191(), PM.ORDER_AVX)
    a = np.concatenate(PM.fir_resampler_pipe(r, _blocks(x, 1, 6), 512)[0])
    b = np.concatenate(PM.fir_resampler_pipe(r, _blocks(x, 1, 6), 4096)[0])
    n = min(a.size, b.size)
    assert n > 8000
    assert_bit_equal(a[:n], b[:n], "blockSizeOut only re-blocks")


@pytest.mark.parametrize("I,D", [(3, 10), (2, 3), (5, 7), (7, 11), (13, 17), (3, 23), (2, 4), (4, 6)])
def test_resampler_phase_closed_form(oracle, I, D):
    """inOff(k) = ceil(k*D/I), filtOff(k) = inOff*I - k*D; the
    group sequence of prepareCoeffs visits exactly those offsets."""
    prep = oracle.prepare_coeffs(8, I, D, S.gauss_taps(100, 1))
    off, pos = 0, 0
    for k in range(200):
        assert pos == -((-k * D) // I)
        assert off == pos * I - k * D
        g = list(prep["offsets"]).index(off)
        assert prep["increments"][g] == (D - off - 1) // I + 1
        pos += (D - off - 1) // I + 1
        off = I - 1 - (D - off - 1) % I
    import math
    assert prep["num_groups"] == I // math.gcd(I, D)


def test_prepare_coeffs_matches_strided_taps(oracle):
    h = S.taps_resamp191()
    prep = oracle.prepare_coeffs(8, 3, 10, h)
    assert prep["num_coeffs"] == 64 and prep["padded_len"] == 64
    for g, off in enumerate(prep["offsets"]):
        strided = h[off::3]
        assert np.array_equal(prep["groups"][g][: strided.size], strided)
        assert not prep["groups"][g][strided.size:].any()


def test_short_block_asserts(oracle):
    f = PM.FilterModel(oracle, S.gauss_taps(128, 1), PM.ORDER_AVX)
    with pytest.raises(PM.PipeAssert):
        PM.fir_filter_pipe(f, [S.real_block(100)], 64)


def test_fm_receiver_runs_and_is_band_limited(oracle):
    nblk = 60
    u8 = S.iq_u8_fm(nblk * B)
    out = PM.fm_receiver(oracle, [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)], S.taps_decim127(), 8,
                         S.taps_resamp191(), 3, 10, S.taps_audio_half64(), 0.2, B)
    assert len(out) == 1 and out[0].size == B
    a = out[0].astype(np.float64)
    # a 1 kHz tone at 48 kHz: the spectrum peaks at 1 kHz
    spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
    assert abs(np.argmax(spec[1:]) + 1 - 1000 / 48000 * a.size) < 3
