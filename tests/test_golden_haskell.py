"""The GHC-side pin of A5 (fmDemod) and A9 (the Pipes' state machines), SURVEY.md 8(c).

tests/golden/haskell_fixtures.npz holds outputs of the REFERENCE'S OWN HASKELL (haskell/GenFixtures.hs, run by a maintainer
of adamwalker/sdr against inputs from tests/golden/make_haskell_inputs.py; no GHC in this repository's build image).
While the file is absent these rows stay "parity unpinned" and the comparisons below are skipped -- what still runs is the
plumbing: the inputs are reproducible, and the comparison code is exercised on a stand-in computed by the restatement
(which pins nothing, and is never written to the fixture's path)."""
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from conftest import assert_bit_equal  # noqa: E402
from oracle import pipes_model as PM  # noqa: E402
import make_haskell_inputs as MK  # noqa: E402
import pack_haskell as PK  # noqa: E402
import signals as S  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "haskell_fixtures.npz")
B = 8192
RAGGED = [1, 2, 777, 4096, 129, 8191]


def _blocks(x, w, n=B):
    return [x[i * w:(i + n) * w] for i in range(0, x.size // w - n + 1, n)]


def _ragged(x, w):
    out, pos = [], 0
    for s in RAGGED:
        if x.size // w - pos < s:
            break
        out.append(x[pos * w:(pos + s) * w])
        pos += s
    rest = x[pos * w:]
    return out + _blocks(rest, w)


def restated_outputs(d, oracle):
    """What the restatement (oracle + pipes_model) computes for the inputs in directory d: name -> float32 array."""
    rd = lambda name, dt=np.float32: np.fromfile(os.path.join(d, name), dt)
    iq = rd("demod_in.cf32")
    out = {"demod_out.f32": np.concatenate(PM.fm_demod_pipe(oracle, _blocks(iq, 2))),
           "demod_ragged_out.f32": np.concatenate(PM.fm_demod_pipe(oracle, _ragged(iq, 2)))}
    td, tr, th = rd("taps_decim.f32"), rd("taps_resamp.f32"), rd("taps_audio_half.f32")
    blk, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, td, PM.ORDER_AVX, complex_=True, factor=8), _blocks(rd("decim_in.cf32"), 2), B)
    out["decim_out.cf32"] = np.concatenate(blk)
    blk, _ = PM.fir_resampler_pipe(PM.ResamplerModel(oracle, 3, 10, tr, PM.ORDER_AVX), _blocks(rd("resamp_in.f32"), 1), B)
    out["resamp_out.f32"] = np.concatenate(blk)
    blk, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, th, PM.ORDER_AVX, sym=True), _blocks(rd("filt_in.f32"), 1), B)
    out["filt_out.f32"] = np.concatenate(blk)
    u8 = rd("rx_in.u8", np.uint8)
    out["rx_out.f32"] = np.concatenate(PM.fm_receiver(oracle, _blocks(u8, 2), td, 8, tr, 3, 10, th, 0.2, B))
    return out


def hip_outputs(d, hip):
    """The same through libsdr_hip.so's host-block operators (the reference's operator surface)."""
    rd = lambda name, dt=np.float32: np.fromfile(os.path.join(d, name), dt)

    def drive(pipe, blocks):
        outs = []
        for b in blocks:
            outs += pipe.push(b)
        return np.concatenate(outs + pipe.flush())

    iq = rd("demod_in.cf32")
    td, tr, th = rd("taps_decim.f32"), rd("taps_resamp.f32"), rd("taps_audio_half.f32")
    out = {"demod_out.f32": drive(hip.fmDemod(), _blocks(iq, 2)),
           "demod_ragged_out.f32": drive(hip.fmDemod(), _ragged(iq, 2)),
           "decim_out.cf32": drive(hip.firDecimator(hip.Decimator(8, td, hip.ORDER_AVX, complex_=True), B), _blocks(rd("decim_in.cf32"), 2)),
           "resamp_out.f32": drive(hip.firResampler(hip.Resampler(3, 10, tr, hip.ORDER_AVX), B), _blocks(rd("resamp_in.f32"), 1)),
           "filt_out.f32": drive(hip.firFilter(hip.Filter(th, hip.ORDER_AVX, sym=True), B), _blocks(rd("filt_in.f32"), 1))}
    chain = hip.FmChain(8, td, 3, 10, tr, th, 0.2, B)
    st = hip.FmStream(chain, B, B)
    outs = []
    for b in _blocks(rd("rx_in.u8", np.uint8), 2):
        outs += st.push(b)
    out["rx_out.f32"] = np.concatenate(outs + st.flush())
    return out


def compare(fix, got, what):
    for name in PK.OUTPUTS:
        exp = fix[name].view(np.float32)
        g = got[name][: exp.size]
        assert g.size == exp.size, f"{what}: {name}: {got[name].size} values, the GHC run has {exp.size}"
        assert_bit_equal(g, exp, f"{what}: {name} vs the reference's own Haskell")


def check_inputs(fix, d):
    for name in PK.INPUTS:
        crc = zlib.crc32(open(os.path.join(d, name), "rb").read())
        assert crc == int(fix["crc__" + name][0]), f"{name}: the regenerated input is not the one the GHC run consumed"


@pytest.fixture(scope="module")
def inputs_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("hsfix"))
    sys.argv, old = ["make_haskell_inputs.py", d], sys.argv
    try:
        MK.main()
    finally:
        sys.argv = old
    return d


def test_inputs_are_reproducible(inputs_dir, tmp_path):
    sys.argv, old = ["make_haskell_inputs.py", str(tmp_path)], sys.argv
    try:
        MK.main()
    finally:
        sys.argv = old
    for name in PK.INPUTS:
        assert open(os.path.join(inputs_dir, name), "rb").read() == open(os.path.join(tmp_path, name), "rb").read(), name
    iq = np.fromfile(os.path.join(inputs_dir, "demod_in.cf32"), np.float32)
    assert iq.size == 2 * 4 * B and np.any(np.signbit(iq) & (iq == 0)) and np.any(np.abs(iq) > 1e17)   # the atan2 corners are in


def test_comparison_plumbing_on_a_stand_in(inputs_dir, oracle):
    """The comparator on a stand-in built from the restatement itself: proves the plumbing, pins nothing."""
    got = restated_outputs(inputs_dir, oracle)
    stand_in = {name: np.ascontiguousarray(v, np.float32).view(np.uint32) for name, v in got.items()}
    for name in PK.INPUTS:
        stand_in["crc__" + name] = np.array([zlib.crc32(open(os.path.join(inputs_dir, name), "rb").read())], np.uint32)
    check_inputs(stand_in, inputs_dir)
    compare(stand_in, got, "stand-in")
    assert got["rx_out.f32"].size >= 5 * B and got["decim_out.cf32"].size == 2 * 2 * B


@pytest.mark.skipif(not os.path.exists(FIX), reason="A5 / A9 parity unpinned: tests/golden/haskell_fixtures.npz (haskell/GenFixtures.hs) not generated yet")
def test_restatement_equals_the_references_haskell(inputs_dir, oracle):
    fix = np.load(FIX)
    check_inputs(fix, inputs_dir)
    compare(fix, restated_outputs(inputs_dir, oracle), "restatement (A5 / A9 pinned)")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FIX), reason="A5 / A9 parity unpinned: tests/golden/haskell_fixtures.npz (haskell/GenFixtures.hs) not generated yet")
def test_hip_equals_the_references_haskell(inputs_dir, hip):
    fix = np.load(FIX)
    check_inputs(fix, inputs_dir)
    compare(fix, hip_outputs(inputs_dir, hip), "libsdr_hip.so (A5 / A9 pinned)")


@pytest.mark.gpu
def test_hip_equals_restatement_on_the_fixture_inputs(inputs_dir, oracle, hip):
    """Without the GHC fixture: the product against the restatement on the very inputs the fixture is defined on (the
    atan2 corner grid included)."""
    exp = restated_outputs(inputs_dir, oracle)
    got = hip_outputs(inputs_dir, hip)
    for name in PK.OUTPUTS:
        assert_bit_equal(got[name][: exp[name].size], exp[name], name)
        assert got[name].size >= exp[name].size
