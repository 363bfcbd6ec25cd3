"""Inputs for haskell/GenFixtures.hs (the GHC-side pin of fmDemod and the Pipes, SURVEY.md 8(c) "parity unpinned").

    python tests/golden/make_haskell_inputs.py <dir>

Writes raw little-endian arrays a GHC build of adamwalker/sdr reads back (see haskell/GenFixtures.hs for the three
commands).  Everything is regenerated from tests/signals.py's seeds; the fmDemod input additionally walks the corners of
GHC's atan2 case analysis (signed zeros, the axes, equal magnitudes, denormals, huge ratios) -- exactly the inputs on which
a restated formula and the real `RealFloat` default could differ (all products stay finite: no NaN payloads to argue about)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import signals as S  # noqa: E402

B = 8192


def demod_input():
    """4 blocks of IQ: an FM carrier (what the receiver sees), noise, then products whose (re, im) hit every clause of
    atan2: the special values are placed as CONSECUTIVE PAIRS (prev, cur) so that cur * conj(prev) lands on them."""
    x = [np.asarray(S.cfloat_block(B), np.float32)]
    fm = (S.iq_u8_fm(B).astype(np.float32) - 128.0) * np.float32(1.0 / 128.0)
    x.append(fm)
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0 ** -149, -(2.0 ** -149), 2.0 ** -126, 1.0e18, -1.0e18,
                   1.0e-30, -1.0e-30, 0.4375, 0.6875, 1.1875, 2.4375, 7.0, 2.0 ** 25, 2.0 ** 26], np.float32)
    rng = np.random.default_rng(55)
    pairs = []
    for _ in range(B):
        pairs += [rng.choice(sp), rng.choice(sp)]
    x.append(np.array(pairs, np.float32))
    # prev = (1, 0): cur * conj(prev) = cur itself, so (re, im) sweep the special grid directly
    grid = []
    vals = list(sp)
    k = 0
    while len(grid) < 2 * B:
        grid += [1.0, 0.0, vals[k % len(vals)], vals[(k // len(vals)) % len(vals)]]
        k += 1
    x.append(np.array(grid[: 2 * B], np.float32))
    return np.concatenate(x)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "haskell_inputs")
    os.makedirs(out, exist_ok=True)
    demod_input().tofile(os.path.join(out, "demod_in.cf32"))
    S.taps_decim127().astype(np.float32).tofile(os.path.join(out, "taps_decim.f32"))
    S.taps_resamp191().astype(np.float32).tofile(os.path.join(out, "taps_resamp.f32"))
    S.taps_audio_half64().astype(np.float32).tofile(os.path.join(out, "taps_audio_half.f32"))
    S.cfloat_block(20 * B, seed=61).astype(np.float32).tofile(os.path.join(out, "decim_in.cf32"))
    S.real_block(12 * B, seed=62).astype(np.float32).tofile(os.path.join(out, "resamp_in.f32"))
    S.real_block(6 * B, seed=63).astype(np.float32).tofile(os.path.join(out, "filt_in.f32"))
    S.iq_u8_fm(200 * B).tofile(os.path.join(out, "rx_in.u8"))
    print("wrote the inputs of haskell/GenFixtures.hs to", out)


if __name__ == "__main__":
    main()
