"""tests/golden/example_taps.npz: the three tap tables of the reference's FM receiver example as DATA.

    python tests/golden/make_example_taps.py          (build container only: reads /root/reference/examples/fm/Coeffs.hs:11-154)

The example designs its filters in Octave (remez) and keeps the results as Haskell `[Float]` literals: 51 taps for the RF
decimator, 31 for the audio resampler, 32 = the first half of a 64-tap symmetric audio filter (examples/fm/fm.hs:30-32 hands
them to fastDecimatorC 8 / fastResamplerR 3 10 / fastFilterSymR).  SURVEY.md 8(d) asks for them as a second tap set: they are
hand-designed equiripple filters, unlike every windowed-sinc set tests/signals.py computes.  A decimal literal of type Float
is `fromRational` of the exact decimal: the nearest binary32, ties to even -- computed here with exact rationals, not through a
double.  Only the numbers travel (this file's output); no text of the reference is kept.
"""
import os
import re
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/examples/fm/Coeffs.hs"
NAMES = {"coeffsRFDecim": "rf_decim", "coeffsAudioResampler": "audio_resampler", "coeffsAudioFilter": "audio_filter_half"}


def nearest_f32(lit):
    """The binary32 nearest to the decimal literal (round-half-even), by exact rational comparison of the candidates."""
    q = Fraction(lit)
    g = np.float32(float(q))
    cands = {g, np.nextafter(g, np.float32(np.inf)), np.nextafter(g, np.float32(-np.inf))}
    best = min(cands, key=lambda c: (abs(Fraction(float(c)) - q), int(np.float32(c).view(np.uint32)) & 1))
    return np.float32(best)


def main():
    text = open(SRC).read()
    out = {}
    for hs, key in NAMES.items():
        m = re.search(hs + r"\s*=\s*\[(.*?)\]", text, re.S)
        out[key] = np.array([nearest_f32(v.strip()) for v in m.group(1).split(",") if v.strip()], np.float32)
    assert [out[k].size for k in ("rf_decim", "audio_resampler", "audio_filter_half")] == [51, 31, 32]
    path = os.path.join(HERE, "example_taps.npz")
    np.savez(path, **out)
    print("wrote", path, {k: v.size for k, v in out.items()})


if __name__ == "__main__":
    main()
