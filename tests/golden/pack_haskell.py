"""Pack what haskell/GenFixtures.hs wrote into tests/golden/haskell_fixtures.npz (inputs by CRC, outputs as uint32 bit
patterns) -- the file tests/test_golden_haskell.py looks for.

    python tests/golden/pack_haskell.py <dir used with make_haskell_inputs.py and gen-fixtures>"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INPUTS = ("demod_in.cf32", "decim_in.cf32", "resamp_in.f32", "filt_in.f32", "rx_in.u8",
          "taps_decim.f32", "taps_resamp.f32", "taps_audio_half.f32")
OUTPUTS = ("demod_out.f32", "demod_ragged_out.f32", "decim_out.cf32", "resamp_out.f32", "filt_out.f32", "rx_out.f32")


def main():
    d = sys.argv[1]
    data = {}
    for name in INPUTS:
        raw = open(os.path.join(d, name), "rb").read()
        data["crc__" + name] = np.array([zlib.crc32(raw)], np.uint32)
    for name in OUTPUTS:
        data[name] = np.fromfile(os.path.join(d, name), np.uint32)
    path = os.path.join(HERE, "haskell_fixtures.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
