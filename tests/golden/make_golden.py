"""Generate tests/golden/reference_outputs.npz from THE REFERENCE ITSELF.

Run in the build container (needs oracle/_ref/libsdr_ref.so, i.e. /root/reference):
    python tests/golden/make_golden.py
The case table lives in tests/golden_cases.py.  Inputs are regenerated from seeds
(tests/signals.py); what is committed is the reference's OUTPUT for each case (uint32
bit patterns) plus a CRC of the inputs, so a drift of the PRNG stream would be
noticed rather than silently changing the question.  The reference ships no golden
vectors of its own: its compiled C is the pin.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from golden_cases import Providers, cases  # noqa: E402
from oracle.oracle import Oracle, Ref  # noqa: E402


def main():
    p = Providers("ref", Oracle(), ref=Ref())
    data = {}
    for name, (arr, c) in cases(p).items():
        data[name] = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32)
        data[name + "__crc"] = np.array([c], np.uint32)
    path = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes,", len(data) // 2, "cases")


if __name__ == "__main__":
    main()
