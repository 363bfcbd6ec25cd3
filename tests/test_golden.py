"""Golden vectors generated from the reference's own compiled C (tests/golden/make_golden.py):
the CPU restatement must reproduce them bit for bit (CPU test), and so must the HIP
drop-in symbols (GPU test)."""
import os

import numpy as np
import pytest

from golden_cases import Providers, cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz")


def _check(p):
    gold = np.load(GOLDEN)
    got = cases(p)
    assert set(got) == {k for k in gold.files if not k.endswith("__crc")}
    for name, (arr, c) in got.items():
        assert int(gold[name + "__crc"][0]) == int(c), f"{name}: seeded inputs changed (PRNG drift?)"
        a = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32)
        b = gold[name]
        assert a.shape == b.shape, name
        bad = int((a != b).sum())
        assert bad == 0, f"{name}: {bad}/{a.size} values differ from the reference's output"


def test_oracle_reproduces_reference_outputs(oracle):
    _check(Providers("oracle", oracle))


@pytest.mark.gpu
def test_hip_reproduces_reference_outputs(hip, oracle):
    _check(Providers("hip", oracle, hip=hip))
