import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/sdr_oracle.c), built on demand.  Checker only."""
    from oracle.oracle import Oracle, build, ORACLE_SO
    if not os.path.exists(ORACLE_SO):
        build()
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own compiled C (oracle/_ref), when it was built here."""
    from oracle.oracle import Ref, have_ref, build
    if not have_ref() and os.path.isdir("/root/reference/c_sources"):
        build()
    if not have_ref():
        pytest.skip("oracle/_ref/libsdr_ref.so not present (built only where /root/reference exists)")
    return Ref()


@pytest.fixture(scope="session")
def hip():
    """The product library through its ctypes binding; requires a GPU."""
    from sdr_amd import build as B
    if not os.path.exists(B.LIB):
        B.build()
    import sdr_amd.lib as L
    if L.device_count() < 1:
        # a GPU test that cannot see a GPU is a failure, not a skip: silent skips would read as "green"
        pytest.fail("libsdr_hip.so sees no HIP device (GPU tests are selected with -m gpu on an MI355X box)")
    return L


@pytest.fixture(autouse=True)
def _output_guard_bands(request):
    """After every GPU test: the NaN canaries around each output buffer the test got from gpu_util.dev_empty_f32 are intact."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gpu_util
        if gpu_util._live:
            gpu_util.check_guards()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ba, bb = a.view(np.uint32), b.view(np.uint32)
    bad = np.nonzero(ba != bb)[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.size}/{a.size} elements differ; first at {i}: {a[i]!r} ({ba[i]:#x}) vs {b[i]!r} ({bb[i]:#x})")
