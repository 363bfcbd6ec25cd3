"""dcBlocker (c_sources/filter.c:152-161): oracle vs the reference build on CPU, and the device
implementation (speculative chunks + verification) vs the oracle, bit for bit, including the settling path."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal


def _signals(n, seed=77):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    return {
        "uniform": rng.uniform(-1, 1, n).astype(np.float32),
        "audio+dc": (0.3 * np.sin(2 * np.pi * t / 480.0) + 0.05 * rng.standard_normal(n) + 0.4).astype(np.float32),
        "constant": np.full(n, 0.75, np.float32),
        "steps": np.repeat(rng.uniform(-100, 100, (n + 999) // 1000), 1000)[:n].astype(np.float32),
        "wide-range": (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32),
        "zeros": np.zeros(n, np.float32),
    }


@pytest.mark.parametrize("n", [0, 1, 7, 8192, 100_000])
def test_oracle_matches_reference(oracle, ref, n):
    for name, x in _signals(n).items():
        exp, efs, efo = ref.dc_blocker(x, 0.25, -0.5)
        got, fs, fo = oracle.dc_blocker(x, 0.25, -0.5)
        assert_bit_equal(got, exp, f"dcBlocker {name}")
        assert np.float32(fs).tobytes() == np.float32(efs).tobytes() and np.float32(fo).tobytes() == np.float32(efo).tobytes()


def _gpu_run(hip, x, ls, lo, run_in=0, use_ws=True, misalign=0):
    from gpu_util import ptr
    n = x.size
    d_in_full = torch.zeros(n + 8, dtype=torch.float32, device="cuda")
    d_in = d_in_full[misalign: misalign + n]
    d_in.copy_(torch.from_numpy(x))
    d_out = torch.empty(n + 8, dtype=torch.float32, device="cuda")[misalign: misalign + n]
    fin = torch.zeros(2, dtype=torch.float32, device="cuda")
    wsb = hip.lib.sdrhip_dc_blocker_workspace_bytes(n)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    hip.check(hip.lib.sdrhip_dc_blocker_run(None, ptr(d_in), ptr(d_out), n, ls, lo, ptr(fin), ptr(ws) if use_ws else None,
                                            wsb if use_ws else 0, run_in), "sdrhip_dc_blocker_run")
    torch.cuda.synchronize()
    stats = ws[:12].cpu().numpy().view(np.uint32)
    return d_out.cpu().numpy(), fin.cpu().numpy(), tuple(int(v) for v in stats)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 8192, 24_576, 100_001, 1 << 20])
def test_gpu_matches_oracle(hip, oracle, n):
    for name, x in _signals(n).items():
        exp, efs, efo = oracle.dc_blocker(x, 0.25, -0.5)
        got, fin, stats = _gpu_run(hip, x, 0.25, -0.5)
        assert_bit_equal(got, exp, f"dcBlocker {name} n={n}")
        assert_bit_equal(fin, np.array([efs, efo], np.float32), f"dcBlocker final state {name}")


@pytest.mark.gpu
def test_gpu_settles_unconverged_chunks(hip, oracle):
    """A run-in far too short to converge: most chunks start from the wrong state and the settle pass has to repair them.
    The result must not depend on it."""
    n = 1 << 18
    for name, x in _signals(n).items():
        exp, efs, efo = oracle.dc_blocker(x, -3.0, 2.0)
        got, fin, stats = _gpu_run(hip, x, -3.0, 2.0, run_in=64)
        assert_bit_equal(got, exp, f"dcBlocker settle {name}")
        assert_bit_equal(fin, np.array([efs, efo], np.float32), "final state")
        if name in ("uniform", "audio+dc"):
            assert stats[2] > 0, "the short run-in should have left work for the repair rounds"
    # default run-in: nothing to settle on ordinary signals
    for name in ("uniform", "audio+dc", "steps"):
        _, _, stats = _gpu_run(hip, _signals(n)[name], -3.0, 2.0)
        assert stats == (0, 0, 0), (name, stats)


@pytest.mark.gpu
def test_gpu_unaligned_and_sequential_paths(hip, oracle):
    x = _signals(70_000)["audio+dc"]
    exp, _, _ = oracle.dc_blocker(x, 0.0, 0.0)
    got, _, _ = _gpu_run(hip, x, 0.0, 0.0, misalign=1)        # 4-byte aligned only: scalar loads
    assert_bit_equal(got, exp, "unaligned")
    got, _, _ = _gpu_run(hip, x, 0.0, 0.0, use_ws=False)      # no workspace: sequential walk
    assert_bit_equal(got, exp, "sequential")


@pytest.mark.gpu
def test_gpu_full_size_blockwise_state(hip, oracle):
    """2^24 samples in one call == the same stream in 8192-sample calls chained through (finalSample, finalOutput),
    which is what the Pipe dcBlockingFilter (Filter.hs:730-739) does."""
    n = 1 << 24
    x = _signals(n, seed=5)["audio+dc"]
    got, fin, stats = _gpu_run(hip, x, 0.0, 0.0)
    exp, efs, efo = oracle.dc_blocker(x, 0.0, 0.0)
    assert_bit_equal(got, exp, "2^24")
    ls = lo = 0.0
    for b in range(0, 1 << 16, 8192):                          # the first 8 blocks through the drop-in symbol
        out, ls, lo = hip.DropIn.dc_blocker(x[b: b + 8192], ls, lo)
        assert_bit_equal(out, exp[b: b + 8192], f"block {b // 8192}")


@pytest.mark.gpu
def test_gpu_pipe_dc_blocking_filter(hip, oracle):
    """The Pipe (Filter.hs:730-739): one output block per input block, state carried across blocks (on the device)."""
    x = _signals(200_000, seed=9)["audio+dc"]
    exp, _, _ = oracle.dc_blocker(x, 0.0, 0.0)
    pipe = hip.dcBlockingFilter()
    cuts = [0, 8192, 16384, 16391, 50_000, 120_000, 200_000]      # short, ragged and long blocks (the long ones speculate)
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        outs += pipe.push(x[a:b])
    outs += pipe.flush()
    assert [o.size for o in outs] == [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    assert_bit_equal(np.concatenate(outs), exp, "dcBlockingFilter pipe")
