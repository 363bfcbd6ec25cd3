"""Device-memory helpers for the GPU tests: torch is plumbing (allocation, copies), nothing else.

Round 5 (VERDICT r04 "missing" 5): every output buffer handed out by dev_empty_f32 sits between two guard bands of a NaN canary
(0x7fc0dead); to_host checks the bands of the buffer it copies back, and conftest's autouse fixture checks every buffer a GPU test
allocated when the test ends -- a kernel that writes one element before or past its output fails the test that ran it (the bug class
is the reference's own: c_sources/convert.c:27,42 reads past its input)."""
import numpy as np
import torch

GUARD = 64                      # floats on either side: 256 B, so the payload keeps the allocation's 256-byte alignment
CANARY = 0x7FC0DEAD
_live = []                      # (whole tensor incl. guards, payload length)


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev_empty_f32(n):
    n = int(n)
    whole = torch.full((n + 2 * GUARD,), CANARY, dtype=torch.int32, device="cuda").view(torch.float32)
    _live.append((whole, n))
    return whole[GUARD:GUARD + n]


def _check(whole, n, what):
    w = whole.view(torch.int32)
    lo, hi = w[:GUARD], w[GUARD + n:]
    bad_lo, bad_hi = int((lo != CANARY).sum()), int((hi != CANARY).sum())
    assert bad_lo == 0 and bad_hi == 0, (f"{what}: a kernel wrote outside its output buffer of {n} floats: {bad_lo} guard words before it, "
                                         f"{bad_hi} behind it were overwritten")


def check_guards(clear=True):
    """Every buffer dev_empty_f32 has handed out since the last call (conftest runs this after each GPU test)."""
    torch.cuda.synchronize()
    try:
        for whole, n in _live:
            _check(whole, n, "guard bands")
    finally:
        if clear:
            _live.clear()


def ptr(t):
    return t.data_ptr()


def to_host(t):
    torch.cuda.synchronize()
    for whole, n in _live:
        if whole.data_ptr() + 4 * GUARD == t.data_ptr():
            _check(whole, n, "to_host")
            break
    return t.cpu().numpy()
