"""GPU parity of the register-resident systolic decimator (kernels_systolic.hip, round 4): the decimate-by-8, 128-tap, AVX-order
launches that are not launch-bound.  Bit-equal with the oracle's decimateAVXRC restatement (decimate.c:105-113), with the restated
Pipes where seams are on, and with the LDS-tiled kernel it replaces -- whole strips, the ragged last strip, launches cut at
arbitrary outputs, u8 and cfloat input."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal
from oracle import pipes_model as PM
import signals as S
from gpu_util import to_dev, dev_empty_f32, ptr, to_host

pytestmark = pytest.mark.gpu

B = 8192
MIN = 64 * 240 * 4          # the launcher's threshold: fewer outputs stay on the tile kernel


def _dup(h):
    return np.repeat(h, 2)


@pytest.fixture
def counted(hip):
    hip.lib.sdrhip_debug_set_systolic(1)             # 1 = the systolic kernel wherever its shape fits (2, the default, also looks at the launch size)
    before = hip.lib.sdrhip_debug_systolic_launches()
    yield lambda: hip.lib.sdrhip_debug_systolic_launches() - before
    hip.lib.sdrhip_debug_set_systolic(2)             # back to the library's own choice by launch size


@pytest.mark.parametrize("extra", [0, 1, 3, 239, 240, 247, 4 * 240 + 5, 38563])
@pytest.mark.parametrize("u8", [True, False])
def test_one_outputs_match_the_oracle(hip, oracle, counted, extra, u8):
    """seam_block = 0 (every output in SIMD order): whole strips and every kind of ragged end."""
    K = MIN + extra
    n = 8 * (K - 1) + 128
    taps = S.taps_decim127()
    h = np.concatenate([taps, np.zeros(1, np.float32)])
    raw = S.iq_u8(n)
    x = oracle.convert_u8(raw) if u8 else S.cfloat_block(n)
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    out = dev_empty_f32(2 * K + 64)
    out.fill_(float("nan"))
    d_in = to_dev(raw if u8 else x)
    (dec.run_u8 if u8 else dec.run)(ptr(d_in), 0, ptr(out), 0, K, 0)
    got = to_host(out)
    assert counted() == 1, "the systolic kernel did not take this launch"
    assert_bit_equal(got[:2 * K], oracle.decimate_rc(4, K, 8, _dup(h), x), f"K = {K}")
    assert np.isnan(got[2 * K:]).all(), "wrote past the launch's outputs"


@pytest.mark.parametrize("u8", [True, False])
def test_seamed_stream_cut_into_launches(hip, oracle, counted, u8):
    """8192-sample seams (Cross outputs by the fix-up kernel), the stream cut at awkward outputs: each launch has its own x0."""
    nblk = 240           # seamed launches up to 5 * 32768 outputs stay on the tile kernel (their Cross outputs are computed in place)
    raw = S.iq_u8(nblk * B)
    x = oracle.convert_u8(raw) if u8 else S.cfloat_block(nblk * B)
    taps = S.taps_decim127()
    model = PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=8)
    blocks, _ = PM.fir_decimator_pipe(model, [x[2 * i * B: 2 * (i + 1) * B] for i in range(nblk)], 4096)
    exp = np.concatenate(blocks)
    K = exp.size // 2
    assert K > 5 * 32768 + 65536
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    d_in = to_dev(raw if u8 else x)
    out = dev_empty_f32(2 * K)
    for cuts in ([], [2 * 3], [K - 65536]):        # launch starts stay 16-byte aligned (even outputs); the last launch of the third cut is small
        out.fill_(float("nan"))
        edges = [0] + cuts + [K]
        for a, b in zip(edges[:-1], edges[1:]):
            (dec.run_u8 if u8 else dec.run)(ptr(d_in), 0, ptr(out) + 8 * a, a, b, B)
        assert_bit_equal(to_host(out), exp, f"cuts {cuts}")
    assert counted() == 3


@pytest.mark.parametrize("u8", [True, False])
def test_same_bits_as_the_tile_kernel_at_2_to_the_24(hip, counted, u8):
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(11)
    if u8:
        d_in = torch.randint(0, 256, (2 * n + 4096,), dtype=torch.uint8, device="cuda", generator=g)
    else:
        d_in = torch.rand(2 * n + 4096, device="cuda", generator=g) * 2 - 1
    taps = S.taps_decim127()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    K = (n - 128) // 8 + 1
    outs = []
    for on in (1, 0, 2):          # round 5's form (fix-up as a second launch), the tile kernel, the library's own choice (fix-up workgroups inside the launch)
        hip.lib.sdrhip_debug_set_systolic(on)
        out = dev_empty_f32(2 * K)
        (dec.run_u8 if u8 else dec.run)(ptr(d_in), 0, ptr(out), 0, K, B)
        torch.cuda.synchronize()
        outs.append(out)
    assert counted() == 2
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(outs[2].view(torch.int32), outs[1].view(torch.int32)), "seam fix-up inside the systolic launch"


def test_random_launch_geometry_equals_the_tile_kernel(hip):
    """Seeded sweep: random first output, output count, seam block (0, 8192, other multiples of 8 above the filter length) and input
    type -- the systolic kernel against the LDS-tiled kernel on the very same launch (the tile kernel is pinned to the oracle by the
    rest of the suite).  Launch starts stay 16-byte aligned (even outputs), as both kernels require."""
    import os
    scale = max(1, int(os.environ.get("SDRHIP_SWEEP_SCALE", "1")))
    rng = np.random.default_rng(20260928 + int(os.environ.get("SDRHIP_SWEEP_SEED", "0")))
    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(3)
    d_u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
    d_cf = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
    taps = S.taps_decim127()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    kmax = (n - 128) // 8 + 1
    took = 0
    for trial in range(24 * scale):
        u8 = bool(rng.integers(0, 2))
        count = int(rng.integers(MIN, 4 * MIN))
        k0 = 2 * int(rng.integers(0, (kmax - count) // 2))
        seam = int(rng.choice([0, 0, B, B, 8 * int(rng.integers(40, 3000))]))
        if seam and count <= 5 * 32768:
            count = 5 * 32768 + 2 * int(rng.integers(1, 5000))        # seamed launches below that stay on the tile kernel by design
            k0 = min(k0, (kmax - count) // 2 * 2)
        outs = []
        before = hip.lib.sdrhip_debug_systolic_launches()
        for on in (1, 0, 2):      # 2: the library's own choice -- seam fix-up inside the launch where the seam is a multiple of 8 of at least 2048 samples
            hip.lib.sdrhip_debug_set_systolic(on)
            out = torch.full((2 * count + 8,), float("nan"), device="cuda")
            (dec.run_u8 if u8 else dec.run)(ptr(d_u8 if u8 else d_cf), 0, ptr(out), k0, k0 + count, seam)
            torch.cuda.synchronize()
            outs.append(out)
        hip.lib.sdrhip_debug_set_systolic(2)
        took += hip.lib.sdrhip_debug_systolic_launches() - before
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (trial, u8, count, k0, seam)
        assert torch.equal(outs[2].view(torch.int32), outs[1].view(torch.int32)), ("in-launch fix-up", trial, u8, count, k0, seam)
        assert torch.isnan(outs[0][2 * count:]).all() and torch.isnan(outs[2][2 * count:]).all()
    assert took == 2 * 24 * scale


def test_the_example_52_tap_filter_on_the_64_tap_instantiation(hip, oracle):
    """Round 6: the reference example's own RF decimation filter (51 taps -> 52, examples/fm/Coeffs.hs as data) takes the systolic kernel's
    64-tap instantiation on u8 input (12 taps of padding skipped; strips of 248 outputs, 6 Cross outputs per boundary): against the tile
    kernel on random launch geometries -- seams 0 / 8192 / odd multiples of 8, launches inside and past the in-launch fix-up's range -- and
    against the oracle's decimateAVXRC on one whole-stream launch."""
    import os
    scale = max(1, int(os.environ.get("SDRHIP_SWEEP_SCALE", "1")))
    rng = np.random.default_rng(5206 + int(os.environ.get("SDRHIP_SWEEP_SEED", "0")))
    taps = S.taps_example_rf_decim()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    n = 1 << 23
    g = torch.Generator(device="cuda").manual_seed(52)
    d_u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
    kmax = (n - 52) // 8 + 1
    took = 0
    for trial in range(16 * scale):
        count = int(rng.integers(MIN, 8 * MIN))
        seam = int(rng.choice([0, B, B, 8 * int(rng.integers(40, 3000))]))
        if seam and count <= 5 * 32768:
            count = 5 * 32768 + 2 * int(rng.integers(1, 5000))
        k0 = 2 * int(rng.integers(0, (kmax - count) // 2))
        outs = []
        before = hip.lib.sdrhip_debug_systolic_launches()
        for mode in (2, 0):
            hip.lib.sdrhip_debug_set_systolic(mode)
            out = torch.full((2 * count + 8,), float("nan"), device="cuda")
            dec.run_u8(ptr(d_u8), 0, ptr(out), k0, k0 + count, seam)
            torch.cuda.synchronize()
            outs.append(out)
        hip.lib.sdrhip_debug_set_systolic(2)
        took += hip.lib.sdrhip_debug_systolic_launches() - before
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (trial, count, k0, seam)
        assert torch.isnan(outs[0][2 * count:]).all()
    assert took == 16 * scale, "the 64-tap systolic instantiation did not take the launches"
    K = MIN + 777
    nn = 8 * (K - 1) + 52
    raw = S.iq_u8(nn)
    out = dev_empty_f32(2 * K)
    dec.run_u8(ptr(to_dev(raw)), 0, ptr(out), 0, K, 0)
    h = np.concatenate([taps, np.zeros(1, np.float32)])
    assert_bit_equal(to_host(out), oracle.decimate_rc(4, K, 8, _dup(h), oracle.convert_u8(raw)), "52-tap systolic vs the oracle's decimateAVXRC")
