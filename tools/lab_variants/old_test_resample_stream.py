"""Round 5: fmDemod + the 3/10 resampler as a streaming kernel (sdr_amd/csrc/kernels_resample_stream.hip) against the tile kernel
with fmDemod in its loader and against the stage kernels -- the reference arithmetic is Demod.hs:21-46 and resample.c:70-87
(resampleAVXRR), and both of those paths are pinned to the oracle elsewhere (test_gpu_chain.py, test_gpu_stream.py); here every
cut of a run into workgroups and tiles, both 16-byte phases of the complex stream, seams and no seams, the stream start (carried
sample 0) and the guarded last tiles must give the same bits."""
import os

import numpy as np
import pytest
import torch

import signals as S
from conftest import assert_bit_equal
from gpu_util import ptr

pytestmark = pytest.mark.gpu
B = 8192


def _chain(hip, block):
    chain = hip.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, block)
    chain.set_fused_tail(0)
    chain.set_small_chain(0)
    chain.set_decim_demod_fusion(False)
    return chain


def _run(hip, chain, u8, s0, n, mode, fused=True):
    hip.lib.sdrhip_debug_set_resample_demod_stream(mode)
    chain.set_demod_fusion(fused)
    q0, q1, _ = chain.plan(s0, s0 + n, s0 + n)
    ws = torch.empty(chain.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ws.fill_(0xA5)
    out = torch.full((q1 - q0 + 64,), float("nan"), dtype=torch.float32, device="cuda")       # guard band behind the outputs
    before = hip.lib.sdrhip_debug_resample_demod_stream_launches()
    chain.run(ptr(u8), s0, n, ptr(out), q0, q1, ptr(ws), ws.numel())
    torch.cuda.synchronize()
    took = hip.lib.sdrhip_debug_resample_demod_stream_launches() - before
    assert torch.isnan(out[q1 - q0:]).all(), "wrote past the last output"
    return out[: q1 - q0], took


@pytest.fixture()
def restore(hip):
    yield
    hip.lib.sdrhip_debug_set_resample_demod_stream(int(os.environ.get("SDRHIP_RESAMP_STREAM", "0")))


# prefetch in registers (mode = workgroups) and in LDS by global_load_lds (mode = 1000 + workgroups)
DMA = [0, 1000]


@pytest.mark.parametrize("dma", DMA)
@pytest.mark.parametrize("block", [B, 0])
@pytest.mark.parametrize("s0_blocks, log2n, slots", [(0, 21, 2), (0, 22, 7), (0, 23, 64), (37, 23, 5), (3, 24, 3), (1, 25, 1024)])
def test_stream_equals_tile_kernel(hip, restore, block, s0_blocks, log2n, slots, dma):
    slots += dma
    n = (1 << log2n) + 8 * 1237          # not a multiple of anything convenient
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    chain = _chain(hip, block)
    s0 = s0_blocks * B
    ref, took0 = _run(hip, chain, u8, s0, n, 0)
    assert took0 == 0
    stage, _ = _run(hip, chain, u8, s0, n, 0, fused=False)
    assert torch.equal(ref.view(torch.int32), stage.view(torch.int32))
    got, took = _run(hip, chain, u8, s0, n, slots)
    assert took >= 1, "the streaming kernel did not take the run"
    ref, took0 = _run(hip, chain, u8, s0, n, 0)           # (mode 0 again: the tile kernel, whatever the run's length)
    assert took0 == 0
    bad = torch.nonzero(ref.view(torch.int32) != got.view(torch.int32))
    assert bad.numel() == 0, f"{bad.numel()} audio samples differ, first at {int(bad[0])} of {ref.numel()}"


@pytest.mark.parametrize("dma", DMA)
@pytest.mark.parametrize("offset", [0, 1, 2, 3, 5, 8])
def test_stream_both_phases_of_the_complex_stream(hip, restore, offset, dma):
    """The run's first input sits at an even or an odd complex sample of the decimator's buffer depending on where the pass starts:
    both instantiations (E = 0, 1) must occur; starts that are not block multiples move the seams through the tiles."""
    n = (1 << 22) + 80 * 977
    u8 = torch.randint(0, 256, (2 * (n + 64),), dtype=torch.uint8, device="cuda")
    chain = _chain(hip, B)
    s0 = 11 * B + 8 * offset
    ref, _ = _run(hip, chain, u8, s0, n, 0)
    for slots in (3 + dma, 200 + dma):
        got, took = _run(hip, chain, u8, s0, n, slots)
        assert took >= 1
        assert torch.equal(ref.view(torch.int32), got.view(torch.int32)), f"offset {offset}, {slots} workgroups"


def test_stream_patchy_signal_takes_the_full_form(hip, restore):
    """Silence (0/0), DC (atan2's axis clauses) and noise cut at odd places: waves of the common case, of the full form and mixed."""
    n = 1 << 23
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(5)
    edges = np.sort(rng.integers(0, n, 300)) * 2
    for k in range(0, len(edges) - 1, 2):
        a, b = int(edges[k]), int(edges[k + 1])
        if k % 4 == 0:
            u8[a:b] = 128
        else:
            u8[a:b:2] = int(rng.integers(0, 256))
            u8[a + 1:b:2] = int(rng.integers(0, 256))
    for block in (B, 0):
        chain = _chain(hip, block)
        ref, _ = _run(hip, chain, u8, 0, n, 0, fused=False)
        for slots in (2, 33, 1002, 1033):
            got, took = _run(hip, chain, u8, 0, n, slots)
            assert took >= 1
            assert torch.equal(ref.view(torch.int32), got.view(torch.int32))


def test_stream_against_the_oracle(hip, restore, oracle):
    """One run straight against the restated Pipes (oracle/pipes_model.py): u8 IQ -> audio, 8192-sample seams."""
    from oracle import pipes_model as PM
    nblk = 230            # more than 65536 outputs: a shorter seamed run goes to the generic kernels (abi_device.cpp: small launches)
    u8 = S.iq_u8_fm(nblk * B)
    blocks = [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10, S.taps_audio_half64(), 0.2, B))
    chain = _chain(hip, B)
    d = torch.from_numpy(u8).cuda()
    for mode in (6, 1006):
        got, took = _run(hip, chain, d, 0, nblk * B, mode)
        assert took >= 1
        assert_bit_equal(got.cpu().numpy()[: exp.size], exp, f"streaming fmDemod + resampler (mode {mode}) vs the oracle")


def test_stream_plan_covers_every_cycle(hip):
    import ctypes as C
    for ncycles in (1, 255, 256, 257, 1000, 419431, 6710886, 20132659):
        for cus in (1, 8, 256, 304):
            nt, per, grid = C.c_int(), C.c_int(), C.c_int()
            hip.lib.sdrhip_debug_resample_demod_stream_plan(ncycles, cus, C.byref(nt), C.byref(per), C.byref(grid))
            assert nt.value == (ncycles + 255) // 256
            assert per.value * grid.value >= nt.value > per.value * (grid.value - 1)
