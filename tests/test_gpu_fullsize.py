"""GPU parity at BASELINE.json's full sizes, through properties that need no oracle pass over
the whole input: impulse responses (exact: one non-zero sample makes every sum a single product),
shift invariance of the seamed chain, launch-cut invariance, and spot checks of individual outputs
against the oracle evaluated on just their receptive field."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal
from oracle.oracle import duplicate
import signals as S
from gpu_util import to_dev, dev_empty_f32, ptr, to_host

pytestmark = pytest.mark.gpu

B = 8192


def test_decimator_impulse_response_at_full_size(hip):
    """configs[1]/[4] sizes: 2^20 samples per block-of-work; an impulse at sample p gives
    out[k] = h[p - 8k] exactly (u8 128 -> 0.0, 255 -> 127/128)."""
    n = 1 << 24
    taps = S.taps_decim127()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    K = (n - 128) // 8 + 1
    u8 = torch.full((2 * n,), 128, dtype=torch.uint8, device="cuda")
    pos = [5, 8191, 8192, 123457, n - 129, n - 1]
    for p in pos:
        u8[2 * p] = 255          # I = 127/128, Q = 0
    out = dev_empty_f32(2 * K)
    dec.run_u8(ptr(u8), 0, ptr(out), 0, K, 0)
    got = to_host(out).reshape(-1, 2)
    exp = np.zeros((K, 2), np.float32)
    a = np.float32(127.0 / 128.0)
    for p in pos:
        for k in range(max(0, (p - 127 + 7) // 8), min(K, p // 8 + 1)):
            j = p - 8 * k
            if 0 <= j < 127:
                exp[k, 0] += taps[j] * a      # windows of distinct impulses never overlap here
    assert_bit_equal(got.reshape(-1), exp.reshape(-1), "impulse response, contiguous")
    # with seams the straddling outputs use the sequential order; a single product is order-independent
    dec.run_u8(ptr(u8), 0, ptr(out), 0, K, B)
    assert_bit_equal(to_host(out), exp.reshape(-1), "impulse response, 8192-sample seams")


def test_decimator_spot_checks_against_oracle_at_full_size(hip, oracle):
    """2^26 cfloat samples (512 MiB): every 65537th output and the outputs around a few seams against
    the oracle evaluated on just their 128-sample windows."""
    n = 1 << 26
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
    taps = S.taps_decim127()
    h = np.concatenate([taps, np.zeros(1, np.float32)])
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    K = (n - 128) // 8 + 1
    out = dev_empty_f32(2 * K)
    dec.run(ptr(x), 0, ptr(out), 0, K, B)
    torch.cuda.synchronize()
    ks = list(range(0, K, 65537)) + [K - 1]
    for blk in (1, 2, 4097, 8191):
        ks += list(range(blk * 1024 - 20, blk * 1024 + 3))
    for k in ks:
        win = x[16 * k: 16 * k + 256].cpu().numpy()
        cross = (8 * k) // B != (8 * k + 127) // B
        if cross:
            exp = oracle.decimate_cross_c(8, h, 1, win, np.zeros(2, np.float32))
        else:
            exp = oracle.decimate_rc(4, 1, 8, duplicate(h), win)
        got = out[2 * k: 2 * k + 2].cpu().numpy()
        assert_bit_equal(got, exp, f"output {k} ({'cross' if cross else 'one'})")


def test_chain_shift_invariance_at_full_size(hip):
    """The seamed chain repeats itself under an input shift of 8192*80 samples (every stage's seam grid
    maps onto itself: 8192*80/8 = 10*8192 decimator outputs, *3/10 = 3*8192 audio samples).  Only the
    outputs that see the very first demod sample (last = 0) may differ (through a 1e-5 edge tap, so
    often not even those): the first 256 are skipped."""
    n = 1 << 26
    shift, qshift = B * 80, 3 * B
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    chain = hip.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    ws = torch.empty(chain.workspace_bytes(n), dtype=torch.uint8, device="cuda")

    def run(buf, total):
        q0, q1, _ = chain.plan(0, total, total)
        out = dev_empty_f32(q1 - q0)
        chain.run(ptr(buf), 0, total, ptr(out), q0, q1, ptr(ws), ws.numel())
        torch.cuda.synchronize()
        return out
    a = run(u8, n)
    b = run(u8[2 * shift:].contiguous(), n - shift)
    skip = 256
    assert torch.equal(a[qshift + skip: qshift + b.numel()].view(torch.int32), b[skip:].view(torch.int32))


def test_chain_launch_cut_invariance_at_full_size(hip):
    """One launch over 2^26 samples == seven uneven launches over sub-ranges of the same buffer."""
    n = 1 << 26
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    chain = hip.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    ws = torch.empty(chain.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    q0, q1, _ = chain.plan(0, n, n)
    full = dev_empty_f32(q1)
    chain.run(ptr(u8), 0, n, ptr(full), 0, q1, ptr(ws), ws.numel())
    cuts = [0, 1, 777, 24576, 1000003, q1 // 2 + 5, q1 - 1, q1]
    parts = dev_empty_f32(q1)
    for a, b in zip(cuts[:-1], cuts[1:]):
        chain.run(ptr(u8), 0, n, ptr(parts) + 4 * a, a, b, ptr(ws), ws.numel())
    torch.cuda.synchronize()
    assert torch.equal(full.view(torch.int32), parts.view(torch.int32))


def test_resampler_and_filter_impulse_at_full_size(hip):
    """configs[3] size (and beyond): impulses through the 3/10 resampler and the symmetric filter."""
    n = 1 << 22
    h = S.taps_resamp191()
    r = hip.Resampler(3, 10, h, hip.ORDER_AVX)
    M = (n * 3 - 192) // 10 + 1
    x = torch.zeros(n, device="cuda")
    pos = [0, 17, 65535, 65536, 1234567, n - 70]
    x[pos] = 1.0
    out = dev_empty_f32(M)
    r.run(ptr(x), 0, ptr(out), 0, M, 0)
    got = to_host(out)
    exp = np.zeros(M, np.float32)
    for p in pos:
        for m in range(max(0, (3 * p - 191 + 9) // 10), min(M, 3 * p // 10 + 1)):
            j = 3 * p - 10 * m           # tap index: x[p] meets h[fo + 3*(p - inOff)] = h[3p - 10m]
            if 0 <= j < 191:
                exp[m] += h[j]
    assert_bit_equal(got, exp, "resampler impulse response")
    half = S.taps_audio_half64()
    f = hip.Filter(half, hip.ORDER_AVX, sym=True)
    K = n - 127
    out = dev_empty_f32(K)
    f.run(ptr(x), 0, ptr(out), 0, K, B)
    got = to_host(out)
    full = np.concatenate([half, half[::-1]])
    exp = np.zeros(K, np.float32)
    for p in pos:
        for k in range(max(0, p - 127), min(K, p + 1)):
            exp[k] += full[p - k]
    assert_bit_equal(got, exp, "symmetric filter impulse response")


def test_example_shaped_taps_chain(hip, oracle):
    """The tap LENGTHS of the reference FM example (51 / 31 / 32 half-taps, examples/fm/Coeffs.hs) --
    a different padding and polyphase geometry than the benchmark taps."""
    from oracle import pipes_model as PM
    nblk = 90
    u8 = S.iq_u8_fm(nblk * B)
    blocks = [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim51(), 8, S.taps_resamp31(), 3, 10,
                                        S.taps_audio_half32(), 0.2, B))
    chain = hip.FmChain(8, S.taps_decim51(), 3, 10, S.taps_resamp31(), S.taps_audio_half32(), 0.2, B)
    total = nblk * B
    q0, q1, _ = chain.plan(0, total, total)
    ws = torch.empty(chain.workspace_bytes(total), dtype=torch.uint8, device="cuda")
    out = dev_empty_f32(q1)
    chain.run(ptr(to_dev(u8)), 0, total, ptr(out), 0, q1, ptr(ws), ws.numel())
    assert exp.size >= 2 * B
    assert_bit_equal(to_host(out)[: exp.size], exp, "example-shaped chain")


@pytest.mark.parametrize("block", [B, 0])
def test_example_real_taps_full_size_fused_equals_stage_kernels(hip, block):
    """The reference example's own taps at a size where the chain takes its large-batch routes (2^24 samples: the 52-tap tile decimator,
    fmDemod inside the 16-float-group resampler's loader since round 6): the same audio as with fmDemod as a kernel of its own, from the
    stream start and from the middle of a stream."""
    n = 1 << 24
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    chain = hip.FmChain(8, S.taps_example_rf_decim(), 3, 10, S.taps_example_audio_resampler(), S.taps_example_audio_filter_half(), 0.2, block)
    ws = torch.empty(chain.workspace_bytes(n), dtype=torch.uint8, device="cuda")
    for s0 in (0, 53 * B):
        q0, q1, _ = chain.plan(s0, s0 + n, s0 + n)
        outs = []
        for fused in (False, True):
            chain.set_demod_fusion(fused)
            ws.fill_(0x5A)
            out = dev_empty_f32(q1 - q0)
            chain.enable_timing(True)
            chain.run(ptr(u8), s0, n, ptr(out), q0, q1, ptr(ws), ws.numel())
            torch.cuda.synchronize()
            stage_ms, _ = chain.read_timing()
            chain.enable_timing(False)
            assert (stage_ms["fm_demod"] == 0.0) == fused, "the fused form books the pair under `resample`"
            outs.append(out)
        chain.set_demod_fusion(True)
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), f"block {block}, s0 {s0}"


def test_example_real_taps_chain(hip, oracle):
    """The reference FM example's OWN filters (examples/fm/Coeffs.hs:11-154 as data: tests/golden/example_taps.npz -- Octave remez
    designs, 51 / 31 / 32 half-taps) through the whole receiver of examples/fm/fm.hs:34-41, against the restated Pipes, device-resident
    and block by block through the host-block operator."""
    from oracle import pipes_model as PM
    hd, hr, ha = S.taps_example_rf_decim(), S.taps_example_audio_resampler(), S.taps_example_audio_filter_half()
    nblk = 90
    u8 = S.iq_u8_fm(nblk * B)
    blocks = [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, hd, 8, hr, 3, 10, ha, 0.2, B))
    chain = hip.FmChain(8, hd, 3, 10, hr, ha, 0.2, B)
    total = nblk * B
    q0, q1, _ = chain.plan(0, total, total)
    ws = torch.empty(chain.workspace_bytes(total), dtype=torch.uint8, device="cuda")
    out = dev_empty_f32(q1)
    for mode in (0, 2):                         # stage kernels, then the library's own route
        chain.set_small_chain(mode)
        out.zero_()
        chain.run(ptr(to_dev(u8)), 0, total, ptr(out), 0, q1, ptr(ws), ws.numel())
        assert exp.size >= 2 * B
        assert_bit_equal(to_host(out)[: exp.size], exp, f"the example's own taps, device-resident chain (small-chain mode {mode})")
    st = hip.FmStream(chain, B, B)
    outs = []
    for blk in blocks:
        outs += st.push(blk)
    outs += st.flush()
    got = np.concatenate(outs)
    assert got.size >= exp.size
    assert_bit_equal(got[: exp.size], exp, "the example's own taps, host-block stream")


@pytest.mark.parametrize("kind", ["noise", "patchy"])
@pytest.mark.parametrize("start_blocks", [0, 37])
def test_chain_demod_fusion_is_invisible(hip, start_blocks, kind):
    """fmDemod inside the resampler's tile loader (sdrhip_fm_chain_set_demod_fusion): the demodulated stream is then written
    only around seams and launch edges, and every audio sample must still be the stage kernels' -- from the stream start
    (carried sample 0) and from the middle of a stream, with 8192-sample seams and without.  `patchy`: stretches of silence
    (decimator output exactly 0: fmDemod's 0/0 clause), of DC (imaginary part of the product exactly 0: atan2's axis clauses)
    and of noise, cut at odd places -- the fused loader evaluates fmDemod's common case and sends a wave with any other sample
    through the full form (demod.hpp: fm_phase_common), so waves of both kinds and mixed ones must occur."""
    n = 1 << 25
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    if kind == "patchy":
        rng = np.random.default_rng(99 + start_blocks)
        edges = np.sort(rng.integers(0, n, 400)) * 2
        for k in range(0, len(edges) - 1, 2):
            a, b = int(edges[k]), int(edges[k + 1])
            if k % 4 == 0:
                u8[a:b] = 128                                      # silence
            else:
                u8[a:b:2] = int(rng.integers(0, 256))              # DC: one I and one Q value
                u8[a + 1:b:2] = int(rng.integers(0, 256))
    for block in (B, 0):
        chain = hip.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, block)
        chain.set_fused_tail(0)                                    # the stage kernels, whatever the environment asks for
        chain.set_small_chain(0)
        ws = torch.empty(chain.workspace_bytes(n), dtype=torch.uint8, device="cuda")
        s0 = start_blocks * B
        q0, q1, _ = chain.plan(s0, s0 + n, s0 + n)
        assert q1 - q0 > (1 << 20)
        outs = []
        for fused in (False, True, False):
            chain.set_demod_fusion(fused)
            ws.fill_(0xA5)                                         # stale workspace contents must not matter
            out = dev_empty_f32(q1 - q0)
            chain.enable_timing(True)
            chain.run(ptr(u8), s0, n, ptr(out), q0, q1, ptr(ws), ws.numel())
            torch.cuda.synchronize()
            stage_ms, _ = chain.read_timing()
            chain.enable_timing(False)
            assert (stage_ms["fm_demod"] == 0.0) == fused, "the fused form books the pair under `resample`"
            outs.append(out)
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), f"block {block}: fused differs"
        assert torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32))
