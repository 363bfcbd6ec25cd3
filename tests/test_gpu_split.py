"""GPU parity of the general LDS-tiled ("lane-split") kernels (kernels_split.hip): every SIMD order of
the decimator / filter / resampler families at sizes where the stream API takes the tiled path, against the restated
Pipes (oracle/pipes_model.py) -- One outputs in the SIMD lane order, seam straddlers in the sequential order."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import pipes_model as PM
import signals as S
from gpu_util import to_dev, dev_empty_f32, ptr, to_host

pytestmark = pytest.mark.gpu

# SDRHIP_SWEEP_SCALE=20 turns the seeded random sweeps into a soak test (more trials, same seeds first)
SWEEP_SCALE = max(1, int(__import__("os").environ.get("SDRHIP_SWEEP_SCALE", "1")))
SWEEP_SEED = int(__import__("os").environ.get("SDRHIP_SWEEP_SEED", "0"))        # other seeds for soak runs

B = 8192
NBLK = 24


def _split(x, width, block):
    n = x.size // width
    return [x[i * block * width:(i + 1) * block * width] for i in range(n // block)]


def _run(desc, d_in, out_width, K, seam, cuts=(), out_block=0):
    """out_block: the block size the model Pipe was given (resamplers only: one corner of the seam rule depends on it)."""
    out = dev_empty_f32(K * out_width)
    # (round 5: the guard bands of gpu_util found cut lists reaching past K on the short cases -- launches that wrote behind the buffer)
    edges = sorted(set([0] + [c for c in cuts if 0 < c < K] + [K]))
    kw = {"out_block": out_block} if out_block and hasattr(desc, "in_offset") else {}
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            desc.run(ptr(d_in), 0, ptr(out) + 4 * out_width * a, a, b, seam, **kw)
    return to_host(out)


def _tiled(hip):
    return hip.lib.sdrhip_debug_tiled_launches()


_PARTIALS = {(PM.ORDER_AVX, False): 8, (PM.ORDER_SSE, False): 4,      # real: 8 / 4 lanes (common.h:34-72)
             (PM.ORDER_AVX, True): 4, (PM.ORDER_SSE, True): 2}        # complex RC: 4 / 2 complex lanes (decimate.c:84-113)


@pytest.fixture(autouse=True)
def _tiled_route(hip):
    """These tests are about the tiled kernels: switch off the one-launch route that short seamed launches of the real
    filters / resamplers take by default (sdrhip_set_small_launch_outputs; tests/test_gpu_stream.py runs both routes)."""
    prev = hip.set_small_launch_outputs(0)
    yield
    hip.set_small_launch_outputs(prev)


def _fits(taps_walked, order, complex_, resampler=False):
    """The tiled kernel keeps one tap per partial-sum chunk in registers: at most 64 chunks."""
    m = _PARTIALS[(order, complex_)]
    if resampler and complex_:
        m *= 2                                                       # the RC2 order: 8 / 4 partials (common.h:108-155)
    return taps_walked // m <= 64


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("complex_", [False, True])
@pytest.mark.parametrize("factor,ntaps", [(2, 77), (4, 128), (5, 31), (8, 77), (10, 200), (16, 127), (32, 64)])
def test_decimator_families(hip, oracle, order, complex_, factor, ntaps):
    w = 2 if complex_ else 1
    x = S.cfloat_block(NBLK * B) if complex_ else S.real_block(NBLK * B)
    taps = S.gauss_taps(ntaps, factor + ntaps)
    model = PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, w, B), 1024)
    exp = np.concatenate(blocks)
    K = exp.size // w
    assert K >= 4096
    d = hip.Decimator(factor, taps, order, complex_=complex_)
    # complex decimators by 4 / 8 / 16 with up to 128 (256) taps have their own kernel (k_decimate_c4, exact or guarded;
    # the SSE order through kernels_fast_orders.hip)
    special = (factor in (4, 8, 16) and order in (PM.ORDER_AVX, PM.ORDER_SSE) and complex_ and d.num_coeffs % 4 == 0
               and factor < d.num_coeffs <= (128 if factor == 4 else 256))
    # real decimators by 2 / 4 / 8 / 16: kernels_decimate_real.hip
    real16 = not complex_ and factor in (2, 4, 8, 16)
    before, before16 = _tiled(hip), hip.lib.sdrhip_debug_decimate_real16_launches()
    got = _run(d, to_dev(x), w, K, B)
    if real16:
        assert hip.lib.sdrhip_debug_decimate_real16_launches() > before16, "the real decimator's own kernel did not take this launch"
    elif _fits(d.num_coeffs, order, complex_) and not special:
        assert _tiled(hip) > before, "the tiled kernel did not take this launch"
    assert_bit_equal(got, exp, "one launch")
    got = _run(d, to_dev(x), w, K, B, cuts=[4097, K - 4099])
    assert_bit_equal(got, exp, "three launches")
    # lone buffer: no seams at all
    got0 = _run(d, to_dev(x), w, K, 0)
    seq = PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor)
    assert_bit_equal(got0[: w * 16], seq.one(16, x), "contiguous (all One)")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("factor", [2, 4, 8, 16])
@pytest.mark.parametrize("ntaps", [20, 28, 37, 128, 300, 1000])
def test_real_decimator16(hip, oracle, order, factor, ntaps):
    """Every instantiation of kernels_decimate_real.hip: filters shorter than one 16-tap step, with every remainder of the
    rolled walk (SSE: 4 / 8 / 12 taps, AVX: 8), launches cut at odd outputs (a thread owns 16 / factor consecutive ones) and
    at float offsets that are not 16-byte aligned, with and without seams (decimate.c:36-66 for One outputs,
    FilterInternal.hs:397-402 at seams)."""
    x = S.real_block(NBLK * B)
    taps = S.gauss_taps(ntaps, 7 * factor + ntaps)
    model = PM.FilterModel(oracle, taps, order, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, 1, B), 1024)
    exp = np.concatenate(blocks)
    K = exp.size
    d = hip.Decimator(factor, taps, order)
    before = hip.lib.sdrhip_debug_decimate_real16_launches()
    got = _run(d, to_dev(x), 1, K, B)
    if d.num_coeffs >= 8:
        assert hip.lib.sdrhip_debug_decimate_real16_launches() > before, "the real decimator's own kernel did not take this launch"
    assert_bit_equal(got, exp, f"/{factor}, {ntaps} taps: one launch")
    got = _run(d, to_dev(x), 1, K, B, cuts=[4097, 4097 + 4099, 4097 + 4099 + 4101, K - 4103])
    assert_bit_equal(got, exp, f"/{factor}, {ntaps} taps: cut into launches")
    # one contiguous buffer (no seams), read from a float offset that is not 16-byte aligned
    model1 = PM.FilterModel(oracle, taps, order, factor=factor)
    exp1 = model1.one((x.size - 3 - d.num_coeffs) // factor + 1, x[3:])
    dx = to_dev(x)
    out = dev_empty_f32(exp1.size)
    d.run(ptr(dx) + 12, 0, ptr(out), 0, exp1.size, 0)
    assert_bit_equal(to_host(out), exp1, f"/{factor}, {ntaps} taps: no seams, unaligned input")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("factor", [2, 4, 8, 16])
@pytest.mark.parametrize("nhalf", [8, 16, 24, 64, 200])
def test_real_decimator16_symmetric(hip, oracle, order, factor, nhalf):
    """The symmetric form of kernels_decimate_real.hip (decimateAVXSymmetricRR / decimateSSESymmetricRR, decimate.c:53-83): the
    pair x[j] + x[2N-1-j] is added first; filters of one half step, whole steps and steps + 8, launches cut at odd outputs,
    with and without seams, unaligned input."""
    if 2 * nhalf <= factor:
        pytest.skip("the reference's Pipe asserts on a filter no longer than the decimation")
    x = S.real_block(NBLK * B)
    half = S.gauss_taps(nhalf, 11 * factor + nhalf)
    model = PM.FilterModel(oracle, half, order, sym=True, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, 1, B), 1024)
    exp = np.concatenate(blocks)
    K = exp.size
    d = hip.Decimator(factor, half, order, sym=True)
    before = hip.lib.sdrhip_debug_decimate_real16_launches()
    got = _run(d, to_dev(x), 1, K, B)
    assert hip.lib.sdrhip_debug_decimate_real16_launches() > before, "the real decimator's own kernel did not take this launch"
    assert_bit_equal(got, exp, f"sym /{factor}, {nhalf} half-taps: one launch")
    got = _run(d, to_dev(x), 1, K, B, cuts=[4097, 4097 + 4099, 4097 + 4099 + 4101, K - 4103])
    assert_bit_equal(got, exp, f"sym /{factor}, {nhalf} half-taps: cut into launches")
    model1 = PM.FilterModel(oracle, half, order, sym=True, factor=factor)
    exp1 = model1.one((x.size - 3 - 2 * nhalf) // factor + 1, x[3:])
    dx = to_dev(x)
    out = dev_empty_f32(exp1.size)
    d.run(ptr(dx) + 12, 0, ptr(out), 0, exp1.size, 0)
    assert_bit_equal(to_host(out), exp1, f"sym /{factor}, {nhalf} half-taps: no seams, unaligned input")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("factor,nhalf", [(1, 32), (2, 64), (4, 24), (8, 64)])
def test_symmetric_real_families(hip, oracle, order, factor, nhalf):
    # symmetric FILTERS with a multiple of 8 half-taps have their own kernel (k_fir_real8_fast: AVX order, and since round 3 the
    # SSE order with four lane partials); they are still compared here, only the "general tiled kernel ran" assertion is dropped
    own_kernel = factor == 1 and nhalf % 8 == 0
    # symmetric DECIMATORS by 2 / 4 / 8 / 16 with a multiple of 8 half-taps: kernels_decimate_real.hip (round 3)
    own_kernel = own_kernel or (factor in (2, 4, 8, 16) and nhalf % 8 == 0)
    if order == PM.ORDER_AVX and factor == 1:
        pytest.skip("the AVX symmetric filter has its own kernel (k_fir_real8_fast)")
    x = S.real_block(NBLK * B)
    half = S.gauss_taps(nhalf, 900 + nhalf)
    model = PM.FilterModel(oracle, half, order, sym=True, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, 1, B), 1024)
    exp = np.concatenate(blocks)
    d = hip.Decimator(factor, half, order, sym=True) if factor > 1 else hip.Filter(half, order, sym=True)
    before = _tiled(hip)
    got = _run(d, to_dev(x), 1, exp.size, B)
    assert own_kernel or _tiled(hip) > before
    assert_bit_equal(got, exp, "symmetric")


@pytest.mark.parametrize("order", [PM.ORDER_SSE])
@pytest.mark.parametrize("complex_", [False, True])
def test_filter_sse_orders(hip, oracle, order, complex_):
    w = 2 if complex_ else 1
    x = S.cfloat_block(4 * B) if complex_ else S.real_block(4 * B)
    taps = S.gauss_taps(77, 5)
    model = PM.FilterModel(oracle, taps, order, complex_=complex_)
    blocks, _ = PM.fir_filter_pipe(model, _split(x, w, B), 1024)
    exp = np.concatenate(blocks)
    f = hip.Filter(taps, order, complex_=complex_)
    before = _tiled(hip)
    got = _run(f, to_dev(x), w, exp.size // w, B, cuts=[5000])
    # 77 taps pad to 80 under the SSE rule: a multiple of 8, so the REAL filter takes k_fir_real8_fast<.., 4> (round 3)
    assert (not complex_ and f.num_coeffs % 8 == 0) or _tiled(hip) > before
    assert_bit_equal(got, exp, "SSE-order filter")
    if not complex_:
        taps2 = S.gauss_taps(75, 6)                       # pads to 76: not a multiple of 8 -> the general tiled kernel
        model2 = PM.FilterModel(oracle, taps2, order)
        blocks2, _ = PM.fir_filter_pipe(model2, _split(x, w, B), 1024)
        f2 = hip.Filter(taps2, order)
        before = _tiled(hip)
        got2 = _run(f2, to_dev(x), w, np.concatenate(blocks2).size, B, cuts=[5000])
        assert _tiled(hip) > before
        assert_bit_equal(got2, np.concatenate(blocks2), "SSE-order filter, 76 taps")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("complex_", [False, True])
@pytest.mark.parametrize("I,D,ntaps", [(3, 10, 191), (2, 3, 150), (5, 7, 191), (7, 11, 100), (3, 23, 150), (1, 4, 64), (1, 8, 120), (1, 2, 45), (4, 6, 90)])
def test_resampler_families(hip, oracle, order, complex_, I, D, ntaps):
    w = 2 if complex_ else 1
    x = S.cfloat_block(NBLK * B) if complex_ else S.real_block(NBLK * B)
    taps = S.gauss_taps(ntaps, 10 * I + D)
    if I == 4:
        pytest.skip("gcd(I, D) != 1: the reference's own wrapper assumes numGroups == I (SURVEY.md A8)")
    model = PM.ResamplerModel(oracle, I, D, taps, order, complex_)
    blocks, _ = PM.fir_resampler_pipe(model, _split(x, w, B), 512)
    exp = np.concatenate(blocks)
    K = exp.size // w
    assert K >= 4096
    r = hip.Resampler(I, D, taps, order, complex_)
    # 3/10 with 64-tap groups has specialised kernels: real AVX / SSE (k_resample3_fast), complex AVX / SSE (k_resample3c_fast)
    is_special = (I, D) == (3, 10) and 185 <= ntaps <= 192
    # real I/D with an odd decimation 3 / 5 / 7 has the thread-per-cycle kernel (kernels_resample_cycle.hip), both lane orders
    is_cycle = D in (3, 5, 7)                            # complex data too (the RC2 orders), round 3
    # interpolation 1 and decimation 2 / 4 / 8 / 16: the real decimator's kernel (one polyphase group, the same lane order)
    is_decim = not complex_ and I == 1 and D in (2, 4, 8, 16)
    before, before_cycle, before16 = _tiled(hip), hip.lib.sdrhip_debug_resample_cycle_launches(), hip.lib.sdrhip_debug_decimate_real16_launches()
    got = _run(r, to_dev(x), w, K, B, out_block=512)
    if is_decim:
        assert hip.lib.sdrhip_debug_decimate_real16_launches() > before16, "the real decimator's kernel did not take this launch"
    elif is_cycle:
        assert hip.lib.sdrhip_debug_resample_cycle_launches() > before_cycle, "the thread-per-cycle kernel did not take this launch"
    elif not is_special:
        assert _tiled(hip) > before, "the tiled kernel did not take this launch"
    assert_bit_equal(got, exp, "one launch")
    got = _run(r, to_dev(x), w, K, B, cuts=[4099, 4099 + 4097, K - 5000], out_block=512)
    assert_bit_equal(got, exp, "cut into launches (every starting group)")


@pytest.mark.parametrize("complex_", [False, True])
@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
@pytest.mark.parametrize("I,D", [(1, 3), (2, 3), (1, 5), (2, 5), (3, 5), (4, 5), (2, 7), (3, 7), (4, 7), (5, 7), (6, 7)])
@pytest.mark.parametrize("ntaps", [37, 150, 700])
def test_cycle_resampler(hip, oracle, order, I, D, ntaps, complex_):
    """Every instantiation of the thread-per-cycle kernel (kernels_resample_cycle.hip), real and complex data: short / medium /
    long filters (a single step of the rolled walk, an odd and an even number of steps, the SSE half step), launches cut at
    every starting group, against the restated Pipe (resample.c:52-87 / :106-142 for One outputs, FilterInternal.hs:410-423 at
    seams)."""
    if complex_ and (I, D) == (6, 7):
        pytest.skip("complex 6/7 stays on the lane-split kernel (250+ registers)")
    w = 2 if complex_ else 1
    x = S.cfloat_block(NBLK * B) if complex_ else S.real_block(NBLK * B)
    taps = S.gauss_taps(ntaps, 100 * I + D + ntaps)
    model = PM.ResamplerModel(oracle, I, D, taps, order, complex_)
    blocks, _ = PM.fir_resampler_pipe(model, _split(x, w, B), 512)
    exp = np.concatenate(blocks)
    K = exp.size // w
    r = hip.Resampler(I, D, taps, order, complex_)
    before = hip.lib.sdrhip_debug_resample_cycle_launches()
    got = _run(r, to_dev(x), w, K, B, out_block=512)
    groups_taps = -(-ntaps // I)
    lanes = 8 if order == PM.ORDER_AVX else 4
    if -(-groups_taps // lanes) * lanes >= 8:
        assert hip.lib.sdrhip_debug_resample_cycle_launches() > before, "the thread-per-cycle kernel did not take this launch"
    assert_bit_equal(got, exp, f"{I}/{D}, {ntaps} taps: one launch")
    cuts = [4099 + q for q in range(I)]
    cuts = [c + 4100 * q for q, c in enumerate(cuts)] + [K - 4500]
    got = _run(r, to_dev(x), w, K, B, cuts=[c for c in cuts if 0 < c < K], out_block=512)
    assert_bit_equal(got, exp, f"{I}/{D}, {ntaps} taps: cut into launches")
    # no seams (one contiguous buffer)
    model1 = PM.ResamplerModel(oracle, I, D, taps, order, complex_)
    blocks1, _ = PM.fir_resampler_pipe(model1, [x], 512)
    exp1 = np.concatenate(blocks1)
    got1 = _run(r, to_dev(x), w, exp1.size // w, 0, out_block=512)
    assert_bit_equal(got1, exp1, f"{I}/{D}, {ntaps} taps: no seams")


@pytest.mark.parametrize("complex_", [False, True])
@pytest.mark.parametrize("factor,ntaps", [(64, 128), (128, 256), (100, 192)])
def test_large_factor_tiles(hip, oracle, complex_, factor, ntaps):
    """Large decimation factors: few cycles fit a tile (the 64 KiB span / one-output-per-lane path), or none (fallback)."""
    w = 2 if complex_ else 1
    nblk = (4200 * factor + ntaps) // B + 2
    x = S.cfloat_block(nblk * B, seed=31) if complex_ else S.real_block(nblk * B, seed=32)
    taps = S.gauss_taps(ntaps, factor)
    model = PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=complex_, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, w, B), 512)
    exp = np.concatenate(blocks)
    K = exp.size // w
    assert K >= 4096
    d = hip.Decimator(factor, taps, PM.ORDER_AVX, complex_=complex_)
    got = _run(d, to_dev(x), w, K, B, cuts=[4099])
    assert_bit_equal(got, exp, f"factor {factor}")


def test_random_sweep(hip, oracle):
    """Seeded random configurations: family, order, factor / ratio, tap count, launch cuts and seam block."""
    rng = np.random.default_rng(20260928 + SWEEP_SEED)
    ran = 0
    for trial in range(90 * SWEEP_SCALE):
        complex_ = bool(rng.integers(0, 2))
        order = [PM.ORDER_AVX, PM.ORDER_SSE][rng.integers(0, 2)]
        w = 2 if complex_ else 1
        seam = int(rng.choice([2048, 4096, 8192]))
        kind = rng.integers(0, 3)
        nblk = 160 * 8192 // seam // 4
        x = S.cfloat_block(nblk * seam, seed=100 + trial) if complex_ else S.real_block(nblk * seam, seed=100 + trial)
        try:
            blocks, desc, label, w, x = _random_case(hip, oracle, rng, trial, kind, complex_, order, w, seam, nblk, x)
        except PM.PipeAssert:
            continue                                               # the reference's own Pipe rejects this shape (Filter.hs asserts)
        if not blocks:
            continue
        exp = np.concatenate(blocks)
        K = exp.size // w
        cuts = sorted(int(c) for c in rng.integers(1, K, size=2)) if K > 3 else []
        got = _run(desc, to_dev(x), w, K, seam, cuts=cuts, out_block=512)
        assert_bit_equal(got, exp, label)
        ran += 1
    assert ran >= 60


def _random_case(hip, oracle, rng, trial, kind, complex_, order, w, seam, nblk, x):
    if True:
        if kind == 0:                                              # decimator
            factor = int(rng.integers(1, 24))
            ntaps = int(rng.integers(3, min(300, seam // 2)))
            taps = S.gauss_taps(ntaps, 7000 + trial)
            model = PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor)
            blocks, _ = PM.fir_decimator_pipe(model, _split(x, w, seam), 512)
            desc = hip.Decimator(factor, taps, order, complex_=complex_)
            label = f"trial {trial}: decimate /{factor}, {ntaps} taps, order {order}, complex {complex_}, seam {seam}"
        elif kind == 1:                                            # symmetric real decimator / filter
            complex_, w = False, 1
            x = S.real_block(nblk * seam, seed=100 + trial)
            factor = int(rng.integers(1, 9))
            nhalf = 8 * int(rng.integers(1, 20))
            half = S.gauss_taps(nhalf, 8000 + trial)
            model = PM.FilterModel(oracle, half, order, sym=True, factor=factor)
            blocks, _ = PM.fir_decimator_pipe(model, _split(x, 1, seam), 512)
            desc = hip.Decimator(factor, half, order, sym=True) if factor > 1 else hip.Filter(half, order, sym=True)
            label = f"trial {trial}: symmetric /{factor}, {nhalf} half taps, order {order}, seam {seam}"
        else:                                                      # resampler, gcd(I, D) = 1
            while True:
                I, D = int(rng.integers(1, 9)), int(rng.integers(2, 30))
                if D > I and np.gcd(I, D) == 1:
                    break
            ntaps = int(rng.integers(I + 1, 40 * I))
            if -(-ntaps // (I * 4)) * (I * 4) < D:                 # padded filter shorter than the decimation step:
                ntaps = D + int(rng.integers(0, 20 * I))           # the reference Pipe mis-steps there (see test below)
            taps = S.gauss_taps(ntaps, 9000 + trial)
            model = PM.ResamplerModel(oracle, I, D, taps, order, complex_)
            blocks, _ = PM.fir_resampler_pipe(model, _split(x, w, seam), 512)
            desc = hip.Resampler(I, D, taps, order, complex_)
            label = f"trial {trial}: resample {I}/{D}, {ntaps} taps, order {order}, complex {complex_}, seam {seam}"
    return blocks, desc, label, w, x


def test_degenerate_resampler_is_refused(hip):
    """Padded filter shorter than the decimation step: the reference's Pipe drops past the end of its buffer and loses its
    place (Filter.hs:702-709), so there is no blocked-stream result to reproduce; a single buffer is still fine."""
    r = hip.Resampler(1, 29, S.gauss_taps(16, 1), hip.ORDER_AVX)
    x = to_dev(S.real_block(4 * B))
    out = dev_empty_f32(1024)
    r.run(ptr(x), 0, ptr(out), 0, 1000, 0)                          # contiguous: allowed
    with pytest.raises(hip.SdrHipError):
        r.run(ptr(x), 0, ptr(out), 0, 1000, B)
    with pytest.raises(hip.SdrHipError):
        hip.firResampler(r, 512)
