"""GPU parity, layer (2): the device stream API against the restated Pipes
(oracle/pipes_model.py): One outputs in SIMD lane order, Cross outputs (seam
straddlers) in sequential order, independent of how the stream is cut into launches."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import pipes_model as PM
import signals as S
from gpu_util import to_dev, dev_empty_f32, ptr, to_host

pytestmark = pytest.mark.gpu

B = 8192


def _split(x, width, block):
    n = x.size // width
    return [x[i * block * width:(i + 1) * block * width] for i in range(n // block)]


def _run_ranges(desc, d_in, in_total, out_width, K, seam, cuts, u8=False, out_block=0):
    """Run [0,K) as several launches cut at `cuts`, each with its own in_base/slice of the input."""
    out = dev_empty_f32(K * out_width)
    edges = [0] + list(cuts) + [K]
    kw = {"out_block": out_block} if out_block else {}
    for a, b in zip(edges[:-1], edges[1:]):
        if b <= a:
            continue
        (desc.run_u8 if u8 else desc.run)(ptr(d_in), 0, ptr(out) + 4 * out_width * a, a, b, seam, **kw)
    return to_host(out)


@pytest.fixture(params=["one launch for short seamed launches", "tiled kernels + seam launch"])
def launch_route(hip, request):
    """Short seamed launches decide their Cross outputs inside one kernel by default (real filter / resampler: the generic
    kernel; tiled complex decimator: in the tile kernel -- sdrhip_set_small_launch_outputs); the tests that feed them run
    under both routes."""
    prev = hip.set_small_launch_outputs(-1 if request.param.startswith("one") else 0)
    yield request.param
    hip.set_small_launch_outputs(prev)


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR])
def test_decimator_complex_stream(hip, oracle, order, launch_route):
    nblk = 5
    u8 = S.iq_u8(nblk * B)
    x = oracle.convert_u8(u8)
    taps = S.taps_decim127()
    model = PM.FilterModel(oracle, taps, order, complex_=True, factor=8)
    blocks, trace = PM.fir_decimator_pipe(model, _split(x, 2, B), 512)
    if order == PM.ORDER_AVX:
        assert trace[:4] == [("one", 512), ("one", 497), ("cross", 15), ("one", 512)]
    exp = np.concatenate(blocks)
    K = exp.size // 2
    dec = hip.Decimator(8, taps, order, complex_=True)
    assert dec.num_coeffs == model.num_coeffs
    got = _run_ranges(dec, to_dev(x), nblk * B, 2, K, B, [])
    assert_bit_equal(got, exp, "contiguous launch")
    got = _run_ranges(dec, to_dev(x), nblk * B, 2, K, B, [1, 1000, 1009, 1024, 3000])
    assert_bit_equal(got, exp, "cut into launches")
    got = _run_ranges(dec, to_dev(u8), nblk * B, 2, K, B, [1024], u8=True)
    assert_bit_equal(got, exp, "u8 input (convert fused)")


def test_decimator_lone_block_is_all_one(hip, oracle):
    """seam_block = 0: what one FFI call on one buffer computes (BASELINE configs[1])."""
    x = S.cfloat_block(B)
    taps = S.taps_decim127()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    got = _run_ranges(dec, to_dev(x), B, 2, 1009, 0, [])
    h = np.concatenate([taps, np.zeros(1, np.float32)])
    assert_bit_equal(got, oracle.decimate_rc(4, 1009, 8, np.repeat(h, 2), x), "lone block")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE])
def test_filter_sym_stream(hip, oracle, order, launch_route):
    x = S.real_block(4 * B)
    half = S.taps_audio_half64()
    model = PM.FilterModel(oracle, half, order, sym=True)
    blocks, trace = PM.fir_filter_pipe(model, _split(x, 1, B), 1024)
    exp = np.concatenate(blocks)
    f = hip.Filter(half, order, sym=True)
    assert f.num_coeffs == 128
    got = _run_ranges(f, to_dev(x), 4 * B, 1, exp.size, B, [8065, 8192, 9000])
    assert_bit_equal(got, exp, "sym filter stream")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR])
@pytest.mark.parametrize("complex_", [False, True])
def test_filter_decimator_generic_stream(hip, oracle, order, complex_, launch_route):
    w = 2 if complex_ else 1
    x = S.cfloat_block(3 * 4096) if complex_ else S.real_block(3 * 4096)
    taps = S.gauss_taps(77, 3)
    for factor in (1, 3, 7):
        model = PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor)
        blocks, _ = PM.fir_decimator_pipe(model, _split(x, w, 4096), 256)
        exp = np.concatenate(blocks)
        d = hip.Decimator(factor, taps, order, complex_=complex_)
        got = _run_ranges(d, to_dev(x), 3 * 4096, w, exp.size // w, 4096, [100])
        assert_bit_equal(got, exp, f"factor {factor}")


@pytest.mark.parametrize("order", [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR])
@pytest.mark.parametrize("complex_", [False, True])
def test_resampler_stream(hip, oracle, order, complex_, launch_route):
    w = 2 if complex_ else 1
    x = S.cfloat_block(4 * B) if complex_ else S.real_block(4 * B)
    taps = S.taps_resamp191()
    model = PM.ResamplerModel(oracle, 3, 10, taps, order, complex_)
    blocks, trace = PM.fir_resampler_pipe(model, _split(x, w, B), 512)
    if order == PM.ORDER_AVX and not complex_:
        per_block = sum(c for k, c in trace[:12] if True)
        assert ("cross", 19) in trace
    exp = np.concatenate(blocks)
    r = hip.Resampler(3, 10, taps, order, complex_)
    assert r.num_coeffs == model.num_coeffs
    got = _run_ranges(r, to_dev(x), 4 * B, w, exp.size // w, B, [], out_block=512)
    assert_bit_equal(got, exp, "contiguous")
    got = _run_ranges(r, to_dev(x), 4 * B, w, exp.size // w, B, [1, 2, 2439, 2458, 5000], out_block=512)
    assert_bit_equal(got, exp, "cut into launches")


@pytest.mark.parametrize("I,D", [(2, 3), (5, 7), (7, 11), (3, 23), (97, 100), (65, 131)])
def test_resampler_other_ratios(hip, oracle, I, D, launch_route):
    """(97, 100) and (65, 131): more than 64 polyphase groups -- the per-group tables then live in device memory."""
    x = S.real_block(3 * 4096)
    taps = S.gauss_taps(150 if I < 64 else 1500, I + D)
    model = PM.ResamplerModel(oracle, I, D, taps, PM.ORDER_AVX)
    blocks, _ = PM.fir_resampler_pipe(model, _split(x, 1, 4096), 128)
    exp = np.concatenate(blocks)
    r = hip.Resampler(I, D, taps, hip.ORDER_AVX)
    for m in (0, 1, 5, 1000):
        assert r.in_offset(m) == -((-m * D) // I)
    got = _run_ranges(r, to_dev(x), 3 * 4096, 1, exp.size, 4096, [77], out_block=128)
    assert_bit_equal(got, exp, f"{I}/{D}")


def test_fm_demod_stream(hip, oracle):
    x = oracle.convert_u8(S.iq_u8_fm(3 * B))
    exp = np.concatenate(PM.fm_demod_pipe(oracle, _split(x, 2, B)))
    d_in = to_dev(x)
    out = dev_empty_f32(3 * B)
    hip.check(hip.lib.sdrhip_fm_demod_run(None, ptr(d_in), 0, ptr(out), 0, 1000, 0.0, 0.0))
    hip.check(hip.lib.sdrhip_fm_demod_run(None, ptr(d_in), 0, ptr(out) + 4000, 1000, 3 * B, 0.0, 0.0))
    assert_bit_equal(to_host(out), exp, "fmDemod stream")


def test_convert_device(hip, oracle):
    u8 = S.iq_u8(100003)[:200001]
    d_in = to_dev(u8)
    out = dev_empty_f32(u8.size)
    hip.check(hip.lib.sdrhip_convert_u8_run(None, ptr(d_in), ptr(out), u8.size))
    assert_bit_equal(to_host(out), oracle.convert_u8(u8), "convert device")


def test_argument_errors(hip):
    taps = S.taps_decim127()
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    d = dev_empty_f32(1024)
    with pytest.raises(hip.SdrHipError):
        dec.run(ptr(d), 100, ptr(d), 0, 10, 0)          # window starts before the buffer
    with pytest.raises(hip.SdrHipError):
        dec.run(ptr(d), 0, ptr(d), 0, 10, 64)           # seam block shorter than the filter
    with pytest.raises(hip.SdrHipError):
        hip.Filter(taps[:30], hip.ORDER_AVX, sym=True)  # half taps not a multiple of 8
    with pytest.raises(hip.SdrHipError):
        hip.Resampler(10, 3, taps)                      # needs decimation > interpolation


def test_shared_descriptor_from_several_host_threads(hip, oracle):
    """A descriptor is immutable after creation except for the lazy tap upload on first use, which is locked: host threads
    may share it, each launching on its own HIP stream (ctypes drops the GIL during the calls)."""
    import threading
    import torch
    x = S.cfloat_block(16 * B)
    taps = S.taps_decim127()
    K = (16 * B - 128) // 8 + 1
    ref_dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    exp = _run_ranges(ref_dec, to_dev(x), 16 * B, 2, K, B, [])
    for attempt in range(4):
        dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)      # fresh: nothing uploaded yet
        d_in = to_dev(x)
        outs = [dev_empty_f32(2 * K) for _ in range(6)]
        streams = [torch.cuda.Stream() for _ in outs]
        torch.cuda.synchronize()
        errs = []

        def work(i):
            try:
                dec.run(ptr(d_in), 0, ptr(outs[i]), 0, K, B, stream=streams[i].cuda_stream)
            except Exception as e:                                      # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(outs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for i, o in enumerate(outs):
            assert_bit_equal(to_host(o), exp, f"attempt {attempt}, thread {i}")


def test_fm_demod_random_bit_patterns(hip, oracle):
    """fmDemod on arbitrary float bit patterns (denormals, infinities, NaNs, huge ratios): the device kernel follows the
    restated GHC formula (Data.Complex multiply, RealFloat atan2, fdlibm atanf) through every special case.  NaN results
    are compared as NaN (payloads are not part of the contract), everything else bit for bit."""
    import torch
    rng = np.random.default_rng(2718)
    n = 1 << 18
    bits = rng.integers(0, 1 << 32, 2 * n, dtype=np.uint64).astype(np.uint32)
    # half of the samples get moderate exponents so that not everything overflows to inf/NaN
    mod = rng.integers(0, 2, 2 * n).astype(bool)
    e = rng.integers(100, 150, 2 * n).astype(np.uint32)
    bits[mod] = (bits[mod] & np.uint32(0x807FFFFF)) | (e[mod] << np.uint32(23))
    x = bits.view(np.float32)
    exp = oracle.fm_demod(x)
    d_in = to_dev(x)
    nan_e = np.isnan(exp)
    ok = ~nan_e
    assert ok.sum() > n // 4
    # the stand-alone kernel: the common case per lane, a wave vote, the full select form behind it (demod.hpp) -- on this input
    # nearly every wave takes the full form
    out = dev_empty_f32(n)
    hip.check(hip.lib.sdrhip_fm_demod_run(None, ptr(d_in), 0, ptr(out), 0, n, 0.0, 0.0))
    got = to_host(out)
    nan_g = np.isnan(got)
    assert np.array_equal(nan_e, nan_g), f"NaN pattern differs at {np.nonzero(nan_e != nan_g)[0][:5]}"
    assert_bit_equal(got[ok], exp[ok], "fmDemod on random bit patterns")
    got2 = hip.DropIn.fm_demod(x)
    assert np.array_equal(np.isnan(got2), nan_e)
    assert_bit_equal(got2[ok], exp[ok], "fmDemodF (drop-in) on random bit patterns")


def test_fm_demod_on_ordinary_and_awkward_signals(hip, oracle):
    """fmDemod on what a receiver sees (an FM signal, noise) and on the inputs that
    leave the common case: zeros of either sign, the axes, denormals, ratios beyond 2^25 and below 2^-29, repeated samples --
    alone in a wave of ordinary samples and in runs longer than a wave."""
    rng = np.random.default_rng(4242)
    n = 1 << 17
    x = rng.uniform(-1, 1, 2 * n).astype(np.float32)
    x[: 2 * 50000] = oracle.convert_u8(S.iq_u8_fm(50000))
    sp = np.array([0, 0, -0.0, 0, 0, -0.0, -0.0, -0.0, 1, 0, -1, 0, 0, 1, 0, -1, -1, -0.0, 1e-40, 1e-40, 1e-30, 1, 1, 1e-30,
                   1e30, 1e-8, -1e-8, 1e30, 0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 3, -4, 1, 1e-9, 1, -1e-9, -1, 1e-9, -1, -1e-9,
                   1e-9, 1, 1e-9, -1, 3e7, 1, 1, 4e7, 1e-38, 1e-38, 1e-38, -1e-38, 1e-45, 0, 1, 1], np.float32)
    for at in (0, 100001, 140000):                                  # lone awkward samples among ordinary ones
        x[at: at + sp.size] = sp
    x[2 * 60000: 2 * 60400] = 0.0                                   # runs longer than a wave
    x[2 * 61000: 2 * 61400] = np.tile(np.array([0.25, -0.75], np.float32), 400)
    x[2 * 62000: 2 * 62400: 2] = 0.0                                # on the imaginary axis
    exp = oracle.fm_demod(x)
    d_in = to_dev(x)
    out = dev_empty_f32(n)
    hip.check(hip.lib.sdrhip_fm_demod_run(None, ptr(d_in), 0, ptr(out), 0, n, 0.0, 0.0))
    assert_bit_equal(to_host(out), exp, "fmDemod")


def test_fm_demod_on_dense_argument_ranges(hip, oracle):
    """atan2's ratio swept on purpose: every second sample is 1 + 0i, so the phases are atan2(+-y, x) of the samples in between.
    Ratios log-uniform over [2^-30, 2^26] (all five argument ranges of fdlibm's atanf, in every quadrant, with power-of-two and
    with random-mantissa denominators), then every range threshold and every ratio that makes a reduced numerator vanish
    (7/16, 1/2, 11/16, 1, 19/16, 3/2, 39/16, and the common case's own limits 2^-29 and 2^25) with their neighbours +-4 ulp --
    kept in a stretch of their own so that the waves of the first part stay in the common case (the forms with a wave vote compute
    those without the full form behind them)."""
    rng = np.random.default_rng(90125)
    m = 1 << 16
    q = np.exp2(rng.uniform(-30, 26, m)).astype(np.float32)
    x = np.where(rng.integers(0, 2, m) == 0, np.exp2(rng.integers(-20, 20, m)), rng.uniform(0.5, 2.0, m) * np.exp2(rng.integers(-20, 20, m))).astype(np.float32)
    x *= rng.choice(np.array([-1.0, 1.0], np.float32), m)
    y = (q * np.abs(x)).astype(np.float32) * rng.choice(np.array([-1.0, 1.0], np.float32), m)
    marks = np.array([2.0 ** -29, 7 / 16, 0.5, 11 / 16, 1.0, 19 / 16, 1.5, 39 / 16, 2.0 ** 25], np.float32)
    edge = []
    for t in marks:
        b = np.float32(t).view(np.uint32)
        for d in range(-4, 5):
            edge.append(np.uint32(int(b) + d).view(np.float32))
    edge = np.array(edge, np.float32)
    ex = np.concatenate([np.full(edge.size, s, np.float32) for s in (1.0, -1.0, 4.0, -0.125)])
    ey = np.concatenate([edge * abs(s) for s in (1.0, -1.0, 4.0, -0.125)]).astype(np.float32)
    ey = np.concatenate([ey, -ey]); ex = np.concatenate([ex, ex])
    pad = (-ex.size) % 1024
    xs = np.concatenate([x, ex, np.ones(pad, np.float32)])
    ys = np.concatenate([y, ey, np.full(pad, 0.75, np.float32)])
    iq = np.zeros(4 * xs.size, np.float32)
    iq[0::4] = 1.0                                                   # d[2i] = 1 + 0i
    iq[2::4] = xs
    iq[3::4] = ys                                                    # d[2i + 1] = x + iy
    exp = oracle.fm_demod(iq)
    d_in = to_dev(iq)
    n = iq.size // 2
    out = dev_empty_f32(n)
    hip.check(hip.lib.sdrhip_fm_demod_run(None, ptr(d_in), 0, ptr(out), 0, n, 0.0, 0.0))
    assert_bit_equal(to_host(out), exp, "fmDemod on swept ratios")


@pytest.mark.parametrize("factor", [8, 4, 16])
@pytest.mark.parametrize("ntaps", [9, 12, 31, 51, 60, 77, 100, 121, 127, 130, 200, 253])
def test_decimator_by_8_any_length_up_to_128(hip, oracle, ntaps, factor, launch_route):
    """The FM chain's decimator kernel serves every tap count up to 128 (exact kernels for 128 and 52, run-time guarded
    blocks otherwise) and the decimation factors 4, 8 and 16: cfloat and u8 input, seams, cut launches -- and it is that
    kernel, not a fallback, that runs."""
    if -(-ntaps // 4) * 4 <= factor:
        pytest.skip("the reference Pipe needs more taps than the decimation step")
    if factor == 4 and ntaps > 128:
        pytest.skip("decimation 4 has no 256-tap instantiation (general tiled kernel)")
    nblk = 12 if factor < 16 else 24
    u8 = S.iq_u8(nblk * B, seed=ntaps)
    x = oracle.convert_u8(u8)
    taps = S.gauss_taps(ntaps, 40 + ntaps)
    model = PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=factor)
    blocks, _ = PM.fir_decimator_pipe(model, _split(x, 2, B), 1024)
    exp = np.concatenate(blocks)
    K = exp.size // 2
    dec = hip.Decimator(factor, taps, hip.ORDER_AVX, complex_=True)
    before = hip.lib.sdrhip_debug_tiled_launches()
    got = _run_ranges(dec, to_dev(x), nblk * B, 2, K, B, [])
    assert hip.lib.sdrhip_debug_tiled_launches() == before, "the general tiled kernel took a launch meant for k_decimate_c4"
    assert_bit_equal(got, exp, "cfloat in")
    assert_bit_equal(_run_ranges(dec, to_dev(x), nblk * B, 2, K, B, [1024, 4096 + 8][: 2 if K > 4200 else 1]), exp, "cfloat in, cut")
    assert_bit_equal(_run_ranges(dec, to_dev(u8), nblk * B, 2, K, B, [2048], u8=True), exp, "u8 in (convert fused)")
    assert_bit_equal(_run_ranges(dec, to_dev(u8), nblk * B, 2, K, 0, [], u8=True)[:64],
                     PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=factor).one(32, x), "u8 in, no seams")


@pytest.mark.parametrize("ntaps", [128, 64, 127, 61])
def test_complex_filter_on_the_tiled_kernel(hip, oracle, ntaps, launch_route):
    """Complex filters of exactly 128 / 64 (padded) taps run on the tiled decimator with D = 1 (kernels_fast_filter.hip); 127 and
    61 taps pad to 128 / 64 under mkFilterC's intended rule.  Large launches (the tile kernel takes >= 16384 outputs), with the
    8192-sample seams of the Pipe and without, cut into launches at odd places."""
    n = 6 * B
    x = S.cfloat_block(n, seed=91)
    taps = S.gauss_taps(ntaps, 300 + ntaps)
    model = PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=1)
    blocks, _ = PM.fir_filter_pipe(model, _split(x, 2, B), 4096)
    exp = np.concatenate(blocks)
    f = hip.Filter(taps, hip.ORDER_AVX, complex_=True)
    got = _run_ranges(f, to_dev(x), n, 2, exp.size // 2, B, [])
    assert_bit_equal(got, exp, f"complex filter {ntaps} taps, one launch")
    got = _run_ranges(f, to_dev(x), n, 2, exp.size // 2, B, [3, 20000, 20001, 40000])
    assert_bit_equal(got, exp, f"complex filter {ntaps} taps, cut into launches")
