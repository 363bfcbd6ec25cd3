"""GPU parity, layer (3): the host-block Pipe operators (firFilter / firDecimator /
firResampler / fmDemod) against the restated reference Pipes, including ragged block
sizes and the reference's assert on too-short blocks."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import pipes_model as PM
import signals as S

pytestmark = pytest.mark.gpu

# SDRHIP_SWEEP_SCALE=20 turns the seeded random sweeps into a soak test (more trials, same seeds first)
SWEEP_SCALE = max(1, int(__import__("os").environ.get("SDRHIP_SWEEP_SCALE", "1")))
SWEEP_SEED = int(__import__("os").environ.get("SDRHIP_SWEEP_SEED", "0"))        # other seeds for soak runs

B = 8192


def _drive(pipe, blocks):
    outs = []
    for b in blocks:
        outs += pipe.push(b)
    outs += pipe.flush()
    return outs


def _cmp(got, exp, what):
    assert len(got) == len(exp), f"{what}: {len(got)} blocks vs {len(exp)}"
    for i, (g, e) in enumerate(zip(got, exp)):
        assert_bit_equal(g, e, f"{what} block {i}")


def _cut(x, width, sizes):
    out, pos = [], 0
    for s in sizes:
        out.append(x[pos * width:(pos + s) * width])
        pos += s
    return out


def test_fir_decimator_pipe(hip, oracle):
    x = oracle.convert_u8(S.iq_u8(10 * B))
    blocks = _cut(x, 2, [B] * 10)
    taps = S.taps_decim127()
    exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=8), blocks, B)
    dec = hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True)
    _cmp(_drive(hip.firDecimator(dec, B), blocks), exp, "firDecimator")
    assert len(exp) == 1 and exp[0].size == 2 * B


def test_fir_decimator_pipe_ragged(hip, oracle):
    sizes = [4096, 8192, 1000, 20000, 777, 8192, 129, 5000]
    x = S.cfloat_block(sum(sizes))
    blocks = _cut(x, 2, sizes)
    taps = S.gauss_taps(100, 11)
    for order in (PM.ORDER_AVX, PM.ORDER_SCALAR):
        exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, taps, order, complex_=True, factor=5), blocks, 700)
        dec = hip.Decimator(5, taps, order, complex_=True)
        _cmp(_drive(hip.firDecimator(dec, 700), blocks), exp, "ragged firDecimator")


def test_fir_filter_pipe(hip, oracle):
    x = S.real_block(6 * B)
    blocks = _cut(x, 1, [B] * 6)
    half = S.taps_audio_half64()
    exp, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, half, PM.ORDER_AVX, sym=True), blocks, B)
    f = hip.Filter(half, hip.ORDER_AVX, sym=True)
    _cmp(_drive(hip.firFilter(f, B), blocks), exp, "firFilter sym")
    taps = S.gauss_taps(51, 2)
    exp, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, taps, PM.ORDER_SSE), blocks, 1000)
    f = hip.Filter(taps, hip.ORDER_SSE)
    _cmp(_drive(hip.firFilter(f, 1000), blocks), exp, "firFilter SSE")


def test_fir_resampler_pipe(hip, oracle):
    x = S.real_block(8 * B)
    taps = S.taps_resamp191()
    for sizes, bo in (([B] * 8, B), ([3000, 9000, 200, 8192, 8192, 300, 4000, 12000], 999)):
        blocks = _cut(x, 1, sizes)
        exp, _ = PM.fir_resampler_pipe(PM.ResamplerModel(oracle, 3, 10, taps, PM.ORDER_AVX), blocks, bo)
        r = hip.Resampler(3, 10, taps, hip.ORDER_AVX)
        _cmp(_drive(hip.firResampler(r, bo), blocks), exp, f"firResampler {sizes[0]}")


def test_fm_demod_pipe(hip, oracle):
    x = oracle.convert_u8(S.iq_u8_fm(3 * B + 100))
    blocks = _cut(x, 2, [B, 100, B, B])
    exp = PM.fm_demod_pipe(oracle, blocks)
    _cmp(_drive(hip.fmDemod(), blocks), exp, "fmDemod")


@pytest.mark.parametrize("adaptive", [32, 0])
def test_map_pipes_random_blocks(hip, oracle, adaptive):
    """fmDemod and dcBlockingFilter Pipes on ragged block sequences pushed as fast as Python can (adaptive submission: blocks
    pile up while the GPU is busy and leave as one run over their concatenation; 0: every block its own run), with pauses and
    polls in between: one output block per input block, the carry crossing block and run boundaries
    (Demod.hs:40-46; Filter.hs:730-739)."""
    import time
    rng = np.random.default_rng(91 + SWEEP_SEED)
    for trial in range(6 * SWEEP_SCALE):
        nblk = int(rng.integers(5, 120))
        sizes = [int(rng.choice([1, 7, 100, 1024, 4096, 8192, 8192, 8192, 20000])) for _ in range(nblk)]
        if trial % 3 == 0:
            sizes = [8192] * nblk
        total = sum(sizes)
        x = oracle.convert_u8(S.iq_u8_fm(total, seed=300 + trial))
        blocks = _cut(x, 2, sizes)
        exp = PM.fm_demod_pipe(oracle, blocks)
        pipe = hip.fmDemod()
        pipe.set_adaptive(adaptive)
        got = []
        for i, b in enumerate(blocks):
            got += pipe.push(b)
            if i % 17 == 16:
                time.sleep(0.001)
                got += pipe.poll()
        got += pipe.flush()
        _cmp(got, exp, f"fmDemod trial {trial} ({nblk} blocks)")
        # dcBlockingFilter on a real signal with an offset
        xr = (S.real_block(total, seed=400 + trial) * 0.5 + 0.25).astype(np.float32)
        e_all, _, _ = oracle.dc_blocker(xr, 0.0, 0.0)
        pipe = hip.dcBlockingFilter()
        pipe.set_adaptive(adaptive)
        got = []
        for i, b in enumerate(_cut(xr, 1, sizes)):
            got += pipe.push(b)
            if i % 13 == 12:
                time.sleep(0.001)
                got += pipe.poll()
        got += pipe.flush()
        assert [g.size for g in got] == sizes, f"dcBlockingFilter trial {trial}: block lengths"
        assert_bit_equal(np.concatenate(got), e_all, f"dcBlockingFilter trial {trial}")


def test_pipe_poll_delivers_without_another_push(hip, oracle):
    """sdrhip_pipe_poll: the output of the block just pushed is there a moment later, no further push, no flush."""
    import time
    x = oracle.convert_u8(S.iq_u8(12 * B))
    blocks = _cut(x, 2, [B] * 12)
    taps = S.taps_decim127()
    exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, taps, PM.ORDER_AVX, complex_=True, factor=8), blocks, 512)
    pipe = hip.firDecimator(hip.Decimator(8, taps, hip.ORDER_AVX, complex_=True), 512)
    got = []
    for i, b in enumerate(blocks):
        got += pipe.push(b)
        want = (((i + 1) * B - 128) // 8 + 1) // 512          # whole 512-output blocks computable from the samples so far
        deadline = time.time() + 2.0
        while len(got) < want and time.time() < deadline:
            got += pipe.poll()
        assert len(got) >= want, f"block {i}: poll delivered {len(got)} of {want} blocks"
    got += pipe.flush()
    _cmp(got, exp, "polled firDecimator")


def test_convert_operator(hip, oracle):
    u8 = S.iq_u8(B)
    got = hip.interleavedIQUnsignedByteToFloatFast(u8)
    assert got.dtype == np.complex64 and got.size == B
    assert_bit_equal(got.view(np.float32), oracle.convert_u8(u8), "interleavedIQUnsignedByteToFloatFast")


def test_pipe_short_block_asserts(hip, oracle):
    """Filter.hs:544 `assert "filter 1"`: a buffer shorter than the filter is an error."""
    f = hip.Filter(S.gauss_taps(128, 1), hip.ORDER_AVX)
    p = hip.firFilter(f, 256)
    with pytest.raises(hip.SdrHipError):
        p.push(S.real_block(100))
    with pytest.raises(PM.PipeAssert):
        PM.fir_filter_pipe(PM.FilterModel(oracle, S.gauss_taps(128, 1), PM.ORDER_AVX), [S.real_block(100)], 256)


@pytest.mark.parametrize("coalesce", [0, 4])
def test_fm_receiver_as_composed_pipes(hip, oracle, coalesce):
    """fm.hs:34-41 composed from the four Pipe operators, host blocks end to end (INTEGRATION.md level 1), push by push
    and with the operators coalescing their equal-sized input blocks."""
    nblk = 100 if coalesce else 60
    u8 = S.iq_u8_fm(nblk * B)
    blocks = [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)]
    exp = PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10, S.taps_audio_half64(), 0.2)
    assert len(exp) >= 1
    deci = hip.firDecimator(hip.Decimator(8, S.taps_decim127(), hip.ORDER_AVX, complex_=True), B)
    demod = hip.fmDemod()
    resp = hip.firResampler(hip.Resampler(3, 10, S.taps_resamp191(), hip.ORDER_AVX), B)
    filt = hip.firFilter(hip.Filter(S.taps_audio_half64(), hip.ORDER_AVX, sym=True), B)
    if coalesce:
        for pipe in (deci, resp, filt):
            pipe.set_coalesce(coalesce)
    audio = []

    def feed(stage_idx, blk, stages):
        if stage_idx == len(stages):
            audio.append(hip.DropIn.scale("scaleAVX", 0.2, blk))
            return
        for o in stages[stage_idx](blk):
            feed(stage_idx + 1, o, stages)

    stages = [lambda b: deci.push(hip.interleavedIQUnsignedByteToFloatFast(b).view(np.float32)),
              demod.push, resp.push, filt.push]
    for b in blocks:
        feed(0, b, stages)
    # drain the double-buffer lag, upstream first
    flushers = [deci.flush, demod.flush, resp.flush, filt.flush]
    for i, fl in enumerate(flushers):
        for o in fl():
            feed(i + 1, o, stages)
    _cmp(audio, exp, "composed FM receiver")


def test_resampler_pipe_random_sweep(hip, oracle):
    """Seeded random resampler Pipes on ragged host blocks, short filters included: where the first output that no longer
    fits a block already starts in the next one the reference Pipe does not cross over (Filter.hs:707-709)."""
    rng = np.random.default_rng(77 + SWEEP_SEED)
    ran = 0
    for trial in range(40 * SWEEP_SCALE):
        complex_ = bool(rng.integers(0, 2))
        order = [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR][rng.integers(0, 3)]
        w = 2 if complex_ else 1
        while True:
            I, D = int(rng.integers(1, 7)), int(rng.integers(2, 26))
            if D > I and np.gcd(I, D) == 1:
                break
        simd = {PM.ORDER_AVX: 8, PM.ORDER_SSE: 4, PM.ORDER_SCALAR: 1}[order]
        ntaps = int(rng.integers(I + 1, 30 * I))
        if -(-ntaps // (I * simd)) * (I * simd) < D:
            ntaps = D + int(rng.integers(0, 10))
        taps = S.gauss_taps(ntaps, 500 + trial)
        lp = -(-ntaps // (I * simd)) * (I * simd)
        min_block = -(-lp // I) + D
        sizes = [int(s) for s in rng.integers(min_block, min_block + 3000, size=12)]
        x = S.cfloat_block(sum(sizes), seed=600 + trial) if complex_ else S.real_block(sum(sizes), seed=600 + trial)
        blocks = _cut(x, w, sizes)
        bso = int(rng.integers(50, 900))
        try:
            exp, _ = PM.fir_resampler_pipe(PM.ResamplerModel(oracle, I, D, taps, order, complex_), blocks, bso)
        except PM.PipeAssert:
            continue
        r = hip.Resampler(I, D, taps, order, complex_)
        _cmp(_drive(hip.firResampler(r, bso), blocks), exp,
             f"trial {trial}: {I}/{D}, {ntaps} taps, order {order}, complex {complex_}, blocks {sizes[:4]}..")
        ran += 1
    assert ran >= 25


def test_decimator_pipe_random_sweep(hip, oracle):
    rng = np.random.default_rng(78 + SWEEP_SEED)
    ran = 0
    for trial in range(30 * SWEEP_SCALE):
        complex_ = bool(rng.integers(0, 2))
        order = [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR][rng.integers(0, 3)]
        w = 2 if complex_ else 1
        ntaps = int(rng.integers(2, 200))
        factor = int(rng.integers(1, min(ntaps, 20) + 1))
        taps = S.gauss_taps(ntaps, 700 + trial)
        sizes = [int(s) for s in rng.integers(ntaps + 8 + factor, ntaps + 4000, size=10)]
        x = S.cfloat_block(sum(sizes), seed=800 + trial) if complex_ else S.real_block(sum(sizes), seed=800 + trial)
        blocks = _cut(x, w, sizes)
        bso = int(rng.integers(50, 900))
        try:
            exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor), blocks, bso)
        except PM.PipeAssert:
            continue
        d = hip.Decimator(factor, taps, order, complex_=complex_)
        _cmp(_drive(hip.firDecimator(d, bso), blocks), exp, f"trial {trial}: /{factor}, {ntaps} taps, order {order}, complex {complex_}")
        ran += 1
    assert ran >= 20


def test_filter_pipe_random_sweep(hip, oracle):
    """firFilter on ragged host blocks: plain real / complex and symmetric real filters, all orders."""
    rng = np.random.default_rng(79 + SWEEP_SEED)
    ran = 0
    for trial in range(30 * SWEEP_SCALE):
        order = [PM.ORDER_AVX, PM.ORDER_SSE, PM.ORDER_SCALAR][rng.integers(0, 3)]
        kind = rng.integers(0, 3)                                   # 0 real, 1 complex, 2 symmetric real
        if kind == 2 and order == PM.ORDER_SCALAR:
            order = PM.ORDER_SSE                                    # the reference has no scalar symmetric kernels
        complex_ = kind == 1
        w = 2 if complex_ else 1
        if kind == 2:
            simd = 8 if order == PM.ORDER_AVX else 4
            taps = S.gauss_taps(simd * int(rng.integers(1, 16)), 900 + trial)
            full = 2 * taps.size
        else:
            taps = S.gauss_taps(int(rng.integers(2, 200)), 900 + trial)
            full = taps.size + 8
        sizes = [int(s) for s in rng.integers(full + 8, full + 5000, size=10)]
        x = S.cfloat_block(sum(sizes), seed=950 + trial) if complex_ else S.real_block(sum(sizes), seed=950 + trial)
        blocks = _cut(x, w, sizes)
        bso = int(rng.integers(50, 1500))
        try:
            exp, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, taps, order, complex_=complex_, sym=(kind == 2)), blocks, bso)
        except PM.PipeAssert:
            continue
        f = hip.Filter(taps, order, complex_=complex_, sym=(kind == 2))
        _cmp(_drive(hip.firFilter(f, bso), blocks), exp, f"trial {trial}: kind {kind}, {taps.size} taps, order {order}")
        ran += 1
    assert ran >= 20


def test_uniform_block_pipes_coalesced_sweep(hip, oracle):
    """Equal-sized blocks (the reference's normal diet): every push on its own, coalesced k at a time, and through the
    zero-copy staging buffer -- then a block of another size ends the uniform run.  Same output blocks every way."""
    rng = np.random.default_rng(80 + SWEEP_SEED)
    ran = 0
    for trial in range(30 * SWEEP_SCALE):
        kind = ["decimator", "resampler", "symfilter"][rng.integers(0, 3)]
        order = [PM.ORDER_AVX, PM.ORDER_SSE][rng.integers(0, 2)]
        complex_ = bool(rng.integers(0, 2)) and kind != "symfilter"
        w = 2 if complex_ else 1
        U = int(rng.choice([1024, 2048, 4096, 8192]))
        nblk = int(rng.integers(6, 20))
        ragged_tail = [int(rng.integers(U // 2 + 300, 2 * U))] if rng.integers(0, 2) else []
        sizes = [U] * nblk + ragged_tail + ([U] * 2 if ragged_tail else [])
        x = S.cfloat_block(sum(sizes), seed=1200 + trial) if complex_ else S.real_block(sum(sizes), seed=1200 + trial)
        blocks = _cut(x, w, sizes)
        bso = int(rng.integers(100, 3000))
        try:
            if kind == "decimator":
                taps = S.gauss_taps(int(rng.integers(4, 200)), 1300 + trial)
                factor = int(rng.integers(1, min(taps.size, 16) + 1))
                exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, taps, order, complex_=complex_, factor=factor), blocks, bso)
                mk = lambda: hip.firDecimator(hip.Decimator(factor, taps, order, complex_=complex_), bso)
            elif kind == "resampler":
                while True:
                    I, D = int(rng.integers(1, 7)), int(rng.integers(2, 24))
                    if D > I and np.gcd(I, D) == 1:
                        break
                simd = 8 if order == PM.ORDER_AVX else 4
                ntaps = int(rng.integers(I + 1, 30 * I))
                if -(-ntaps // (I * simd)) * (I * simd) < D:
                    ntaps = D + int(rng.integers(0, 10))
                taps = S.gauss_taps(ntaps, 1300 + trial)
                exp, _ = PM.fir_resampler_pipe(PM.ResamplerModel(oracle, I, D, taps, order, complex_), blocks, bso)
                mk = lambda: hip.firResampler(hip.Resampler(I, D, taps, order, complex_), bso)
            else:
                simd = 8 if order == PM.ORDER_AVX else 4
                half = S.gauss_taps(simd * int(rng.integers(1, 12)), 1300 + trial)
                exp, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, half, order, sym=True), blocks, bso)
                mk = lambda: hip.firFilter(hip.Filter(half, order, sym=True), bso)
        except PM.PipeAssert:
            continue
        label = f"trial {trial}: {kind}, order {order}, complex {complex_}, U {U}, {nblk} blocks, tail {ragged_tail}"
        _cmp(_drive(mk(), blocks), exp, label + " (push by push)")
        pipe = mk()
        pipe.set_coalesce(int(rng.integers(2, 9)))
        got = []
        for i, b in enumerate(blocks):
            if i % 3 == 1:
                view = pipe.input_buffer(b.size // w)
                view[:] = b
                got += pipe.push(view)
            else:
                got += pipe.push(b)
        got += pipe.flush()
        _cmp(got, exp, label + " (coalesced, zero-copy every third push)")
        # adaptive submission: how the pushes are grouped depends on how busy the GPU is, the blocks do not
        pipe = mk()
        pipe.set_adaptive(int(rng.integers(2, 40)))
        got = []
        for i, b in enumerate(blocks):
            if i % 4 == 2:
                view = pipe.input_buffer(b.size // w)
                view[:] = b
                got += pipe.push(view)
            else:
                got += pipe.push(b)
        got += pipe.flush()
        _cmp(got, exp, label + " (adaptive submission, zero-copy every fourth push)")
        ran += 1
    assert ran >= 20


def test_short_block_through_a_coalescing_pipes_lent_buffer(hip, oracle):
    """coalesce > 1 with blocks already staged, the caller takes input_buffer(n) and then pushes FEWER elements
    through it (a short last block).  The short block ends the uniform run, the staged blocks are submitted and the Pipe
    moves to its other staging slot: the block must be copied there, not assumed to be in place."""
    t127 = S.taps_decim127()
    for n_short in (4096, 1000, 8192 - 8):
        sizes = [8192, 8192, n_short, 8192, 8192]
        xc = S.cfloat_block(sum(sizes))
        blocks = _cut(xc, 2, sizes)
        exp, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, t127, PM.ORDER_AVX, complex_=True, factor=8), blocks, 1024)
        pipe = hip.firDecimator(hip.Decimator(8, t127, hip.ORDER_AVX, complex_=True), 1024)
        pipe.set_coalesce(4)
        got = []
        got += pipe.push(blocks[0])
        got += pipe.push(blocks[1])
        view = pipe.input_buffer(8192)                      # room for a whole block ...
        view[: 2 * n_short] = blocks[2]                     # ... of which only the front is filled and pushed
        view[2 * n_short:] = np.float32(1e30)               # anything read past the pushed part would show
        got += pipe.push(view[: 2 * n_short])
        got += pipe.push(blocks[3])
        got += pipe.push(blocks[4])
        got += pipe.flush()
        _cmp(got, exp, f"short block of {n_short} through the lent buffer of a coalescing firDecimator")
    # the same on a real filter Pipe
    half = S.taps_audio_half64()
    sizes = [8192, 3000, 8192]
    xr = S.real_block(sum(sizes))
    blocks = _cut(xr, 1, sizes)
    exp, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, half, PM.ORDER_AVX, sym=True), blocks, B)
    pipe = hip.firFilter(hip.Filter(half, hip.ORDER_AVX, sym=True), B)
    pipe.set_coalesce(8)
    got = pipe.push(blocks[0])
    view = pipe.input_buffer(8192)
    view[:3000] = blocks[1]
    got += pipe.push(view[:3000])
    got += pipe.push(blocks[2])
    got += pipe.flush()
    _cmp(got, exp, "short block through the lent buffer of a coalescing firFilter")


def test_pipes_cross_the_in_place_threshold(hip, oracle):
    """Small pushes run in place (kernels read / write the pinned buffers over PCIe), large ones go through the copy
    engines; the carried tail comes from the host-side history either way.  Pushes on both sides of the threshold, in both
    orders, and the 65536-float block of BASELINE configs[3] (in place) -- same blocks as the restated Pipe."""
    taps = S.taps_resamp191()
    sizes = [65536, 200000, 65536, 8192, 300000, 65536, 131072, 140000, 4096, 65536]          # 512 KiB = 131072 floats
    x = S.real_block(sum(sizes))
    blocks = _cut(x, 1, sizes)
    exp, _ = PM.fir_resampler_pipe(PM.ResamplerModel(oracle, 3, 10, taps, PM.ORDER_AVX), blocks, B)
    r = hip.Resampler(3, 10, taps, hip.ORDER_AVX)
    _cmp(_drive(hip.firResampler(r, B), blocks), exp, "firResampler across the in-place threshold")
    # complex decimator: 8192-sample blocks in place, a 100000-sample block through the copy path
    xc = S.cfloat_block(8192 * 3 + 100000 + 8192 * 2)
    csizes = [8192, 8192, 100000, 8192, 8192, 8192]
    cblocks = _cut(xc, 2, csizes)
    t127 = S.taps_decim127()
    expd, _ = PM.fir_decimator_pipe(PM.FilterModel(oracle, t127, PM.ORDER_AVX, complex_=True, factor=8), cblocks, 1024)
    d = hip.Decimator(8, t127, hip.ORDER_AVX, complex_=True)
    _cmp(_drive(hip.firDecimator(d, 1024), cblocks), expd, "firDecimator across the in-place threshold")


def test_pipes_save_and_restore(hip, oracle):
    """Checkpoint / resume of every pipe kind (sdrhip_pipe_save / _restore): a pipe saved mid-stream (ragged blocks, output
    pending), destroyed, restored into a fresh pipe over a fresh descriptor of the same taps, and fed the rest, yields the
    uninterrupted pipe's blocks."""
    sizes = [4096, 8192, 1000, 20000, 777, 8192, 129, 5000, 8192, 3000]
    xc = S.cfloat_block(sum(sizes))
    xr = S.real_block(sum(sizes))
    taps = S.gauss_taps(100, 11)
    half = S.taps_audio_half64()
    rtaps = S.taps_resamp191()
    makers = [
        ("firDecimator", lambda: hip.firDecimator(hip.Decimator(5, taps, hip.ORDER_AVX, complex_=True), 700), xc, 2),
        ("firFilter", lambda: hip.firFilter(hip.Filter(half, hip.ORDER_AVX, sym=True), 1024), xr, 1),
        ("firResampler", lambda: hip.firResampler(hip.Resampler(3, 10, rtaps, hip.ORDER_AVX), 512), xr, 1),
        ("fmDemod", lambda: hip.fmDemod(), xc, 2),
        ("dcBlockingFilter", lambda: hip.Pipe("dc_blocker"), xr, 1),
    ]
    for name, make, x, width in makers:
        blocks = _cut(x, width, sizes)
        exp = _drive(make(), blocks)
        for cut in (1, 4, 7):
            first = make()
            got = []
            for b in blocks[:cut]:
                got += first.push(b)
            state = first.save()
            del first
            second = make()
            got += second.restore(state, max_block=max(sizes))
            for b in blocks[cut:]:
                got += second.push(b)
            got += second.flush()
            _cmp(got, exp, f"{name}, saved after {cut} blocks")
    with pytest.raises(hip.SdrHipError):
        hip.fmDemod().restore(state)          # a dcBlockingFilter state into an fmDemod pipe


def test_save_right_after_large_pushes_and_corrupt_states(hip, oracle):
    """(a) sdrhip_pipe_state_bytes is exact (it drains the pipe as save will), so a save right after pushes of
    blocks far larger than the old heuristic slack -- a map pipe (fmDemod) with 100k-sample blocks, a low-decimation filter --
    succeeds at the first call; (b) a state whose history is shorter than the carried tail is refused by restore instead of
    sending the kernels in front of the staging buffer."""
    x = S.cfloat_block(3 * 100000)
    blocks = _cut(x, 2, [100000, 100000, 100000])
    exp = PM.fm_demod_pipe(oracle, blocks)
    pipe = hip.fmDemod()
    got = pipe.push(blocks[0]) + pipe.push(blocks[1])
    state = pipe.save()                                   # both 100k-sample blocks may still be in flight here; what the drain
                                                          # makes ready travels inside the state
    fresh = hip.fmDemod()
    got2 = fresh.restore(state, max_block=100000)
    got2 += fresh.push(blocks[2]) + fresh.flush()
    _cmp(got + got2, exp, "fmDemod pipe saved right after two 100k-sample pushes")

    half = S.taps_audio_half64()
    xr = S.real_block(3 * 70000)
    rblocks = _cut(xr, 1, [70000, 70000, 70000])
    expf, _ = PM.fir_filter_pipe(PM.FilterModel(oracle, half, PM.ORDER_AVX, sym=True), rblocks, 1024)
    pf = hip.firFilter(hip.Filter(half, hip.ORDER_AVX, sym=True), 1024)
    out = pf.push(rblocks[0]) + pf.push(rblocks[1])
    st = pf.save()
    pf2 = hip.firFilter(hip.Filter(half, hip.ORDER_AVX, sym=True), 1024)
    out2 = pf2.restore(st)
    out2 += pf2.push(rblocks[2]) + pf2.flush()
    _cmp(out + out2, expf, "firFilter pipe saved right after two 70000-float pushes")

    # (b) shrink the history recorded in the state: header = magic, version (u32), 10 x i32, then E_prev, m_done, head_cap, hist_n (i64)
    import struct
    bad = bytearray(st)
    off_hist_n = 4 * 2 + 4 * 10 + 8 * 3
    (hist_n,) = struct.unpack_from("<q", bad, off_hist_n)
    assert hist_n >= 127
    struct.pack_into("<q", bad, off_hist_n, 8)
    pf3 = hip.firFilter(hip.Filter(half, hip.ORDER_AVX, sym=True), 1024)
    with pytest.raises(hip.SdrHipError):
        pf3.restore(bytes(bad))
