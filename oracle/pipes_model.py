"""CPU restatement of the reference's Pipes state machines.  TEST INFRASTRUCTURE ONLY.

Follows hs_sources/SDR/Filter.hs line by line:
  * Buffer / newBuffer / advanceOutBuf            :504-523
  * firFilter    (simple / crossover)              :532-569
  * firDecimator (simple / crossover)              :574-611
  * firResampler (simple / crossover, (group, offset) state)  :679-727
  * the fast* constructors' padding rules         :163-175, 234-245, 277-290, 317-331, 408-425
and hs_sources/SDR/Demod.hs:40-46 for fmDemod.

The within-buffer kernels are the AVX/SSE/scalar C kernels (restated in
sdr_oracle.c, proven bit-exact against the reference build); the cross-buffer
kernels are the sequential Haskell ones (FilterInternal.hs:397-423).

Each `*_pipe` function consumes a list of input blocks and returns the list of
output blocks the Pipe would have yielded (only exactly-full blocks are ever
yielded), plus a trace of (kind, count) kernel calls for bookkeeping tests.
GHC cannot be run in this image, so these models are pinned only by the
reference's own `assert` invariants (Filter.hs:544-720), which are checked here.
"""
import numpy as np

from .oracle import Oracle, duplicate, round_up

ORDER_SCALAR, ORDER_SSE, ORDER_AVX = 0, 1, 2
_REAL_LANES = {ORDER_SCALAR: 1, ORDER_SSE: 4, ORDER_AVX: 8}
_CPLX_LANES = {ORDER_SCALAR: 1, ORDER_SSE: 2, ORDER_AVX: 4}


class PipeAssert(AssertionError):
    """The reference's hand-rolled `assert` (Filter.hs:526-527) firing."""


def _assert(loc, cond):
    if not cond:
        raise PipeAssert(loc)


def quot_up(q, d):
    return (q + d - 1) // d


class _OutBuf:
    """Buffer + advanceOutBuf, Filter.hs:504-523."""

    def __init__(self, block, width):
        self.block = block
        self.width = width
        self.buf = np.empty(block * width, np.float32)
        self.offset = 0
        self.yielded = []

    def space(self):
        return self.block - self.offset

    def write(self, data):
        n = data.size // self.width
        self.buf[self.offset * self.width:(self.offset + n) * self.width] = data
        if n == self.space():
            self.yielded.append(self.buf)
            self.buf = np.empty(self.block * self.width, np.float32)
            self.offset = 0
        else:
            self.offset += n


class FilterModel:
    """Filter / Decimator record (Filter.hs:116-131) built as fastFilter*/fastDecimator* do."""

    def __init__(self, oracle, coeffs, order=ORDER_AVX, complex_=False, sym=False, factor=1):
        self.o = oracle
        self.order, self.complex, self.sym, self.factor = order, complex_, sym, factor
        c = np.asarray(coeffs, np.float32)
        if sym:
            # mkFilterSymR / mkDecimatorSymR, Filter.hs:234-245, 358-371
            self.lanes = _REAL_LANES[order]
            self.one_taps = c
            self.cross_taps = np.concatenate([c, c[::-1]])
            self.num_coeffs = 2 * c.size
        elif complex_:
            # mkDecimatorC, Filter.hs:322-331 (mkFilterC intends the same, see sdr_amd/csrc/abi_device.cpp)
            self.lanes = _CPLX_LANES[order]
            self.num_coeffs = round_up(c.size, self.lanes)
            padded = np.concatenate([c, np.zeros(self.num_coeffs - c.size, np.float32)])
            self.one_taps = padded if order == ORDER_SCALAR else duplicate(padded)
            self.cross_taps = padded
        else:
            # mkFilter / mkDecimator, Filter.hs:167-175, 282-290
            self.lanes = _REAL_LANES[order]
            self.num_coeffs = round_up(c.size, self.lanes)
            padded = np.concatenate([c, np.zeros(self.num_coeffs - c.size, np.float32)])
            self.one_taps = padded
            self.cross_taps = padded
        self.width = 2 if complex_ else 1

    def one(self, count, buf):
        if self.complex:
            return self.o.decimate_rc(self.lanes, count, self.factor, self.one_taps, buf)
        if self.sym:
            return self.o.decimate_sym_rr(self.lanes, count, self.factor, self.one_taps, buf)
        return self.o.decimate_rr(self.lanes, count, self.factor, self.one_taps, buf)

    def cross(self, count, last, nxt):
        if self.complex:
            return self.o.decimate_cross_c(self.factor, self.cross_taps, count, last, nxt)
        return self.o.decimate_cross_r(self.factor, self.cross_taps, count, last, nxt)


def fir_decimator_pipe(model, blocks, block_size_out):
    """firDecimator, Filter.hs:574-611 (firFilter :532-569 is factor == 1)."""
    w, D, L = model.width, model.factor, model.num_coeffs
    out = _OutBuf(block_size_out, w)
    trace = []
    it = iter(blocks)

    def length(b):
        return b.size // w

    try:
        buf_in = np.asarray(next(it), np.float32)
    except StopIteration:
        return [], trace
    state = "simple"
    buf_last = buf_next = None
    while True:
        if state == "simple":
            _assert("decimate 1", length(buf_in) >= L)
            count = min((length(buf_in) - L) // D + 1, out.space())
            out.write(model.one(count, buf_in))
            trace.append(("one", count))
            buf_in = buf_in[count * D * w:]
            if length(buf_in) >= L:
                continue
            try:
                nxt = np.asarray(next(it), np.float32)
            except StopIteration:
                break
            # Filter.hs:594-598: an empty remainder still goes through crossover in the
            # reference only if non-empty is asserted ("decimate 3"); a zero-length
            # remainder can occur when (len - L) is a multiple of D and L == D.
            buf_last, buf_next, state = buf_in, nxt, "cross"
        else:
            _assert("decimate 2", length(buf_last) < L)
            _assert("decimate 3", length(buf_last) > 0)
            count = min(quot_up(length(buf_last), D), out.space())
            out.write(model.cross(count, buf_last, buf_next))
            trace.append(("cross", count))
            if length(buf_last) <= count * D:
                buf_in = buf_next[(count * D - length(buf_last)) * w:]
                state = "simple"
            else:
                buf_last = buf_last[count * D * w:]
    return out.yielded, trace


def fir_filter_pipe(model, blocks, block_size_out):
    return fir_decimator_pipe(model, blocks, block_size_out)


class ResamplerModel:
    """Resampler record built as fastResampler{C,SSE,AVX}{R,C} do (Filter.hs:408-446)."""

    def __init__(self, oracle, interpolation, decimation, coeffs, order=ORDER_AVX, complex_=False):
        self.o = oracle
        self.I, self.D = interpolation, decimation
        self.complex = complex_
        self.coeffs = np.asarray(coeffs, np.float32)
        self.simd = _REAL_LANES[order]
        self.lanes = _CPLX_LANES[order] if complex_ else _REAL_LANES[order]
        self.prep = oracle.prepare_coeffs(self.simd, interpolation, decimation, self.coeffs)
        self.num_coeffs = round_up(self.coeffs.size, interpolation * self.simd)  # numCoeffsR, Filter.hs:422
        self.width = 2 if complex_ else 1

    def one(self, dat, count, buf):
        group = dat[0]
        if self.complex:
            res, g = self.o.resample_rc(self.lanes, count, self.prep, group, buf)
        else:
            res, g = self.o.resample_rr(self.lanes, count, self.prep, group, buf)
        # func1, Filter.hs:423
        offset = self.I - 1 - ((self.I + g * self.D - 1) % self.I)
        return res, (g, offset), offset

    def cross(self, dat, count, last, nxt):
        group, offset = dat
        if self.complex:
            res, off2 = self.o.resample_cross_c(self.I, self.D, self.coeffs, offset, count, last, nxt)
        else:
            res, off2 = self.o.resample_cross_r(self.I, self.D, self.coeffs, offset, count, last, nxt)
        return res, ((group + count) % self.I, off2), off2  # Filter.hs:419-421


def fir_resampler_pipe(model, blocks, block_size_out):
    """firResampler, Filter.hs:679-727."""
    w, I, D, L = model.width, model.I, model.D, model.num_coeffs
    out = _OutBuf(block_size_out, w)
    trace = []
    it = iter(blocks)

    def length(b):
        return b.size // w

    try:
        buf_in = np.asarray(next(it), np.float32)
    except StopIteration:
        return [], trace
    dat, filter_offset = (0, 0), 0
    state = "simple"
    buf_last = buf_next = None
    while True:
        if state == "simple":
            _assert("resample 1", length(buf_in) * I >= L - filter_offset)
            count = min((length(buf_in) * I - L + filter_offset) // D + 1, out.space())
            res, dat, end_offset = model.one(dat, count, buf_in)
            _assert("resample 2", (count * D + end_offset - filter_offset) % I == 0)
            out.write(res)
            trace.append(("one", count))
            used = quot_up(count * D - filter_offset, I)
            buf_in = buf_in[used * w:]
            filter_offset = end_offset
            if length(buf_in) * I >= L - end_offset:
                continue
            try:
                nxt = np.asarray(next(it), np.float32)
            except StopIteration:
                break
            if length(buf_in) == 0:
                buf_in = nxt
            else:
                buf_last, buf_next, state = buf_in, nxt, "cross"
        else:
            _assert("resample 3", length(buf_last) * I < L - filter_offset)
            computable = quot_up(length(buf_last) * I + filter_offset, D)
            count = min(computable, out.space())
            _assert("resample 4", count != 0)
            res, dat, end_offset = model.cross(dat, count, buf_last, buf_next)
            _assert("resample 5", (count * D + end_offset - filter_offset) % I == 0)
            out.write(res)
            trace.append(("cross", count))
            used = quot_up(count * D - filter_offset, I)
            filter_offset = end_offset
            if used >= length(buf_last):
                buf_in = buf_next[(used - length(buf_last)) * w:]
                state = "simple"
            else:
                buf_last = buf_last[used * w:]
    return out.yielded, trace


def fm_demod_pipe(oracle, blocks):
    """fmDemod, Demod.hs:40-46: one output vector per input vector, carrying the last sample."""
    last = (0.0, 0.0)
    outs = []
    for b in blocks:
        b = np.asarray(b, np.float32)
        outs.append(oracle.fm_demod(b, last))
        last = (float(b[-2]), float(b[-1]))
    return outs


def fm_receiver(oracle, u8_blocks, decim_taps, factor, resamp_taps, I, D, audio_half, gain=None,
                block=8192, order=ORDER_AVX):
    """examples/fm/fm.hs:34-41 end to end on a finite list of u8 IQ blocks."""
    iq = [oracle.convert_u8(b) for b in u8_blocks]                                   # P.map convert
    deci = FilterModel(oracle, decim_taps, order, complex_=True, factor=factor)
    d_blocks, _ = fir_decimator_pipe(deci, iq, block)                                # firDecimator deci samples
    y_blocks = fm_demod_pipe(oracle, d_blocks)                                       # fmDemod
    resp = ResamplerModel(oracle, I, D, resamp_taps, order)
    z_blocks, _ = fir_resampler_pipe(resp, y_blocks, block)                          # firResampler resp samples
    filt = FilterModel(oracle, audio_half, order, sym=True)
    a_blocks, _ = fir_filter_pipe(filt, z_blocks, block)                             # firFilter filt samples
    if gain is not None:
        a_blocks = [oracle.scale(gain, a) for a in a_blocks]                         # P.map (VG.map (* 0.2))
    return a_blocks
