// pipes.cpp -- layer (3): the reference's Pipes operators on HOST blocks.
//
//   firFilter    Filter.hs:532-569      firDecimator  Filter.hs:574-611
//   firResampler Filter.hs:679-727      fmDemod       Demod.hs:40-46
//
// The reference's state machines alternate `simple` (outputs whose window fits in
// the current input buffer: the C SIMD kernel) and `crossover` (outputs straddling
// two buffers: the sequential Haskell kernel), re-blocking the outputs into vectors
// of exactly blockSizeOut (advanceOutBuf, Filter.hs:516-523: a block is yielded only
// when exactly full).  Here each pushed block triggers at most two launches on the
// pipe's stream -- the Cross outputs straddling the boundary with the previous
// block, then the One outputs inside the new block -- over a device buffer that
// holds [carried tail | new block].  Block sizes may vary from push to push.
//
// The pinned staging buffer of a slot holds [carried tail | staged blocks] contiguously; the tail (the last < Lp/I + D/I
// elements earlier pushes delivered and pending outputs still need) comes from a small host-side history, so the device
// never shuffles it.  Large submissions are double-buffered over three HIP streams: push(i) enqueues H2D(i) on the upload
// stream, kernels(i) on the compute stream (after H2D(i)'s event) and D2H(i) on the download stream (after the kernels'
// event), and only then harvests slot i-1 -- so the upload of block i and the download of block i-1 overlap the compute
// between them.  Small submissions (<= kDirectBytes of input) skip both copies: the kernels read the pinned staging buffer
// and write the pinned result buffer directly over PCIe, which leaves one kernel launch (seams decided inside it) and one event per push
// -- the reference's own block sizes (8192 .. 65536 elements) are launch-bound, not bandwidth-bound.  Results lag by one
// push; sdrhip_pipe_flush() drains the in-flight slot.
#include <stdlib.h>
#include <string.h>

#include <deque>

#include "descriptors.hpp"

using namespace sdrhip;

enum PipeKind { PK_FILTER, PK_DECIMATOR, PK_RESAMPLER, PK_DEMOD, PK_DCBLOCK };

struct sdrhip_pipe {
    PipeKind kind;
    const FirDesc* fir = nullptr;
    const ResampDesc* rs = nullptr;
    int block_out = 0;
    bool cplx_in = false, cplx_out = false;
    int I = 1, D = 1, Lp = 1;

    // Slots = submissions in flight (round 3: four; SDRHIP_STREAM_SLOTS=2..4).  An in-place push of one host block is one
    // small kernel, ~20 us of latency end to end over PCIe, and the host is done submitting it in ~6: the slots keep the GPU
    // fed.  Results lag nslots - 1 pushes (sdrhip_pipe_flush drains).
    static constexpr int kMaxSlots = 4;
    int nslots = kMaxSlots;
    hipStream_t stream = nullptr;   // compute (copy mode, map pipes)
    hipStream_t cstream[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};   // compute of the in-place pushes of slot si (cstream[0] == stream)
    hipStream_t up = nullptr;       // H2D
    hipStream_t down = nullptr;     // D2H
    // device input of the slots (copy mode), each holding [tail | blocks]
    DevBuf din[kMaxSlots];
    int cur_slot() const { return (int)(pushes % nslots); }
    // FIR-like pipes: room (elements) in front of the staged elements for the carried tail, and the host-side history
    // it is copied from: the stream's last hist_n elements
    int64_t head_cap = 0;
    std::vector<float> hist;
    int64_t hist_n = 0;
    bool direct_ok = getenv("SDRHIP_NO_DIRECT_STREAM") == nullptr;
    static constexpr size_t kDirectBytes = 512 << 10;     // [tail | staged] up to this size is read in place over PCIe
    static constexpr size_t kAdaptiveBytes = 4 << 20;     // adaptive submission stages at most this much
    int64_t E_prev = 0;     // global end of the previous block
    int64_t m_done = 0;     // outputs computed so far
    float last_re = 0.0f, last_im = 0.0f;  // fmDemod carry (Demod.hs:41,46)
    // coalescing of equal-sized pushes (FIR-like stages): samples staged in the current slot, not yet submitted
    int staged = 0;
    int coalesce = 0;          // blocks per submission (0/1: every push)
    // > 1: submit when the next slot is free, else keep staging up to this many blocks.  On by default (SDRHIP_STREAM_ADAPTIVE=0
    // or sdrhip_pipe_set_adaptive(p, 0) switch it off): a source that is slower than the GPU never notices, a faster one
    // gets the throughput of coalesced pushes at the reference's own block size
    int adaptive = getenv("SDRHIP_STREAM_ADAPTIVE") ? atoi(getenv("SDRHIP_STREAM_ADAPTIVE")) : kAdaptiveBlocks;
    static constexpr int kAdaptiveBlocks = 32;
    // blocks per submission in force for blocks of `uni` elements: the adaptive cap stays inside what is read in place
    // an explicit sdrhip_pipe_set_coalesce(p, N > 1) takes precedence: exactly N blocks per submission, as documented
    bool adaptive_on() const { return adaptive > 1 && coalesce <= 1; }
    int coalesce_eff(int uni) const
    {
        if (adaptive_on() && uni > 0) {
            const int64_t esz = (int64_t)(cplx_in ? 2 : 1) * 4;
            // (measured, 8192-sample cfloat blocks into firDecimator: batches of up to 0.5 / 1 / 4 / 16 MiB -> 2.7 / 2.3-3.0 /
            // 3.0-4.9 / 3.7-4.8 G elements/s; batches past kDirectBytes go through the copy engines)
            static const int64_t bytes = getenv("SDRHIP_ADAPTIVE_BYTES") ? atoll(getenv("SDRHIP_ADAPTIVE_BYTES")) : (int64_t)kAdaptiveBytes;
            const int64_t fit = bytes / ((int64_t)uni * esz);
            const int64_t b = adaptive < fit ? adaptive : fit;
            if (b >= 2) return (int)b;
        }
        return coalesce;
    }
    int uniform_n = 0;         // size of the first block; all_uniform: every block so far had it
    bool all_uniform = true;
    int lent = 0;              // elements behind `staged` the caller may have filled through sdrhip_pipe_input_buffer

    struct Slot {
        PinBuf hin, hout;
        DevBuf dout;
        hipEvent_t ev = nullptr;      // D2H complete
        hipEvent_t ev_up = nullptr;   // H2D complete
        hipEvent_t ev_k = nullptr;    // kernels complete
        int64_t n_out = 0;   // elements produced by the in-flight work
        bool busy = false;
        bool direct = false; // the last submission ran in place: `ev` also releases the staging buffer
    } slot[kMaxSlots];
    int64_t pushes = 0;
    // is slot si's last submission still running on the GPU?
    bool in_flight(int si) const { return slot[si].busy && hipEventQuery(slot[si].ev) == hipErrorNotReady; }

    // produced output floats not yet popped: contiguous storage + read cursor (memcpy in/out)
    std::vector<float> fifo;
    size_t fifo_head = 0;
    size_t fifo_size() const { return fifo.size() - fifo_head; }
    std::deque<int> demod_blocks;      // fmDemod / dcBlockingFilter: output block lengths (one per input block)
    DevBuf dc_state, dc_ws;            // dcBlockingFilter: {lastSample, lastOutput} carried on the device
    bool is_map() const { return kind == PK_DEMOD || kind == PK_DCBLOCK; }

    float* staged_base(Slot& sl) const { return (float*)sl.hin.p + (size_t)head_cap * (cplx_in ? 2 : 1); }   // staged element 0
    int esz_in() const { return cplx_in ? 2 : 1; }
    int esz_out() const { return cplx_out ? 2 : 1; }
    int64_t in_offset(int64_t m) const { return ceil_div64(m * (int64_t)D, I); }

    ~sdrhip_pipe()
    {
        for (hipStream_t st : {up, stream, cstream[1], cstream[2], cstream[3], down})
            if (st) (void)hipStreamSynchronize(st);
        for (auto& s : slot)
            for (hipEvent_t e : {s.ev, s.ev_up, s.ev_k})
                if (e) (void)hipEventDestroy(e);
        for (hipStream_t st : {up, stream, cstream[1], cstream[2], cstream[3], down})
            if (st) (void)hipStreamDestroy(st);
    }
};

static int pipe_new(sdrhip_pipe** out, PipeKind kind)
{
    sdrhip_pipe* p = new sdrhip_pipe();
    p->kind = kind;
    if (const char* env = getenv("SDRHIP_STREAM_SLOTS"))
        if (atoi(env) >= 2 && atoi(env) <= sdrhip_pipe::kMaxSlots) p->nslots = atoi(env);
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    p->cstream[0] = p->stream;
    for (int i = 1; i < p->nslots; i++)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->cstream[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking);
    for (auto& sl : p->slot)
        for (hipEvent_t* ev : {&sl.ev, &sl.ev_up, &sl.ev_k})
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("pipe: stream/event creation failed: %s", hipGetErrorString(e));
        delete p;
        return SDRHIP_ERR_HIP;
    }
    *out = p;
    return SDRHIP_OK;
}

static int harvest(sdrhip_pipe* p, int si)
{
    sdrhip_pipe::Slot& s = p->slot[si];
    if (!s.busy) return SDRHIP_OK;
    SDRHIP_CHECK_HIP(hipEventSynchronize(s.ev));
    const float* h = (const float*)s.hout.p;
    if (p->fifo_head > 0 && p->fifo_head == p->fifo.size()) {
        p->fifo.clear();
        p->fifo_head = 0;
    } else if (p->fifo_head > (1u << 20) && p->fifo_head * 2 > p->fifo.size()) {
        p->fifo.erase(p->fifo.begin(), p->fifo.begin() + p->fifo_head);   // compact occasionally
        p->fifo_head = 0;
    }
    const size_t old = p->fifo.size();
    p->fifo.resize(old + (size_t)s.n_out);
    memcpy(p->fifo.data() + old, h, (size_t)s.n_out * sizeof(float));
    s.busy = false;
    return SDRHIP_OK;
}

// harvest, oldest first, every in-flight submission the GPU has finished (never waits)
static int harvest_done(sdrhip_pipe* p)
{
    for (int64_t k = p->pushes - (p->nslots - 1); k < p->pushes; k++) {
        if (k < 0) continue;
        const int si = (int)(k % p->nslots);
        if (!p->slot[si].busy) continue;
        if (hipEventQuery(p->slot[si].ev) != hipSuccess) break;      // still running (an error surfaces in the blocking harvest)
        int rc = harvest(p, si);
        if (rc != SDRHIP_OK) return rc;
    }
    return SDRHIP_OK;
}

static int ready_blocks(const sdrhip_pipe* p)
{
    if (p->is_map()) {
        // complete blocks = those whose floats have all been harvested
        size_t have = p->fifo_size(), n = 0;
        for (int len : p->demod_blocks) {
            if (have < (size_t)len) break;
            have -= len;
            n++;
        }
        return (int)n;
    }
    return (int)(p->fifo_size() / ((size_t)p->block_out * p->esz_out()));
}

// Submit the n elements staged in the current slot's pinned buffer.  uniform_seam > 0: they are whole blocks of that size
// and so was everything before them, so the seams of the reference's input buffers are the multiples of uniform_seam and
// ONE stream-API run covers the batch (interior seams included).  uniform_seam == 0: a single block of arbitrary size: the
// outputs straddling the boundary with the previous block first (all Cross), then the ones inside the new block (all One).
static int fir_submit(sdrhip_pipe* p, int n, int64_t uniform_seam)
{
    const int si = p->cur_slot();
    sdrhip_pipe::Slot& sl = p->slot[si];
    int rc;
    const size_t ein = (size_t)p->esz_in() * 4, eout = (size_t)p->esz_out() * 4;
    const int64_t E_prev = p->E_prev, E = E_prev + n;
    // outputs computable once these samples are in: window end <= E*I
    int64_t m_end = (E * p->I >= p->Lp) ? (E * p->I - p->Lp) / p->D + 1 : 0;
    // first output starting at/after the previous boundary: everything before it
    // that is not yet done straddles that boundary (Cross)
    int64_t m_split = ceil_div64(E_prev * p->I, p->D);
    if (m_split < p->m_done) m_split = p->m_done;
    // ... unless the Pipe does not cross over at this boundary at all: the first pending output already has its first
    // input in the new block (`VG.length bufIn' == 0 -> simple next`, Filter.hs:707-709; resamplers only)
    if (E_prev > 0 && !seam_has_crossover(E_prev * p->I, p->I, p->D, p->Lp)) m_split = p->m_done;
    // ... and the last of them, when its first input is already in the new block, only if the output block had room for
    // it (kernels.hpp: late_output_is_one)
    if (m_split > p->m_done && late_output_is_one(m_split - 1, E_prev * p->I, p->I, p->D, p->block_out)) m_split--;
    // staging buffer = [carried tail | staged elements]: the tail starts at the first input any pending output needs,
    // rounded down to a multiple of 4 elements (16-byte aligned device reads wherever the block sizes allow)
    int64_t keep_from = p->in_offset(p->m_done);
    if (keep_from > E_prev) keep_from = E_prev;
    keep_from -= keep_from & 3;
    if (E_prev - keep_from > p->hist_n) {
        // the history always covers the carried tail (pipe_init_history sizes it for that); a state that says otherwise came
        // from a corrupt or hand-built checkpoint, and running on would index in front of the staging buffer
        set_error("pipe: the history holds %lld elements but pending output %lld needs %lld", (long long)p->hist_n,
                  (long long)p->m_done, (long long)(E_prev - keep_from));
        return SDRHIP_ERR_STATE;
    }
    const int64_t tail = E_prev - keep_from;
    if (tail > p->head_cap) {
        set_error("pipe: carried tail of %lld elements exceeds the head room (%lld)", (long long)tail, (long long)p->head_cap);
        return SDRHIP_ERR_STATE;
    }
    float* first = p->staged_base(sl) - tail * p->esz_in();
    if (tail > 0) memcpy(first, p->hist.data() + (size_t)(p->hist_n - tail) * p->esz_in(), (size_t)tail * ein);
    {
        const int64_t have = tail + n;
        const int64_t keep = have < p->head_cap ? have : p->head_cap;
        memmove(p->hist.data(), first + (size_t)(have - keep) * p->esz_in(), (size_t)keep * ein);
        p->hist_n = keep;
    }
    const bool direct = p->direct_ok && (size_t)(tail + n) * ein <= sdrhip_pipe::kDirectBytes;
    // in-place pushes go to their slot's own compute stream: nothing push i+1 computes depends on what push i left on the
    // device (the carried tail comes from the host-side history), so consecutive pushes overlap on the GPU
    hipStream_t cs = direct ? p->cstream[si] : p->stream;
    const float* din = nullptr;
    if (direct) {
        din = (const float*)sl.hin.dev_ptr(first);
    } else {
        DevBuf& dbuf = p->din[si];
        if ((rc = dbuf.ensure((size_t)(tail + n) * ein + 64)) != SDRHIP_OK) return rc;
        // slot si's device buffer was last read by the kernels of submission i-2, harvested before the slot was reopened
        SDRHIP_CHECK_HIP(hipMemcpyAsync(dbuf.p, first, (size_t)(tail + n) * ein, hipMemcpyHostToDevice, p->up));
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_up, p->up));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent(cs, sl.ev_up, 0));
        din = (const float*)dbuf.p;
    }
    const int64_t in_base = keep_from;

    const int64_t n_out = m_end - p->m_done;
    sl.n_out = 0;
    if (n_out > 0) {
        if ((rc = sl.hout.ensure((size_t)n_out * eout)) != SDRHIP_OK) return rc;
        float* dout = nullptr;
        if (direct) {
            dout = (float*)sl.hout.dev;
        } else {
            if ((rc = sl.dout.ensure((size_t)n_out * eout)) != SDRHIP_OK) return rc;
            dout = (float*)sl.dout.p;
        }
        if (uniform_seam > 0) {
            if (p->kind == PK_RESAMPLER)
                rc = resamp_run(p->rs, cs, din, in_base, dout, p->m_done, m_end, uniform_seam, p->block_out);
            else
                rc = fir_run(p->fir, cs, din, false, in_base, dout, p->m_done, m_end, uniform_seam);
            if (rc != SDRHIP_OK) return rc;
        } else {
            const int64_t ncross = m_split - p->m_done;
            if (p->kind == PK_RESAMPLER) {
                if (ncross > 0 && (rc = resamp_run(p->rs, cs, din, in_base, dout, p->m_done, m_split, -1)) != SDRHIP_OK) return rc;
                if ((rc = resamp_run(p->rs, cs, din, in_base, dout + ncross * p->esz_out(), m_split, m_end, 0)) != SDRHIP_OK) return rc;
            } else {
                if (ncross > 0 && (rc = fir_run(p->fir, cs, din, false, in_base, dout, p->m_done, m_split, -1)) != SDRHIP_OK) return rc;
                if ((rc = fir_run(p->fir, cs, din, false, in_base, dout + ncross * p->esz_out(), m_split, m_end, 0)) != SDRHIP_OK) return rc;
            }
        }
        if (!direct) {
            SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_k, cs));
            SDRHIP_CHECK_HIP(hipStreamWaitEvent(p->down, sl.ev_k, 0));
            SDRHIP_CHECK_HIP(hipMemcpyAsync(sl.hout.p, sl.dout.p, (size_t)n_out * eout, hipMemcpyDeviceToHost, p->down));
            SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, p->down));
        }
        sl.n_out = n_out * p->esz_out();
        sl.busy = true;
    }
    // direct mode, ONE event per push: the results are in pinned memory and the staging buffer is free again when the
    // kernels are done
    if (direct) SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, cs));
    sl.direct = direct;
    p->m_done = m_end;
    p->E_prev = E;
    p->pushes++;
    p->staged = 0;
    return harvest(p, p->cur_slot());     // the oldest submission: its slot is the next to be filled
}

// make the current slot's pinned staging buffer writable and big enough
static int fir_open_slot(sdrhip_pipe* p, size_t elems)
{
    sdrhip_pipe::Slot& sl = p->slot[p->cur_slot()];
    int rc;
    if (p->staged == 0) {
        if ((rc = harvest(p, p->cur_slot())) != SDRHIP_OK) return rc;
        SDRHIP_CHECK_HIP(hipEventSynchronize(sl.direct ? sl.ev : sl.ev_up));   // the slot's previous upload / in-place read has left the buffer
    }
    const size_t head_bytes = (size_t)p->head_cap * p->esz_in() * 4;
    if (sl.hin.cap < head_bytes + elems * p->esz_in() * 4) {
        // growing must keep what is already staged -- and what the caller wrote in place behind it (a block handed out by
        // sdrhip_pipe_input_buffer and not pushed yet); the staged elements live behind the head room
        PinBuf bigger;
        if ((rc = bigger.ensure(head_bytes + elems * p->esz_in() * 4)) != SDRHIP_OK) return rc;
        size_t keep = (size_t)(p->staged + p->lent) * p->esz_in() * 4;
        if (sl.hin.cap < head_bytes) keep = 0;
        else if (keep > sl.hin.cap - head_bytes) keep = sl.hin.cap - head_bytes;
        if (keep > 0) memcpy((char*)bigger.p + head_bytes, (char*)sl.hin.p + head_bytes, keep);
        std::swap(sl.hin.p, bigger.p);
        std::swap(sl.hin.cap, bigger.cap);
        std::swap(sl.hin.dev, bigger.dev);
    }
    return SDRHIP_OK;
}

// the reference's `assert "filter 1" / "decimate 1" / "resample 1"` for a block of n elements arriving at E_at: after the
// crossover the rest of the new buffer must still hold one whole filter
static int fir_check_block(const sdrhip_pipe* p, int64_t E_at, int64_t m_pending, int n)
{
    const int64_t E = E_at + n;
    const int64_t m_end = (E * p->I >= p->Lp) ? (E * p->I - p->Lp) / p->D + 1 : 0;
    int64_t m_split = ceil_div64(E_at * p->I, p->D);
    if (m_split < m_pending) m_split = m_pending;
    if (m_end <= m_split) {
        set_error("pipe: input block of %d elements is shorter than the filter (numCoeffs %d): the reference asserts "
                  "(Filter.hs:544,586,691)", n, p->Lp);
        return SDRHIP_ERR_ARG;
    }
    return SDRHIP_OK;
}

// head room and history of a FIR-like pipe: the carried tail is E_prev - in_offset(m_done) < (Lp + D) / I + 1 elements,
// plus up to 3 of alignment slack
static void pipe_init_history(sdrhip_pipe* p)
{
    const int64_t max_tail = ((int64_t)p->Lp + p->D) / p->I + 2;
    p->head_cap = (max_tail + 3 + 3) / 4 * 4 + 4;
    p->hist.assign((size_t)p->head_cap * p->esz_in(), 0.0f);
    p->hist_n = 0;
}

static int fir_like_push(sdrhip_pipe* p, const float* block, int n)
{
    int rc;
    const size_t ein = (size_t)p->esz_in() * 4;
    // the size of the first ACCEPTED block is the uniform size: a block the checks below refuse must not latch it
    const int uni = p->uniform_n == 0 ? n : p->uniform_n;
    const bool all_uniform = p->all_uniform && n == uni;
    const int ce = p->coalesce_eff(uni);
    const bool coalescing = ce > 1 && all_uniform;
    if ((int64_t)ce * uni > (int64_t)1 << 30) {
        set_error("pipe: %d coalesced blocks of %d elements exceed the staging limit", ce, uni);
        return SDRHIP_ERR_ARG;
    }
    // zero-copy push: `block` is the staging buffer's own write position (sdrhip_pipe_input_buffer); noted before the
    // buffer can be re-allocated below (growth keeps the lent region, so the data is then already in place)
    const bool in_place = p->slot[p->cur_slot()].hin.p != nullptr &&
                          block == p->staged_base(p->slot[p->cur_slot()]) + (size_t)p->staged * p->esz_in();
    const uint64_t pushes_at_entry = (uint64_t)p->pushes;
    if (!coalescing && p->staged > 0) {
        // a block of another size ends the uniform run: what is staged goes out as one uniform batch first
        if ((rc = fir_submit(p, p->staged, p->uniform_n)) != SDRHIP_OK) return rc;
    }
    // that submission moved on to the other slot: a block the caller wrote into the OLD slot's lent region (still intact:
    // the submitted run ends where the lent region starts) is not in place any more and must be copied like any other
    const bool still_in_place = in_place && (uint64_t)p->pushes == pushes_at_entry;
    // the block must be acceptable to the reference's Pipe where it arrives
    const int64_t m_pending = p->staged > 0 ? ((p->E_prev + p->staged) * p->I >= p->Lp ? ((p->E_prev + p->staged) * p->I - p->Lp) / p->D + 1 : 0)
                                             : p->m_done;
    if ((rc = fir_check_block(p, p->E_prev + p->staged, m_pending, n)) != SDRHIP_OK) return rc;
    p->uniform_n = uni;
    p->all_uniform = all_uniform;
    const int64_t cap = coalescing ? (int64_t)ce * p->uniform_n : n;
    if ((rc = fir_open_slot(p, (size_t)(cap > p->staged + n ? cap : p->staged + n))) != SDRHIP_OK) return rc;
    float* dst = p->staged_base(p->slot[p->cur_slot()]) + (size_t)p->staged * p->esz_in();
    if (!still_in_place) memcpy(dst, block, (size_t)n * ein);   // else: the caller filled the staging buffer in place
    p->lent = 0;
    p->staged += n;
    const int64_t pushes_before = p->pushes;
    if (!coalescing) {
        // equal-sized blocks from the start: the seams are the multiples of that size and one run covers Cross and One
        // outputs alike; otherwise (ragged blocks) the two-part submission
        if ((rc = fir_submit(p, p->staged, p->all_uniform ? p->uniform_n : 0)) != SDRHIP_OK) return rc;
    } else if (p->staged >= (int64_t)ce * p->uniform_n ||
               (p->adaptive_on() && !p->in_flight((p->cur_slot() + 1) % p->nslots))) {
        // (adaptive: a GPU that keeps up gets every push at once; one still busy with the slot this submission would move on
        // to lets the blocks pile up in the staging buffer and takes them as one launch when it frees up)
        if ((rc = fir_submit(p, p->staged, p->uniform_n)) != SDRHIP_OK) return rc;
    }
    // a push that went out also collects whatever the GPU has finished meanwhile (a source slower than the GPU gets the
    // results of push i at push i + 1 instead of i + nslots - 1); staged pushes skip the query
    if (p->pushes != pushes_before && (rc = harvest_done(p)) != SDRHIP_OK) return rc;
    return ready_blocks(p);
}

extern "C" {

int sdrhip_pipe_fir_filter(sdrhip_pipe** pp, const sdrhip_filter* f, int block_size_out)
{
    SDRHIP_REQUIRE(pp && f && block_size_out > 0, "sdrhip_pipe_fir_filter");
    int rc = pipe_new(pp, PK_FILTER);
    if (rc != SDRHIP_OK) return rc;
    sdrhip_pipe* p = *pp;
    p->fir = f;
    p->block_out = block_size_out;
    p->cplx_in = p->cplx_out = f->cplx;
    p->I = 1;
    p->D = 1;
    p->Lp = f->Lp;
    pipe_init_history(p);
    return SDRHIP_OK;
}

int sdrhip_pipe_fir_decimator(sdrhip_pipe** pp, const sdrhip_decimator* d, int block_size_out)
{
    SDRHIP_REQUIRE(pp && d && block_size_out > 0, "sdrhip_pipe_fir_decimator");
    int rc = pipe_new(pp, PK_DECIMATOR);
    if (rc != SDRHIP_OK) return rc;
    sdrhip_pipe* p = *pp;
    p->fir = d;
    p->block_out = block_size_out;
    p->cplx_in = p->cplx_out = d->cplx;
    p->I = 1;
    p->D = d->factor;
    p->Lp = d->Lp;
    pipe_init_history(p);
    return SDRHIP_OK;
}

int sdrhip_pipe_fir_resampler(sdrhip_pipe** pp, const sdrhip_resampler* r, int block_size_out)
{
    SDRHIP_REQUIRE(pp && r && block_size_out > 0, "sdrhip_pipe_fir_resampler");
    SDRHIP_REQUIRE(r->Lp >= r->D, "sdrhip_pipe_fir_resampler: padded filter shorter than the decimation step: the reference Pipe "
                                  "mis-steps at buffer boundaries (Filter.hs:702-709)");
    int rc = pipe_new(pp, PK_RESAMPLER);
    if (rc != SDRHIP_OK) return rc;
    sdrhip_pipe* p = *pp;
    p->rs = r;
    p->block_out = block_size_out;
    p->cplx_in = p->cplx_out = r->cplx;
    p->I = r->I;
    p->D = r->D;
    p->Lp = r->Lp;
    pipe_init_history(p);
    return SDRHIP_OK;
}

int sdrhip_pipe_fm_demod(sdrhip_pipe** pp)
{
    SDRHIP_REQUIRE(pp != nullptr, "sdrhip_pipe_fm_demod");
    int rc = pipe_new(pp, PK_DEMOD);
    if (rc != SDRHIP_OK) return rc;
    sdrhip_pipe* p = *pp;
    p->cplx_in = true;
    p->cplx_out = false;
    return SDRHIP_OK;
}

int sdrhip_pipe_dc_blocker(sdrhip_pipe** pp)
{
    SDRHIP_REQUIRE(pp != nullptr, "sdrhip_pipe_dc_blocker");
    int rc = pipe_new(pp, PK_DCBLOCK);
    if (rc != SDRHIP_OK) return rc;
    sdrhip_pipe* p = *pp;
    p->cplx_in = false;
    p->cplx_out = false;
    if ((rc = p->dc_state.ensure(16)) != SDRHIP_OK) { delete p; *pp = nullptr; return rc; }
    hipError_t e = hipMemsetAsync(p->dc_state.p, 0, 16, p->stream);   // func 0 0, Filter.hs:732
    if (e != hipSuccess) { set_error("sdrhip_pipe_dc_blocker: %s", hipGetErrorString(e)); delete p; *pp = nullptr; return SDRHIP_ERR_HIP; }
    return SDRHIP_OK;
}

// ---- map stages (fmDemod, dcBlockingFilter): one output vector per input vector --------------------------------------
// Blocks are staged one behind the other in the slot's pinned buffer and go out as ONE run over all of them -- the carry of a
// block is its predecessor's last sample (fmDemod, Demod.hs:41,46) / the filter's running pair kept on the device
// (dcBlockingFilter, Filter.hs:730-739), which is exactly what a run over the concatenation computes; the block lengths are
// remembered for the pops.  Submission is adaptive as for the FIR-like stages.  Runs of up to kDirectBytes read and write
// pinned memory in place; larger ones go through the copy engines.
static int map_submit(sdrhip_pipe* p)
{
    const int n = p->staged;
    if (n == 0) return SDRHIP_OK;
    const int si = p->cur_slot();
    sdrhip_pipe::Slot& sl = p->slot[si];
    int rc;
    const size_t ein = (size_t)p->esz_in() * 4, eout = (size_t)p->esz_out() * 4;
    if ((rc = sl.hout.ensure((size_t)n * eout)) != SDRHIP_OK) return rc;
    // (dcBlockingFilter never in place: its lanes re-read their run-in and a short block is one lane's dependent loads)
    const bool direct = p->direct_ok && p->kind == PK_DEMOD && (size_t)n * ein <= sdrhip_pipe::kDirectBytes;
    const float* d_in;
    float* d_out;
    if (direct) {
        d_in = (const float*)sl.hin.dev;
        d_out = (float*)sl.hout.dev;
    } else {
        DevBuf& d = p->din[si];              // last read by the kernels of submission i - nslots, harvested when the slot was opened
        if ((rc = d.ensure((size_t)n * ein)) != SDRHIP_OK) return rc;
        if ((rc = sl.dout.ensure((size_t)n * eout)) != SDRHIP_OK) return rc;
        SDRHIP_CHECK_HIP(hipMemcpyAsync(d.p, sl.hin.p, (size_t)n * ein, hipMemcpyHostToDevice, p->up));
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_up, p->up));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent(p->stream, sl.ev_up, 0));
        d_in = (const float*)d.p;
        d_out = (float*)sl.dout.p;
    }
    if (p->kind == PK_DEMOD) {
        launch_fm_demod_fast(p->stream, d_in, d_out, n, false, p->last_re, p->last_im);
        const float* h = (const float*)sl.hin.p;
        p->last_re = h[2 * (size_t)(n - 1)];
        p->last_im = h[2 * (size_t)(n - 1) + 1];
    } else {
        if (dc_blocker_workspace_bytes(n) > p->dc_ws.cap) SDRHIP_CHECK_HIP(hipStreamSynchronize(p->stream));     // growing frees the old buffer
        if ((rc = p->dc_ws.ensure(dc_blocker_workspace_bytes(n))) != SDRHIP_OK) return rc;
        launch_dc_blocker(p->stream, n, 0.0f, 0.0f, d_in, d_out, (float*)p->dc_state.p, p->dc_ws.p, 0, (const float*)p->dc_state.p);
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
    if (direct) {
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, p->stream));           // results in pinned memory, staging buffer free again
    } else {
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_k, p->stream));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent(p->down, sl.ev_k, 0));
        SDRHIP_CHECK_HIP(hipMemcpyAsync(sl.hout.p, sl.dout.p, (size_t)n * eout, hipMemcpyDeviceToHost, p->down));
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, p->down));
    }
    sl.direct = direct;
    sl.n_out = (int64_t)n * p->esz_out();
    sl.busy = true;
    p->staged = 0;
    p->pushes++;
    return harvest(p, p->cur_slot());      // the oldest submission: its slot is the next to be filled
}

static int map_push(sdrhip_pipe* p, const float* block, int n)
{
    int rc;
    const size_t ein = (size_t)p->esz_in() * 4;
    static const int64_t abytes = getenv("SDRHIP_ADAPTIVE_BYTES") ? atoll(getenv("SDRHIP_ADAPTIVE_BYTES")) : (int64_t)sdrhip_pipe::kAdaptiveBytes;
    const int64_t cap_bytes = p->adaptive > 1 ? abytes : 0;                  // what may pile up before a push has to go out
    // room for this block behind what is staged: a pinned buffer does not grow with staged blocks in it
    if (p->staged > 0 && ((size_t)(p->staged + n) * ein > p->slot[p->cur_slot()].hin.cap || (int64_t)p->staged + n > (1 << 30)) &&
        (rc = map_submit(p)) != SDRHIP_OK) return rc;
    sdrhip_pipe::Slot& sl = p->slot[p->cur_slot()];
    if (p->staged == 0) {
        // the slot's previous submission: results harvested, staging buffer no longer read (in place: the same event)
        if ((rc = harvest(p, p->cur_slot())) != SDRHIP_OK) return rc;
        // room for what may pile up, sized by the blocks this pipe actually sees (not the whole cap for tiny blocks)
        const int64_t pile = p->adaptive > 1 ? (int64_t)p->adaptive * (int64_t)n * (int64_t)ein : 0;
        const size_t want_cap = (size_t)(pile < cap_bytes ? pile : cap_bytes);
        const size_t want = (size_t)n * ein > want_cap ? (size_t)n * ein : want_cap;
        if ((rc = sl.hin.ensure(want)) != SDRHIP_OK) return rc;
    }
    memcpy((char*)sl.hin.p + (size_t)p->staged * ein, block, (size_t)n * ein);
    p->staged += n;
    p->demod_blocks.push_back(n);
    const int64_t pushes_before = p->pushes;
    const bool room = (int64_t)(p->staged + n) * (int64_t)ein <= cap_bytes && (size_t)(p->staged + n) * ein <= sl.hin.cap;
    if (!room || !p->in_flight((p->cur_slot() + 1) % p->nslots)) {
        if ((rc = map_submit(p)) != SDRHIP_OK) return rc;
    }
    if (p->pushes != pushes_before && (rc = harvest_done(p)) != SDRHIP_OK) return rc;
    return ready_blocks(p);
}

int sdrhip_pipe_push(sdrhip_pipe* p, const float* block, int n)
{
    SDRHIP_REQUIRE(p != nullptr && block != nullptr && n > 0, "sdrhip_pipe_push");
    return p->is_map() ? map_push(p, block, n) : fir_like_push(p, block, n);
}

int sdrhip_pipe_set_coalesce(sdrhip_pipe* p, int blocks)
{
    SDRHIP_REQUIRE(p != nullptr && blocks >= 0, "sdrhip_pipe_set_coalesce");
    SDRHIP_REQUIRE(!p->is_map(), "sdrhip_pipe_set_coalesce: filter / decimator / resampler pipes only");
    SDRHIP_REQUIRE(p->staged == 0, "sdrhip_pipe_set_coalesce: blocks are staged (flush first)");
    SDRHIP_REQUIRE(blocks <= 1 || p->uniform_n == 0 || (int64_t)blocks * p->uniform_n <= (int64_t)1 << 30,
                   "sdrhip_pipe_set_coalesce: coalesced batch too large");
    p->coalesce = blocks;
    return SDRHIP_OK;
}

int sdrhip_pipe_set_adaptive(sdrhip_pipe* p, int max_blocks)
{
    SDRHIP_REQUIRE(p != nullptr && max_blocks >= 0 && max_blocks != 1, "sdrhip_pipe_set_adaptive: 0 (off) or at least two blocks");
    SDRHIP_REQUIRE(p->staged == 0, "sdrhip_pipe_set_adaptive: blocks are staged (flush first)");
    p->adaptive = max_blocks;
    return SDRHIP_OK;
}

float* sdrhip_pipe_input_buffer(sdrhip_pipe* p, int n)
{
    if (p == nullptr || n <= 0 || p->is_map()) { set_error("sdrhip_pipe_input_buffer: filter / decimator / resampler pipes, n > 0"); return nullptr; }
    const int ce = p->coalesce_eff(p->uniform_n ? p->uniform_n : n);
    const bool coalescing = ce > 1 && p->all_uniform && (p->uniform_n == 0 || p->uniform_n == n);
    if (!coalescing && p->staged > 0 && fir_submit(p, p->staged, p->uniform_n) != SDRHIP_OK) return nullptr;
    // a fresh pipe has no uniform size yet: the block about to be pushed defines it, so size the buffer for a whole
    // coalesced batch of such blocks now (growing it at the push would move the block the caller is about to fill)
    const int64_t cap = coalescing ? (int64_t)ce * (p->uniform_n ? p->uniform_n : n) : n;
    if (cap > (int64_t)1 << 30) { set_error("sdrhip_pipe_input_buffer: coalesced batch too large"); return nullptr; }
    if (fir_open_slot(p, (size_t)(cap > p->staged + n ? cap : p->staged + n)) != SDRHIP_OK) return nullptr;
    p->lent = n;
    return p->staged_base(p->slot[p->cur_slot()]) + (size_t)p->staged * p->esz_in();
}

int sdrhip_pipe_poll(sdrhip_pipe* p)
{
    SDRHIP_REQUIRE(p != nullptr, "sdrhip_pipe_poll");
    int rc = harvest_done(p);
    if (rc != SDRHIP_OK) return rc;
    return ready_blocks(p);
}

int sdrhip_pipe_flush(sdrhip_pipe* p)
{
    SDRHIP_REQUIRE(p != nullptr, "sdrhip_pipe_flush");
    int rc;
    if (p->staged > 0 && (rc = p->is_map() ? map_submit(p) : fir_submit(p, p->staged, p->uniform_n)) != SDRHIP_OK) return rc;
    // oldest first
    const int first = p->cur_slot();
    for (int k = 0; k < p->nslots; k++)
        if ((rc = harvest(p, (first + k) % p->nslots)) != SDRHIP_OK) return rc;
    return ready_blocks(p);
}

int sdrhip_pipe_pop(sdrhip_pipe* p, float* out, int capacity)
{
    SDRHIP_REQUIRE(p != nullptr && out != nullptr, "sdrhip_pipe_pop");
    if (ready_blocks(p) <= 0) return 0;
    int len = p->is_map() ? p->demod_blocks.front() : p->block_out;
    SDRHIP_REQUIRE(capacity >= len, "sdrhip_pipe_pop: capacity smaller than the block");
    size_t nf = (size_t)len * p->esz_out();
    memcpy(out, p->fifo.data() + p->fifo_head, nf * sizeof(float));
    p->fifo_head += nf;
    if (p->is_map()) p->demod_blocks.pop_front();
    return len;
}

// ---- checkpoint / resume (as sdrhip_fm_stream_save / _restore, chain.cpp) -------------------------------------------
// Between two pushes a Pipe's state is its position (elements consumed, outputs produced), the last head_cap input elements,
// the carried fmDemod sample / dcBlocker pair, and the output not yet popped: what the reference keeps in the Pipe's closure
// (Filter.hs:536-727: overlap remainder, resampler (group, offset); Demod.hs:41,46; Filter.hs:730-739).
namespace {
struct PipeStateHeader {
    uint32_t magic, version;
    int32_t kind, block_out, I, D, Lp, cplx_in, cplx_out, uniform_n, all_uniform, n_blocks;
    int64_t E_prev, m_done, head_cap, hist_n, pending;       // hist_n: elements; pending: floats in the fifo
    float last_re, last_im, dc[4];
};
constexpr uint32_t kPipeMagic = 0x50504453u;   // "SDPP"
}  // namespace

size_t sdrhip_pipe_state_bytes(sdrhip_pipe* p)
{
    if (p == nullptr) return 0;
    // Exact, not an estimate: the pipe is drained here exactly as sdrhip_pipe_save will drain it (what is staged goes out, both
    // slots are harvested into the fifo), so the size returned is what the save that follows needs -- whatever the ratio of
    // the stage and the size of the blocks in flight.  0 = the drain failed (sdrhip_last_error).
    if (sdrhip_pipe_flush(p) < 0) return 0;
    return sizeof(PipeStateHeader) + (size_t)p->hist_n * p->esz_in() * sizeof(float) + p->fifo_size() * sizeof(float) +
           p->demod_blocks.size() * sizeof(int32_t);
}

int sdrhip_pipe_save(sdrhip_pipe* p, void* buf, size_t capacity, size_t* used)
{
    SDRHIP_REQUIRE(p != nullptr && buf != nullptr && used != nullptr, "sdrhip_pipe_save");
    int rc = sdrhip_pipe_flush(p);
    if (rc < 0) return rc;
    PipeStateHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kPipeMagic;
    h.version = 1;
    h.kind = (int32_t)p->kind;
    h.block_out = p->block_out;
    h.I = p->I; h.D = p->D; h.Lp = p->Lp;
    h.cplx_in = p->cplx_in; h.cplx_out = p->cplx_out;
    h.uniform_n = p->uniform_n; h.all_uniform = p->all_uniform;
    h.n_blocks = (int32_t)p->demod_blocks.size();
    h.E_prev = p->E_prev; h.m_done = p->m_done; h.head_cap = p->head_cap; h.hist_n = p->hist_n;
    h.pending = (int64_t)p->fifo_size();
    h.last_re = p->last_re; h.last_im = p->last_im;
    if (p->kind == PK_DCBLOCK && p->dc_state.p) {
        SDRHIP_CHECK_HIP(hipStreamSynchronize(p->stream));
        SDRHIP_CHECK_HIP(hipMemcpy(h.dc, p->dc_state.p, 16, hipMemcpyDeviceToHost));
    }
    const size_t hist_bytes = (size_t)h.hist_n * p->esz_in() * sizeof(float);
    const size_t need = sizeof h + hist_bytes + (size_t)h.pending * sizeof(float) + (size_t)h.n_blocks * sizeof(int32_t);
    if (capacity < need) {
        set_error("sdrhip_pipe_save: %zu bytes needed, %zu given", need, capacity);
        return SDRHIP_ERR_ARG;
    }
    unsigned char* o = (unsigned char*)buf;
    memcpy(o, &h, sizeof h); o += sizeof h;
    if (hist_bytes) memcpy(o, p->hist.data(), hist_bytes);
    o += hist_bytes;
    if (h.pending) memcpy(o, p->fifo.data() + p->fifo_head, (size_t)h.pending * sizeof(float));
    o += (size_t)h.pending * sizeof(float);
    for (int len : p->demod_blocks) { const int32_t v = len; memcpy(o, &v, sizeof v); o += sizeof v; }
    *used = need;
    return SDRHIP_OK;
}

int sdrhip_pipe_restore(sdrhip_pipe* p, const void* buf, size_t bytes)
{
    SDRHIP_REQUIRE(p != nullptr && buf != nullptr && bytes >= sizeof(PipeStateHeader), "sdrhip_pipe_restore");
    SDRHIP_REQUIRE(p->pushes == 0 && p->E_prev == 0 && p->staged == 0 && p->fifo_size() == 0,
                   "sdrhip_pipe_restore: only into a pipe that has not been pushed to");
    PipeStateHeader h;
    memcpy(&h, buf, sizeof h);
    SDRHIP_REQUIRE(h.magic == kPipeMagic && h.version == 1, "sdrhip_pipe_restore: not a pipe state");
    SDRHIP_REQUIRE(h.kind == (int32_t)p->kind && h.block_out == p->block_out && h.I == p->I && h.D == p->D && h.Lp == p->Lp &&
                       h.cplx_in == (int32_t)p->cplx_in && h.cplx_out == (int32_t)p->cplx_out && h.head_cap == p->head_cap,
                   "sdrhip_pipe_restore: the state belongs to a pipe of another kind or geometry");
    SDRHIP_REQUIRE(h.hist_n >= 0 && h.hist_n <= h.head_cap && h.pending >= 0 && h.n_blocks >= 0 && h.E_prev >= h.hist_n && h.m_done >= 0,
                   "sdrhip_pipe_restore: inconsistent state");
    if (p->kind == PK_FILTER || p->kind == PK_DECIMATOR || p->kind == PK_RESAMPLER) {
        // the next pending output must start inside what the pipe has seen, every output computable from those elements must be
        // done at most once, and the history must reach back to its first input (3 elements of alignment slack, fir_submit):
        // otherwise the kernels would be sent in front of the staging buffer
        const int64_t first_in = p->in_offset(h.m_done);
        const int64_t m_max = (h.E_prev * p->I >= p->Lp) ? (h.E_prev * p->I - p->Lp) / p->D + 1 : 0;
        int64_t keep_from = first_in < h.E_prev ? first_in : h.E_prev;
        keep_from -= keep_from & 3;
        SDRHIP_REQUIRE(h.m_done <= m_max && first_in <= h.E_prev + (p->Lp + p->D) / p->I + 1 && h.hist_n >= h.E_prev - keep_from,
                       "sdrhip_pipe_restore: the position and the history of the state do not belong together");
    }
    const size_t hist_bytes = (size_t)h.hist_n * p->esz_in() * sizeof(float);
    SDRHIP_REQUIRE(bytes >= sizeof h + hist_bytes + (size_t)h.pending * sizeof(float) + (size_t)h.n_blocks * sizeof(int32_t),
                   "sdrhip_pipe_restore: truncated state");
    const unsigned char* in = (const unsigned char*)buf + sizeof h;
    if (p->hist.size() * sizeof(float) < hist_bytes) p->hist.resize(hist_bytes / sizeof(float));
    if (hist_bytes) memcpy(p->hist.data(), in, hist_bytes);
    in += hist_bytes;
    p->hist_n = h.hist_n;
    p->E_prev = h.E_prev;
    p->m_done = h.m_done;
    p->uniform_n = h.uniform_n;
    p->all_uniform = h.all_uniform != 0;
    p->last_re = h.last_re;
    p->last_im = h.last_im;
    p->fifo.assign((const float*)in, (const float*)in + h.pending);
    p->fifo_head = 0;
    in += (size_t)h.pending * sizeof(float);
    p->demod_blocks.clear();
    for (int i = 0; i < h.n_blocks; i++) { int32_t v; memcpy(&v, in, sizeof v); in += sizeof v; p->demod_blocks.push_back(v); }
    if (p->kind == PK_DCBLOCK && p->dc_state.p) {
        SDRHIP_CHECK_HIP(hipStreamSynchronize(p->stream));       // the create call's memset
        SDRHIP_CHECK_HIP(hipMemcpy(p->dc_state.p, h.dc, 16, hipMemcpyHostToDevice));
    }
    return ready_blocks(p);
}

void sdrhip_pipe_destroy(sdrhip_pipe* p) { delete p; }

}  // extern "C"
