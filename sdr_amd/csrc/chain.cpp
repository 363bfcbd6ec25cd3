// chain.cpp -- the FM receiver chain of examples/fm/fm.hs:34-41 as one device-resident
// object:  u8 IQ -> [convert fused] -> firDecimator -> fmDemod -> firResampler ->
// firFilter(sym) -> *gain.  Every stage runs with global stream indices, so a shard
// (plus a right halo) of the stream can be processed anywhere -- on any GPU --
// and yields exactly the bits the single-stream Pipes would have produced.
//
// Index spaces:  n input samples -> k decimator outputs (window [k*D1, k*D1+P1))
//   -> y[k] = phase(d[k] * conj d[k-1]) -> m resampler outputs (inputs from
//   inOff(m) = ceil(m*D2/I2)) -> q audio outputs (window [q, q+L3)).
// Seams: all four Pipes of fm.hs run with blockSizeOut = `block` and the source
// delivers `block`-sample buffers, so each stage's input blocks are `block` long.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <exception>
#include "descriptors.hpp"

using namespace sdrhip;

static const int kFusedTailAutoOutputs = 768;   // fused tail in auto mode: runs of at most this many audio outputs
static const int kStages = 6;  // decimate(+seam fix-up), fmDemod, resample, filter, fused tail (fmDemod+resample+filter+gain in one kernel), whole chain in one kernel
// the whole chain as ONE kernel (kernels_small.hip) in auto mode: runs of at most this many audio outputs (~7.3 M input samples);
// measured on MI355X (tools/shard_pass_probe.py): see DESIGN.md 5 "one-kernel chain"
static const int64_t kSmallChainAutoOutputs = 159 * 1728;   // round 5 (tools/launch_sweep.py): the crossover with the stage kernels sits at ~900
                                                            // source blocks (43 us either way); the old bound of 256 blocks left runs of 384 .. 768
                                                            // blocks on the stage kernels, 1.2 .. 1.6 times slower than this kernel

struct sdrhip_fm_chain {
    FirDesc decim;     // complex, factor D1
    ResampDesc resamp; // real I2/D2
    FirDesc audio;     // symmetric real
    float gain = 1.0f;
    int64_t block = 0;
    // The fused convert + decimate kernel (k_decimate_c4) covers the AVX order, decimation 4 / 8 / 16, up to 128 (4) or 256 (8, 16) taps.  Any other
    // first stage converts the u8 IQ to cfloat in the workspace first (convert.c as its own kernel, 10 B per sample) and
    // then runs the tiled cfloat decimator -- two passes, but not the one-thread-per-output u8 fallback.
    bool fused_first_stage() const
    {
        const int D = decim.factor;
        if (!((D == 4 || D == 8 || D == 16) && decim.Lp > D && decim.Lp % 4 == 0)) return false;
        if (decim.h_scaled.empty()) return false;      // a tap too small to pre-scale by 1/128 exactly (descriptors.hpp)
        if (decim.corder == CO_L4) return decim.Lp <= (D == 4 ? 128 : 256);
        return decim.corder == CO_L2 && decim.Lp <= 128;      // the SSE order's fused instantiations (kernels_fast_orders.hip)
    }

    // fmDemod -> resampler -> audio filter (* gain) as ONE kernel (kernels_tail.hip), y and z never leaving LDS.  Measured on
    // MI355X (2^29 samples per run): 0.43-0.49 ms against 0.38 + 0.02 ms for the three stage kernels and their seam fix-ups --
    // every stage is VALU-bound, fusion saves HBM traffic that was not the limit and pays 7 % recomputed overlap; but a run
    // that fills at most one tile (a push of one to six 8192-sample source blocks) costs ONE launch of a few microseconds
    // instead of eight.  mode 0 = never, 1 = always, 2 = auto (runs of at most one tile): sdrhip_fm_chain_set_fused_tail,
    // SDRHIP_FUSED_TAIL=0/1/2.  (Rounds 2-4 also measured sub-batch pipelining of one run over two streams and "fmDemod kernel + fused
    // resampler / filter": both slower, both gone; LABNOTES.)
    int fused_tail = getenv("SDRHIP_FUSED_TAIL") ? atoi(getenv("SDRHIP_FUSED_TAIL")) : 2;
    // fmDemod in the resampler's tile loader: ON by default since round 4 (with the packed-pair resampler the pass gains 1.0 %,
    // 1.005-1.011 against 1.017-1.019 ms, alternating in one process: the pair itself is slower, 0.269 against 0.159 + 0.089 ms, but
    // 0.54 GB less traffic per pass leaves the power-capped decimator 4 % more clock)
    bool fuse_demod = getenv("SDRHIP_FUSE_DEMOD") ? atoi(getenv("SDRHIP_FUSE_DEMOD")) != 0 : true;
    bool tail_shape_ok(int64_t n_out) const
    {
        // auto: runs of up to two source blocks (measured per push, in place: fused 27.3 / 29.1 / 36.6 us for 1 / 2 / 4 blocks,
        // the stage kernels on their one-launch routes 30.0 / 30.7 / 32.6 -- the single workgroup of a one-tile run is serial)
        if (fused_tail == 0 || (fused_tail == 2 && n_out > kFusedTailAutoOutputs)) return false;
        return !resamp.cplx && resamp.lanes == 8 && !audio.cplx && audio.sym && audio.lanes == 8 && audio.factor == 1;
    }
    // The whole chain as one kernel for launch-bound runs (kernels_small.hip): 0 = never, 1 = whenever the configuration is the
    // one it is written for, 2 = auto (runs of at most small_chain_max audio outputs): sdrhip_fm_chain_set_small_chain,
    // SDRHIP_SMALL_CHAIN=0/1/2, SDRHIP_SMALL_CHAIN_MAX=<outputs>, SDRHIP_SMALL_CHAIN_TILE=<audio outputs per workgroup, 0 = by size>.
    int small_chain = getenv("SDRHIP_SMALL_CHAIN") ? atoi(getenv("SDRHIP_SMALL_CHAIN")) : 2;
    int64_t small_chain_max = getenv("SDRHIP_SMALL_CHAIN_MAX") ? atoll(getenv("SDRHIP_SMALL_CHAIN_MAX")) : kSmallChainAutoOutputs;
    int small_chain_tile = getenv("SDRHIP_SMALL_CHAIN_TILE") ? atoi(getenv("SDRHIP_SMALL_CHAIN_TILE")) : 0;
    bool input_over_link = false;   // set by the host-block operator around its in-place pushes: the input is pinned HOST memory
    bool small_chain_ok(int64_t n_out) const
    {
        if (small_chain == 0 || (small_chain == 2 && n_out > small_chain_max)) return false;
        if (!(decim.factor == 8 && decim.Lp == 128 && decim.corder == CO_L4 && !decim.h_scaled.empty())) return false;
        return !resamp.cplx && resamp.lanes == 8 && !audio.cplx && audio.sym && audio.lanes == 8 && audio.factor == 1;
    }
    // Two runs in flight (sdrhip_fm_chain_set_overlap, round 4): consecutive runs alternate between two internal streams
    // ("lanes") and the two halves of the workspace, so the memory-heavy tail kernels of run k execute beside the power-bound
    // decimator of run k+1.  Consecutive runs of a stream are independent given their raw input (fm.hs:34-41: a block's
    // audio needs nothing of the previous block's results), which is what makes this legal.
    int overlap = 0;
    hipStream_t lane[2] = {nullptr, nullptr};
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    bool lane_busy[2] = {false, false};
    unsigned run_idx = 0;
    int ensure_lanes()
    {
        if (lane[0]) return SDRHIP_OK;
        for (int j = 0; j < 2; j++) {
            SDRHIP_CHECK_HIP(hipStreamCreateWithFlags(&lane[j], hipStreamNonBlocking));
            SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&ev_in[j], hipEventDisableTiming));
            SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&ev_out[j], hipEventDisableTiming));
        }
        return SDRHIP_OK;
    }

    // optional per-stage timing with HIP events on the stream each kernel is launched on
    bool timing = false;
    struct Span { int stage; hipEvent_t b, e; };
    std::vector<hipEvent_t> ev_pool;     // all timing events ever created
    size_t ev_used = 0;
    std::vector<Span> spans;             // recorded since the last read
    int runs = 0;
    int new_event(hipEvent_t* ev)
    {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            SDRHIP_CHECK_HIP(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
        *ev = ev_pool[ev_used++];
        return SDRHIP_OK;
    }
    ~sdrhip_fm_chain()
    {
        for (int j = 0; j < 2; j++) {
            if (lane[j]) (void)hipStreamSynchronize(lane[j]);
            if (ev_in[j]) (void)hipEventDestroy(ev_in[j]);
            if (ev_out[j]) (void)hipEventDestroy(ev_out[j]);
            if (lane[j]) (void)hipStreamDestroy(lane[j]);
        }
        for (auto ev : ev_pool) (void)hipEventDestroy(ev);
    }

    // reach of resampler output m in y: the One kernel walks nloop floats, the Cross
    // kernel at most ceil(ntaps/I) <= nloop
    int y_reach() const { return resamp.nloop; }

    // first input sample in the receptive field of audio output q
    int64_t start(int64_t q) const
    {
        int64_t k = resamp.in_offset(q);
        if (k > 0) k -= 1;  // fmDemod looks one decimator output back (Demod.hs:28)
        return k * decim.factor;
    }
    // one past the last input sample in the receptive field of audio output q
    int64_t end(int64_t q) const
    {
        int64_t m_last = q + audio.Lp - 1;
        int64_t k_last = resamp.in_offset(m_last) + y_reach() - 1;
        return k_last * decim.factor + decim.Lp;
    }
    // smallest q with start(q) >= s
    int64_t first_q_from(int64_t s) const
    {
        if (s <= 0) return 0;
        int64_t lo = 0, hi = (s / decim.factor + 2) * resamp.I / resamp.D + 4;
        while (start(hi) < s) hi *= 2;
        while (lo < hi) {
            int64_t mid = lo + (hi - lo) / 2;
            if (start(mid) >= s) hi = mid; else lo = mid + 1;
        }
        return lo;
    }
    // number of audio outputs that exist for a stream of total_in samples (by data
    // availability; the Pipes additionally withhold a trailing partial block)
    int64_t total_q(int64_t total_in) const
    {
        if (total_in < 0) return INT64_MAX / 4;
        if (total_in < decim.Lp) return 0;
        int64_t K = (total_in - decim.Lp) / decim.factor + 1;           // decimator outputs == demod outputs
        if (K * resamp.I < resamp.Lp) return 0;
        int64_t M = (K * resamp.I - resamp.Lp) / resamp.D + 1;          // resampler outputs
        if (M < audio.Lp) return 0;
        return M - audio.Lp + 1;
    }
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" {

int sdrhip_fm_chain_create(sdrhip_fm_chain** c, int order, int decim_factor, const float* decim_taps, int n_decim_taps,
                           int interpolation, int decimation, const float* resamp_taps, int n_resamp_taps,
                           const float* audio_half_taps, int n_audio_half, float gain, int64_t block)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_create");
    *c = nullptr;
    SDRHIP_REQUIRE(block >= 0, "sdrhip_fm_chain_create");
    sdrhip_fm_chain* ch = new sdrhip_fm_chain();
    int rc = fir_create(&ch->decim, order, true, decim_factor, decim_taps, n_decim_taps);
    if (rc == SDRHIP_OK) rc = resamp_create(&ch->resamp, order, false, interpolation, decimation, resamp_taps, n_resamp_taps);
    if (rc == SDRHIP_OK) rc = fir_sym_create(&ch->audio, order, 1, audio_half_taps, n_audio_half);
    if (rc == SDRHIP_OK && block != 0 &&
        !(block >= ch->decim.Lp && block * ch->resamp.I >= ch->resamp.Lp && block >= ch->audio.Lp)) {
        set_error("sdrhip_fm_chain_create: block %lld shorter than a stage's filter (Filter.hs:544,586,691)", (long long)block);
        rc = SDRHIP_ERR_ARG;
    }
    if (rc != SDRHIP_OK) { delete ch; return rc; }
    ch->gain = gain;
    ch->block = block;
    *c = ch;
    return SDRHIP_OK;
}

void sdrhip_fm_chain_destroy(sdrhip_fm_chain* c) { delete c; }

int sdrhip_fm_chain_plan(const sdrhip_fm_chain* c, int64_t s0, int64_t s1, int64_t total_in, int64_t* q0, int64_t* q1,
                         int64_t* halo)
{
    SDRHIP_REQUIRE(c != nullptr && q0 && q1 && halo, "sdrhip_fm_chain_plan");
    SDRHIP_REQUIRE(s0 >= 0 && s1 >= s0, "sdrhip_fm_chain_plan");
    int64_t Q = c->total_q(total_in);
    int64_t a = c->first_q_from(s0), b = c->first_q_from(s1);
    if (a > Q) a = Q;
    if (b > Q) b = Q;
    *q0 = a;
    *q1 = b;
    int64_t h = 0;
    if (b > a) {
        h = c->end(b - 1) - s1;
        if (h < 0) h = 0;
    }
    *halo = h;
    return SDRHIP_OK;
}

int64_t sdrhip_fm_chain_ready(const sdrhip_fm_chain* c, int64_t n_samples)
{
    if (!c || n_samples < 0) return -1;
    if (c->end(0) > n_samples) return 0;
    int64_t lo = 0, hi = 1;                        // invariant: end(lo) <= n_samples < end(hi); end() is non-decreasing
    while (c->end(hi) <= n_samples) { lo = hi; hi *= 2; }
    while (lo + 1 < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (c->end(mid) <= n_samples) lo = mid; else hi = mid;
    }
    return hi;
}

int64_t sdrhip_fm_chain_max_halo(const sdrhip_fm_chain* c)
{
    if (!c) return -1;
    int64_t worst = 0;
    for (int64_t q = 0; q <= 2 * c->resamp.I + 2; q++) {
        int64_t len = c->end(q) - c->start(q);
        if (len > worst) worst = len;
    }
    return worst;
}

int64_t sdrhip_fm_chain_halo_samples(const sdrhip_fm_chain* c)
{
    const int64_t h = sdrhip_fm_chain_max_halo(c);
    return h < 0 ? h : (h + 7) / 8 * 8;      // whole 16-byte vectors of u8 IQ, the same on every rank
}

int sdrhip_fm_chain_halo_exchange(const sdrhip_fm_chain* chain, sdrhip_comm* comm, void* stream, uint8_t* d_buf, int64_t shard_samples)
{
    SDRHIP_REQUIRE(chain != nullptr && comm != nullptr && d_buf != nullptr && shard_samples > 0, "sdrhip_fm_chain_halo_exchange");
    const int64_t halo = sdrhip_fm_chain_halo_samples(chain);
    SDRHIP_REQUIRE(halo <= shard_samples, "sdrhip_fm_chain_halo_exchange: shard shorter than the halo");
    return sdrhip_halo_exchange(comm, stream, d_buf, d_buf + 2 * shard_samples, (size_t)(2 * halo));
}

size_t sdrhip_fm_chain_halo_staging_bytes(const sdrhip_fm_chain* chain, int count)
{
    if (!chain || count < 1) return 0;
    return 2 * (size_t)count * (size_t)(2 * sdrhip_fm_chain_halo_samples(chain));
}

int sdrhip_fm_chain_halo_exchange_batch(const sdrhip_fm_chain* chain, sdrhip_comm* comm, void* stream, uint8_t* d_buf, int64_t shard_samples,
                                        size_t row_bytes, int count, void* d_staging)
{
    SDRHIP_REQUIRE(chain != nullptr && comm != nullptr && d_buf != nullptr && shard_samples > 0 && count >= 1, "sdrhip_fm_chain_halo_exchange_batch");
    const size_t hb = (size_t)(2 * sdrhip_fm_chain_halo_samples(chain));            // bytes of one halo
    SDRHIP_REQUIRE((int64_t)hb <= 2 * shard_samples, "sdrhip_fm_chain_halo_exchange_batch: shard shorter than the halo");
    if (count == 1) return sdrhip_halo_exchange(comm, stream, d_buf, d_buf + 2 * shard_samples, hb);
    SDRHIP_REQUIRE(d_staging != nullptr && row_bytes >= (size_t)(2 * shard_samples) + hb, "sdrhip_fm_chain_halo_exchange_batch: rows hold a shard and its halo");
    hipStream_t s = (hipStream_t)stream;
    uint8_t* send = (uint8_t*)d_staging;
    uint8_t* recv = send + (size_t)count * hb;
    // heads of the count rows -> one contiguous message; the received message -> the count halo regions
    SDRHIP_CHECK_HIP(hipMemcpy2DAsync(send, hb, d_buf, row_bytes, hb, (size_t)count, hipMemcpyDeviceToDevice, s));
    const int rc = sdrhip_halo_exchange(comm, stream, send, recv, (size_t)count * hb);
    if (rc != SDRHIP_OK) return rc;
    SDRHIP_CHECK_HIP(hipMemcpy2DAsync(d_buf + 2 * shard_samples, row_bytes, recv, hb, hb, (size_t)count, hipMemcpyDeviceToDevice, s));
    return SDRHIP_OK;
}

size_t sdrhip_fm_chain_workspace_bytes(const sdrhip_fm_chain* c, int64_t n_in)
{
    if (!c || n_in < 0) return 0;
    int64_t nk = n_in / c->decim.factor + 4;
    int64_t nm = nk * c->resamp.I / c->resamp.D + 4;
    // + the overlap each of the (up to 16) sub-batches recomputes and its alignment padding
    const size_t conv = c->fused_first_stage() ? 0 : align_up((size_t)(n_in + 16) * 8, 256) + 16 * align_up((size_t)(c->decim.Lp + 16) * 8, 256);
    const size_t one = conv + align_up((size_t)nk * 8, 256) + align_up((size_t)nk * 4, 256) + align_up((size_t)nm * 4, 256) + 256 + 16 * (64 << 10);
    return c->overlap ? 2 * align_up(one, 256) : one;      // two runs in flight: one half per lane
}

namespace {
struct SubRange {
    int64_t q0, q1, m0, m1, ky0, ky1, kd0, kd1;
    size_t off_x, off_d, off_y, off_z;   // converted input (unfused first stage only), decimated, demodulated, resampled
    int64_t xa, xb;                      // samples [xa, xb) of the stream are converted into off_x
};
}  // namespace

static int chain_run_on(sdrhip_fm_chain* c, void* stream, const uint8_t* d_in_iq, int64_t s0, int64_t n_in,
                        float* d_audio, int64_t q0, int64_t q1, void* d_workspace, size_t workspace_bytes);

int sdrhip_fm_chain_run(sdrhip_fm_chain* c, void* stream, const uint8_t* d_in_iq, int64_t s0, int64_t n_in,
                        float* d_audio, int64_t q0, int64_t q1, void* d_workspace, size_t workspace_bytes)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_run");
    if (!c->overlap) return chain_run_on(c, stream, d_in_iq, s0, n_in, d_audio, q0, q1, d_workspace, workspace_bytes);
    // two runs in flight: this run goes to lane j, after everything queued on the caller's stream so far (its input's
    // producer); the caller's stream is then made to wait for the PREVIOUS run (lane 1 - j), not for this one
    hipStream_t s = (hipStream_t)stream;
    int rc = c->ensure_lanes();
    if (rc != SDRHIP_OK) return rc;
    const int j = (int)(c->run_idx++ & 1u);
    SDRHIP_CHECK_HIP(hipEventRecord(c->ev_in[j], s));
    SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->lane[j], c->ev_in[j], 0));
    const size_t half = (workspace_bytes / 2) & ~(size_t)255;
    rc = chain_run_on(c, (void*)c->lane[j], d_in_iq, s0, n_in, d_audio, q0, q1, d_workspace ? (char*)d_workspace + (size_t)j * half : nullptr, half);
    // (also when the run failed half way: kernels it did enqueue on the lane may still be reading the input and writing the
    // audio and the workspace half, and a later run or join must be ordered behind them)
    const hipError_t erec = hipEventRecord(c->ev_out[j], c->lane[j]);
    c->lane_busy[j] = true;
    if (rc != SDRHIP_OK) return rc;
    SDRHIP_CHECK_HIP(erec);
    if (c->lane_busy[1 - j]) SDRHIP_CHECK_HIP(hipStreamWaitEvent(s, c->ev_out[1 - j], 0));
    return SDRHIP_OK;
}

int sdrhip_fm_chain_join(sdrhip_fm_chain* c, void* stream)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_join");
    for (int j = 0; j < 2; j++)
        if (c->lane_busy[j]) {
            SDRHIP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, c->ev_out[j], 0));
            c->lane_busy[j] = false;
        }
    return SDRHIP_OK;
}

int sdrhip_fm_chain_set_overlap(sdrhip_fm_chain* c, int on)
{
    SDRHIP_REQUIRE(c != nullptr && (on == 0 || on == 1), "sdrhip_fm_chain_set_overlap");
    // switching modes with runs still in flight: drain them first (a host-side wait; this is a configuration call)
    for (int j = 0; j < 2; j++)
        if (c->lane_busy[j]) {
            SDRHIP_CHECK_HIP(hipStreamSynchronize(c->lane[j]));
            c->lane_busy[j] = false;
        }
    c->overlap = on;
    return SDRHIP_OK;
}

static int chain_run_on(sdrhip_fm_chain* c, void* stream, const uint8_t* d_in_iq, int64_t s0, int64_t n_in,
                        float* d_audio, int64_t q0, int64_t q1, void* d_workspace, size_t workspace_bytes)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_run");
    SDRHIP_REQUIRE(q1 >= q0 && q0 >= 0 && s0 >= 0 && n_in >= 0, "sdrhip_fm_chain_run");
    if (q1 == q0) return SDRHIP_OK;
    SDRHIP_REQUIRE(d_in_iq && d_audio && d_workspace, "sdrhip_fm_chain_run");
    hipStream_t s = (hipStream_t)stream;

    const int64_t nq = q1 - q0;
    if (c->small_chain_ok(nq)) {
        // launch-bound run: the whole chain in ONE kernel, nothing through the workspace
        const int64_t kd0 = c->resamp.in_offset(q0) > 0 ? c->resamp.in_offset(q0) - 1 : 0;
        const int64_t kd1 = c->resamp.in_offset(q1 + c->audio.Lp - 2) + c->y_reach();
        const int64_t n_lo = kd0 * c->decim.factor, n_hi = (kd1 - 1) * c->decim.factor + c->decim.Lp;
        if (n_lo < s0 || n_hi > s0 + n_in) {
            set_error("sdrhip_fm_chain_run: outputs [%lld,%lld) need samples [%lld,%lld) but d_in holds [%lld,%lld)",
                      (long long)q0, (long long)q1, (long long)n_lo, (long long)n_hi, (long long)s0, (long long)(s0 + n_in));
            return SDRHIP_ERR_ARG;
        }
        int rc2;
        if ((rc2 = c->decim.ensure_device()) != SDRHIP_OK || (rc2 = c->resamp.ensure_device()) != SDRHIP_OK ||
            (rc2 = c->audio.ensure_device()) != SDRHIP_OK) return rc2;
        hipEvent_t b = nullptr, e = nullptr;
        if (c->timing) {
            if ((rc2 = c->new_event(&b)) != SDRHIP_OK) return rc2;
            SDRHIP_CHECK_HIP(hipEventRecord(b, s));
        }
        const bool last_zero = (int)c->decim.h_plain.size() == c->decim.Lp && c->decim.h_plain[c->decim.Lp - 1] == 0.0f;
        const bool took = launch_fm_chain_small(s, d_in_iq, s0, n_in, d_audio, q0, q1, c->decim.factor, c->decim.Lp, c->decim.d_scaled, last_zero,
                                                c->resamp.d_groups, c->resamp.row_stride, c->resamp.nloop, c->resamp.increments.data(),
                                                c->resamp.num_groups, c->resamp.I, c->resamp.D, c->resamp.Lp, c->resamp.d_plain, c->resamp.ntaps,
                                                c->audio.d_taps, c->audio.ntaps_kernel, c->audio.d_cross, c->gain, c->block,
                                                c->small_chain_tile != 0 ? c->small_chain_tile : (c->input_over_link ? -1 : 0));
        if (took) {
            if (c->timing) {
                if ((rc2 = c->new_event(&e)) != SDRHIP_OK) return rc2;
                SDRHIP_CHECK_HIP(hipEventRecord(e, s));
                c->spans.push_back({5, b, e});
                c->runs++;
            }
            SDRHIP_CHECK_HIP(hipGetLastError());
            return SDRHIP_OK;
        }
        if (c->timing && c->ev_used > 0) c->ev_used--;      // not this configuration after all: the stage kernels below
    }

    // the stages' ranges, back to front
    SubRange r;
    size_t off = 0;
    {
        r.q0 = q0;
        r.q1 = q1;
        r.m0 = r.q0;
        r.m1 = r.q1 + c->audio.Lp - 1;                                                   // resampler outputs z[m0,m1)
        r.ky0 = c->resamp.in_offset(r.m0);
        r.ky1 = c->resamp.in_offset(r.m1 - 1) + c->y_reach();                           // demod outputs
        r.kd0 = r.ky0 > 0 ? r.ky0 - 1 : 0;                                               // decimator outputs
        r.kd1 = r.ky1;
        r.off_x = off;
        r.xa = r.xb = 0;
        if (!c->fused_first_stage()) {
            const int64_t a = r.kd0 * c->decim.factor;
            r.xa = s0 + ((a - s0) & ~(int64_t)7);                   // 16-byte aligned in the u8 stream
            r.xb = (r.kd1 - 1) * c->decim.factor + c->decim.Lp;
            off += align_up((size_t)(r.xb - r.xa) * 8, 256);
        }
        r.off_d = off;
        r.off_y = r.off_d + align_up((size_t)(r.kd1 - r.kd0) * 8, 256);
        r.off_z = r.off_y + align_up((size_t)(r.ky1 - r.ky0) * 4, 256);
        off = r.off_z + align_up((size_t)(r.m1 - r.m0) * 4, 256);
    }
    const int64_t n_lo = r.kd0 * c->decim.factor, n_hi = (r.kd1 - 1) * c->decim.factor + c->decim.Lp;
    if (n_lo < s0 || n_hi > s0 + n_in) {
        set_error("sdrhip_fm_chain_run: outputs [%lld,%lld) need samples [%lld,%lld) but d_in holds [%lld,%lld)",
                  (long long)q0, (long long)q1, (long long)n_lo, (long long)n_hi, (long long)s0, (long long)(s0 + n_in));
        return SDRHIP_ERR_ARG;
    }
    if (off > workspace_bytes) {
        set_error("sdrhip_fm_chain_run: workspace too small (%zu < %zu)", workspace_bytes, off);
        return SDRHIP_ERR_ARG;
    }
    char* ws = (char*)d_workspace;
    int rc;
    hipStream_t st = s;
    auto begin_span = [&](int stage, hipStream_t on, hipEvent_t* b) -> int {
        if (!c->timing) return SDRHIP_OK;
        int r2 = c->new_event(b);
        if (r2 != SDRHIP_OK) return r2;
        (void)stage;
        SDRHIP_CHECK_HIP(hipEventRecord(*b, on));
        return SDRHIP_OK;
    };
    auto end_span = [&](int stage, hipStream_t on, hipEvent_t b) -> int {
        if (!c->timing) return SDRHIP_OK;
        hipEvent_t e;
        int r2 = c->new_event(&e);
        if (r2 != SDRHIP_OK) return r2;
        SDRHIP_CHECK_HIP(hipEventRecord(e, on));
        c->spans.push_back({stage, b, e});
        return SDRHIP_OK;
    };
    if (c->timing) c->runs++;

    float* d_d = (float*)(ws + r.off_d);
    float* d_y = (float*)(ws + r.off_y);
    float* d_z = (float*)(ws + r.off_z);
    hipEvent_t b = nullptr;
    // K1+K2 on the caller's stream: u8 -> cfloat -> decimate (convert.c:37-50 fused into decimate.c:105-113)
    if ((rc = begin_span(0, s, &b)) != SDRHIP_OK) return rc;
    if (c->fused_first_stage()) {
        if ((rc = fir_run(&c->decim, s, d_in_iq, true, s0, d_d, r.kd0, r.kd1, c->block)) != SDRHIP_OK) return rc;
    } else {
        float* d_x = (float*)(ws + r.off_x);
        launch_convert_u8(s, d_in_iq + 2 * (r.xa - s0), d_x, 2 * (r.xb - r.xa));
        if ((rc = fir_run(&c->decim, s, d_x, false, r.xa, d_d, r.kd0, r.kd1, c->block)) != SDRHIP_OK) return rc;
    }
    if ((rc = end_span(0, s, b)) != SDRHIP_OK) return rc;
    if (c->tail_shape_ok(r.q1 - r.q0)) {
        if ((rc = c->resamp.ensure_device()) != SDRHIP_OK || (rc = c->audio.ensure_device()) != SDRHIP_OK) return rc;
        if ((rc = begin_span(4, st, &b)) != SDRHIP_OK) return rc;
        const bool took = launch_fm_tail_fused(st, d_d, r.kd0, r.kd1, r.ky0, r.ky1, d_audio + (r.q0 - q0), r.q0, r.q1, c->resamp.d_groups,
                                               c->resamp.row_stride, c->resamp.nloop, c->resamp.increments.data(), c->resamp.num_groups,
                                               c->resamp.I, c->resamp.D, c->resamp.Lp, c->resamp.d_plain, c->resamp.ntaps, c->audio.d_taps,
                                               c->audio.ntaps_kernel, c->audio.d_cross, c->gain, c->block);
        if (took) {
            if ((rc = end_span(4, st, b)) != SDRHIP_OK) return rc;
            SDRHIP_CHECK_HIP(hipGetLastError());
            return SDRHIP_OK;
        }
        if (c->timing && c->ev_used > 0) c->ev_used--;      // the span's begin event goes back to the pool
    }
    if (c->fuse_demod) {
        // K3+K4: fmDemod inside the resampler's tile loader on large batches (y never reaches HBM), a stand-alone fmDemod
        // launch first otherwise; timed as the resample stage
        if ((rc = begin_span(2, st, &b)) != SDRHIP_OK) return rc;
        if ((rc = resamp_run_demod(&c->resamp, st, d_d + 2 * (r.ky0 - r.kd0), r.ky0 > r.kd0, r.ky1 - r.ky0, d_y, r.ky0, d_z, r.m0, r.m1,
                                   c->block, c->block, nullptr)) != SDRHIP_OK) return rc;
        if ((rc = end_span(2, st, b)) != SDRHIP_OK) return rc;
    } else {
        // K3: fmDemod; at stream start the carried sample is 0 (Demod.hs:41)
        if ((rc = begin_span(1, st, &b)) != SDRHIP_OK) return rc;
        launch_fm_demod_fast(st, d_d + 2 * (r.ky0 - r.kd0), d_y, r.ky1 - r.ky0, r.ky0 > r.kd0, 0.0f, 0.0f);
        if ((rc = end_span(1, st, b)) != SDRHIP_OK) return rc;
        // K4: polyphase resample
        if ((rc = begin_span(2, st, &b)) != SDRHIP_OK) return rc;
        if ((rc = resamp_run(&c->resamp, st, d_y, r.ky0, d_z, r.m0, r.m1, c->block, c->block)) != SDRHIP_OK) return rc;
        if ((rc = end_span(2, st, b)) != SDRHIP_OK) return rc;
    }
    // K5: symmetric audio filter (+ fm.hs:40 `P.map (VG.map (* 0.2))` as the kernel's epilogue: a separate
    // f32 multiply of the rounded output)
    if ((rc = begin_span(3, st, &b)) != SDRHIP_OK) return rc;
    if ((rc = fir_run(&c->audio, st, d_z, false, r.m0, d_audio + (r.q0 - q0), r.q0, r.q1, c->block, c->gain)) != SDRHIP_OK) return rc;
    if ((rc = end_span(3, st, b)) != SDRHIP_OK) return rc;
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

// ---- one run with fixed arguments as a hipGraph: launch-bound batches (a 2^20-sample shard is nine small kernels) pay one
// graph launch instead of nine kernel launches.  Captured from the very code path sdrhip_fm_chain_run takes.
struct sdrhip_fm_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    ~sdrhip_fm_graph()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (cap) (void)hipStreamDestroy(cap);
    }
};

int sdrhip_fm_chain_graph_create(sdrhip_fm_graph** out, sdrhip_fm_chain* c, const uint8_t* d_in_iq, int64_t s0, int64_t n_in, float* d_audio,
                                 int64_t q0, int64_t q1, void* d_workspace, size_t workspace_bytes)
{
    SDRHIP_REQUIRE(out != nullptr && c != nullptr, "sdrhip_fm_chain_graph_create");
    *out = nullptr;
    SDRHIP_REQUIRE(!c->timing && !c->overlap, "sdrhip_fm_chain_graph_create: per-stage timing and two runs "
                                               "in flight record events on the chain's own streams: switch them off for a captured run");
    sdrhip_fm_graph* g = new sdrhip_fm_graph();
    hipError_t e = hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("sdrhip_fm_chain_graph_create: %s", hipGetErrorString(e)); delete g; return SDRHIP_ERR_HIP; }
    // a plain run first: tap uploads, kernel attributes and argument checks happen outside the capture
    int rc = sdrhip_fm_chain_run(c, (void*)g->cap, d_in_iq, s0, n_in, d_audio, q0, q1, d_workspace, workspace_bytes);
    if (rc == SDRHIP_OK && hipStreamSynchronize(g->cap) != hipSuccess) { set_error("sdrhip_fm_chain_graph_create: warm-up run failed"); rc = SDRHIP_ERR_HIP; }
    if (rc != SDRHIP_OK) { delete g; return rc; }
    e = hipStreamBeginCapture(g->cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); delete g; return SDRHIP_ERR_HIP; }
    rc = sdrhip_fm_chain_run(c, (void*)g->cap, d_in_iq, s0, n_in, d_audio, q0, q1, d_workspace, workspace_bytes);
    e = hipStreamEndCapture(g->cap, &g->graph);
    if (rc != SDRHIP_OK) { delete g; return rc; }
    if (e != hipSuccess || g->graph == nullptr) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); delete g; return SDRHIP_ERR_HIP; }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); delete g; return SDRHIP_ERR_HIP; }
    *out = g;
    return SDRHIP_OK;
}

int sdrhip_fm_chain_graph_launch(sdrhip_fm_graph* g, void* stream)
{
    SDRHIP_REQUIRE(g != nullptr && g->exec != nullptr, "sdrhip_fm_chain_graph_launch");
    SDRHIP_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return SDRHIP_OK;
}

void sdrhip_fm_chain_graph_destroy(sdrhip_fm_graph* g) { delete g; }

int sdrhip_fm_chain_set_fused_tail(sdrhip_fm_chain* c, int enable)
{
    SDRHIP_REQUIRE(c != nullptr && enable >= 0 && enable <= 2, "sdrhip_fm_chain_set_fused_tail");
    c->fused_tail = enable;
    return SDRHIP_OK;
}

int sdrhip_fm_chain_set_small_chain(sdrhip_fm_chain* c, int mode, int64_t max_outputs, int tile_outputs)
{
    SDRHIP_REQUIRE(c != nullptr && mode >= 0 && mode <= 2 && tile_outputs >= 0, "sdrhip_fm_chain_set_small_chain");
    c->small_chain = mode;
    c->small_chain_max = max_outputs > 0 ? max_outputs : kSmallChainAutoOutputs;
    c->small_chain_tile = tile_outputs;
    return SDRHIP_OK;
}

long long sdrhip_debug_small_chain_launches(void) { return fm_chain_small_launch_count(); }
long long sdrhip_debug_resample_cycle_launches(void) { return resample_cycle_launch_count(); }
long long sdrhip_debug_decimate_real16_launches(void) { return decimate_real16_launch_count(); }
void sdrhip_debug_set_systolic(int on) { set_systolic(on); }
long long sdrhip_debug_systolic_launches(void) { return systolic_launch_count(); }
void sdrhip_debug_systolic_plan(int count, int* nstrips, int* nwhole) { systolic_plan(count, nstrips, nwhole); }

int sdrhip_fm_chain_set_demod_fusion(sdrhip_fm_chain* c, int enable)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_set_demod_fusion");
    c->fuse_demod = enable != 0;
    return SDRHIP_OK;
}

int sdrhip_fm_chain_enable_timing(sdrhip_fm_chain* c, int enable)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_enable_timing");
    c->timing = enable != 0;
    c->spans.clear();
    c->ev_used = 0;
    c->runs = 0;
    return SDRHIP_OK;
}

int sdrhip_fm_chain_read_timing(sdrhip_fm_chain* c, double* ms_sum, int* runs)
{
    SDRHIP_REQUIRE(c != nullptr && ms_sum != nullptr && runs != nullptr, "sdrhip_fm_chain_read_timing");
    for (int i = 0; i < kStages; i++) ms_sum[i] = 0.0;
    for (const auto& sp : c->spans) {
        SDRHIP_CHECK_HIP(hipEventSynchronize(sp.e));
        float ms = 0.0f;
        SDRHIP_CHECK_HIP(hipEventElapsedTime(&ms, sp.b, sp.e));
        ms_sum[sp.stage] += ms;
    }
    *runs = c->runs;
    c->spans.clear();
    c->ev_used = 0;
    c->runs = 0;
    return SDRHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Host-block streaming front end of the chain: what a Haskell `Pipe (Vector CUChar) (Vector Float)`
// replacing the five middle stages of examples/fm/fm.hs:34-41 binds to.  Source blocks (u8 IQ, host
// memory, `block` samples each -- or a multiple) go in, audio blocks of exactly `block_size_out`
// floats come out, bit-identical to what the reference's four Pipes + convert + gain yield.
//
// Per submission the pinned staging buffer of the current slot holds [carried tail | new samples]
// contiguously: the tail (the ~4.4k samples earlier pushes delivered and later outputs still need) is
// kept in a small host-side history and copied in front of the new samples by the host (8 KB), so the
// device never shuffles it.  Then
//   * large submissions: ONE hipMemcpyAsync H2D on the upload stream, the chain on the compute stream,
//     one D2H on the download stream -- three HIP streams, two slots, upload of block i over compute
//     of i-1 over download of i-2;
//   * small submissions (<= kDirectSamples): NO copies at all -- the decimator kernel reads the pinned
//     host buffer directly over PCIe and the last kernel writes the audio straight into pinned host
//     memory: ONE kernel launch for a push of a few blocks (kernels_small.hip; two launches -- decimator with its seams,
//     fused tail -- where the one-kernel chain does not apply) and one event per push instead of ~15 API calls, which is
//     what the reference's own block size (8192 samples) needs to beat one CPU thread.
// Results lag at most nslots - 1 submissions (sdrhip_fm_stream_flush drains); with adaptive submission (the default for
// operators that run in place) a push that finds the next slot still busy is staged behind the earlier ones and leaves with them.
// ---------------------------------------------------------------------------
// Round 5: the caller-side copy of a LARGE push into the pinned staging buffer, split over a few threads.  A push of 4096 source
// blocks is a 64 MiB memcpy: one thread moves ~23 GB/s while the link behind it takes ~45 (profiles/r04_host_stream.txt: 11.4
// against 22.5 Gsample/s with the source writing the staging buffer itself) -- the copying push lost half the link to a
// single-threaded memcpy.  Helpers are started by the first large push and live as long as the operator; the caller copies its
// own share, so small pushes never touch them (SDRHIP_COPY_THREADS: helpers, default 3; 0 = plain memcpy).
namespace {
struct CopyPool {
    static constexpr size_t kMinBytes = 4u << 20;          // below this a push is one memcpy
    static constexpr size_t kPiece = 1u << 20;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    uint8_t* dst = nullptr;
    const uint8_t* src = nullptr;
    size_t bytes = 0;
    std::atomic<size_t> next{0};
    int generation = 0, active = 0;
    bool stop = false;
    // helper threads beside the caller's own (SDRHIP_COPY_THREADS, clamped to 0 .. 16; 0 = plain memcpy)
    int helpers = [] {
        const char* e = getenv("SDRHIP_COPY_THREADS");
        const long v = e ? strtol(e, nullptr, 10) : 3;
        return (int)(v < 0 ? 0 : v > 16 ? 16 : v);
    }();

    void drain()
    {
        for (;;) {
            const size_t o = next.fetch_add(kPiece, std::memory_order_relaxed);
            if (o >= bytes) return;
            memcpy(dst + o, src + o, bytes - o < kPiece ? bytes - o : kPiece);
        }
    }
    void worker()
    {
        int seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_job.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            drain();
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }
    void copy(uint8_t* d, const uint8_t* s, size_t n)
    {
        if (n < kMinBytes || helpers <= 0) {
            memcpy(d, s, n);
            return;
        }
        if (workers.empty()) {
            // a thread that cannot be started (std::system_error: resource limits) must not unwind through the extern "C" push: the
            // copy goes on with the helpers that did start, or as a plain memcpy
            try {
                for (int i = 0; i < helpers; i++) workers.emplace_back([this] { worker(); });
            } catch (const std::exception&) {
                helpers = (int)workers.size();
            }
            if (workers.empty()) {
                helpers = 0;
                memcpy(d, s, n);
                return;
            }
        }
        {
            std::lock_guard<std::mutex> lk(m);
            dst = d; src = s; bytes = n;
            next.store(0, std::memory_order_relaxed);
            active = (int)workers.size();
            generation++;
        }
        cv_job.notify_all();
        drain();                                            // the caller's own share
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_job.notify_all();
        for (auto& t : workers) t.join();
    }
};
}  // namespace

struct sdrhip_fm_stream {
    sdrhip_fm_chain* c = nullptr;
    CopyPool copier;
    int max_block = 0;
    int block_out = 0;
    // Slots: submissions in flight.  Two for operators that take large pushes (upload of block i over compute of i-1 over
    // download of i-2, each slot holding a pinned staging buffer of the largest push); FOUR for operators whose largest push
    // runs in place (round 3): such a push is one small kernel, i.e. ~20 us of latency end to end over PCIe, and the host is
    // done submitting it in ~6 -- the slots are what keeps the GPU fed.  Results then lag three pushes instead of one
    // (sdrhip_fm_stream_flush drains; SDRHIP_STREAM_SLOTS=2 restores the short lag).
    static constexpr int kMaxSlots = 4;
    int nslots = 2;
    hipStream_t compute[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};   // compute[0]: copy mode; compute[si]: in-place pushes of slot si
    hipStream_t up = nullptr, down = nullptr;
    DevBuf din[kMaxSlots];   // device input of the slots (copy mode)
    DevBuf ws[kMaxSlots];    // one workspace per compute stream
    int64_t N = 0;         // samples received so far
    int64_t q_done = 0;    // audio outputs computed so far
    int64_t head_cap = 0;  // samples of room in front of the staged samples (for the carried tail), multiple of 8
    std::vector<uint8_t> hist;   // the last `head_cap` samples of the stream (host copy)
    int64_t hist_n = 0;          // valid samples in hist (they are the stream's samples [N - hist_n, N))
    bool direct_ok = getenv("SDRHIP_NO_DIRECT_STREAM") == nullptr;
    // [tail | new] up to this many samples is read in place over PCIe (tunable for experiments: SDRHIP_DIRECT_SAMPLES)
    int64_t direct_samples = getenv("SDRHIP_DIRECT_SAMPLES") ? atoll(getenv("SDRHIP_DIRECT_SAMPLES")) : kDirectSamples;
    // round 6 (tools/stream_direct_threshold_probe.py, after the in-place pushes got the largest tile): in place 11.1 / 11.7 / 12.2 Gsample/s
    // at 48 / 64 / 96 blocks per push against 8.2 / 10.5 / 13.4 through the copy engines (zero-copy pushes; memcpy pushes cross at ~110 blocks)
    // ... and with the slot-stream staging copy (below) 19-24 Gsample/s from 2 to 160 blocks per push against 13-17 through the three-stream
    // copy path at 96 ... 128 blocks; the two meet at ~256 blocks (21.7 either way)
    static constexpr int64_t kDirectSamples = 200 * 8192;
    // pushes (or piled-up batches) of at least this many samples are copied to device memory on their slot's stream before the chain runs
    int64_t stage_samples = getenv("SDRHIP_STAGE_SAMPLES") ? atoll(getenv("SDRHIP_STAGE_SAMPLES")) : 2 * 8192;
    struct Slot {
        PinBuf hin, hout;
        DevBuf dout;
        hipEvent_t ev = nullptr, ev_up = nullptr, ev_k = nullptr;
        int64_t n_out = 0;
        bool busy = false;
        bool direct = false;       // the last submission ran in place: `ev` also releases the staging buffer
    } slot[kMaxSlots];
    int64_t pushes = 0;        // submissions so far (slot = pushes % nslots)
    int cur() const { return (int)(pushes % nslots); }
    int staged = 0;            // samples copied into the current slot's staging buffer, not yet submitted
    int coalesce = 0;          // submit once this many samples are staged (0: every push)
    int adaptive = 0;          // > 0: submit when the next slot is free, else keep staging up to this many samples
    int capacity() const
    {
        int c = coalesce > max_block ? coalesce : max_block;
        return adaptive > c ? adaptive : c;
    }
    // is slot si's last submission still running on the GPU?
    bool in_flight(int si) const { return slot[si].busy && hipEventQuery(slot[si].ev) == hipErrorNotReady; }
    std::vector<float> fifo;
    size_t head = 0;

    ~sdrhip_fm_stream()
    {
        for (hipStream_t st : {up, compute[0], compute[1], compute[2], compute[3], down})
            if (st) (void)hipStreamSynchronize(st);
        for (auto& sl : slot)
            for (hipEvent_t e : {sl.ev, sl.ev_up, sl.ev_k})
                if (e) (void)hipEventDestroy(e);
        for (hipStream_t st : {up, compute[0], compute[1], compute[2], compute[3], down})
            if (st) (void)hipStreamDestroy(st);
    }
    int ready() const { return (int)((fifo.size() - head) / (size_t)block_out); }
    // harvest, oldest first, every in-flight submission the GPU has finished (never waits)
    int harvest_done()
    {
        for (int64_t k = pushes - (nslots - 1); k < pushes; k++) {
            if (k < 0) continue;
            const int si = (int)(k % nslots);
            if (!slot[si].busy) continue;
            if (hipEventQuery(slot[si].ev) != hipSuccess) break;     // still running (an error surfaces in the blocking harvest)
            int rc = harvest(si);
            if (rc != SDRHIP_OK) return rc;
        }
        return SDRHIP_OK;
    }
    uint8_t* staged_base(Slot& sl) const { return (uint8_t*)sl.hin.p + 2 * head_cap; }    // where staged sample 0 lives
    int harvest(int si)
    {
        Slot& sl = slot[si];
        if (!sl.busy) return SDRHIP_OK;
        SDRHIP_CHECK_HIP(hipEventSynchronize(sl.ev));
        if (head > 0 && head == fifo.size()) { fifo.clear(); head = 0; }
        else if (head > (1u << 20) && head * 2 > fifo.size()) { fifo.erase(fifo.begin(), fifo.begin() + head); head = 0; }
        const size_t old = fifo.size();
        fifo.resize(old + (size_t)sl.n_out);
        memcpy(fifo.data() + old, sl.hout.p, (size_t)sl.n_out * sizeof(float));
        sl.busy = false;
        return SDRHIP_OK;
    }
};

extern "C" {

int sdrhip_fm_stream_create(sdrhip_fm_stream** out, sdrhip_fm_chain* chain, int max_block_samples, int block_size_out)
{
    SDRHIP_REQUIRE(out != nullptr && chain != nullptr && max_block_samples > 0 && block_size_out > 0, "sdrhip_fm_stream_create");
    SDRHIP_REQUIRE(chain->block == 0 || max_block_samples % chain->block == 0,
                   "sdrhip_fm_stream_create: blocks must be whole multiples of the chain's seam block");
    *out = nullptr;
    sdrhip_fm_stream* st = new sdrhip_fm_stream();
    st->c = chain;
    st->max_block = max_block_samples;
    st->block_out = block_size_out;
    // the carried tail never exceeds the receptive field of one audio output (+ the 8-sample alignment of its start)
    st->head_cap = (sdrhip_fm_chain_max_halo(chain) + 8 + 15) / 8 * 8;
    st->hist.resize((size_t)(2 * st->head_cap));
    // four slots when even the largest push runs in place (see the struct), else two
    {
        const char* env = getenv("SDRHIP_STREAM_SLOTS");
        int want = (st->direct_ok && st->head_cap + (int64_t)max_block_samples <= st->direct_samples) ? sdrhip_fm_stream::kMaxSlots : 2;
        if (env && atoi(env) >= 2 && atoi(env) <= sdrhip_fm_stream::kMaxSlots) want = atoi(env);
        st->nslots = want;
    }
    // adaptive submission by default for operators that run in place (sdrhip_fm_stream_set_adaptive; SDRHIP_STREAM_ADAPTIVE=0
    // switches the default off, =n caps it at n source blocks): up to what is still read in place over PCIe
    if (st->nslots == sdrhip_fm_stream::kMaxSlots) {
        const char* env = getenv("SDRHIP_STREAM_ADAPTIVE");
        const int64_t unit = chain->block > 0 ? chain->block : 8;
        int64_t cap = (st->direct_samples - st->head_cap) / unit;
        // ... but no more than 64 source blocks' worth (or two of the caller's largest pushes): the one-stream route is within 10 % of its full rate
        // with batches of that size (32: 13-15, 64: 18-21, 199: 20-22 Gsample/s), and what piles up beyond only adds to the push-to-audio lag
        const int64_t enough = std::max<int64_t>((64 * (int64_t)8192 + unit - 1) / unit, 2 * (((int64_t)max_block_samples + unit - 1) / unit));
        if (cap > enough) cap = enough;
        if (env && atoll(env) < cap) cap = atoll(env);
        cap *= unit;
        if (cap >= 2 * (int64_t)max_block_samples && cap <= (1 << 30)) st->adaptive = (int)cap;
    }
    hipError_t e = hipSuccess;
    for (int i = 0; i < st->nslots; i++)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->compute[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->down, hipStreamNonBlocking);
    for (auto& sl : st->slot)
        for (hipEvent_t* ev : {&sl.ev, &sl.ev_up, &sl.ev_k})
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("sdrhip_fm_stream_create: %s", hipGetErrorString(e));
        delete st;
        return SDRHIP_ERR_HIP;
    }
    *out = st;
    return SDRHIP_OK;
}

void sdrhip_fm_stream_destroy(sdrhip_fm_stream* st) { delete st; }

}  // extern "C"


// Submit everything staged in the current slot: [carried tail | staged samples] -> chain -> audio -> host.
static int stream_submit(sdrhip_fm_stream* st)
{
    sdrhip_fm_chain* c = st->c;
    const int n = st->staged;
    if (n == 0) return SDRHIP_OK;
    const int si = st->cur();
    sdrhip_fm_stream::Slot& sl = st->slot[si];
    int rc;
    // outputs whose receptive field is complete once these samples are in
    const int64_t N1 = st->N + n;
    int64_t q_new = sdrhip_fm_chain_ready(c, N1);
    if (q_new < st->q_done) q_new = st->q_done;
    // the tail starts at the first sample the next pending output needs, rounded down to a multiple of 8 samples
    // (16-byte aligned tiles for the LDS-tiled decimator)
    int64_t keep_from = c->start(st->q_done) & ~(int64_t)7;
    if (keep_from > st->N) keep_from = st->N & ~(int64_t)7;
    const int64_t tail = st->N - keep_from;
    if (tail > st->hist_n || tail > st->head_cap) {
        set_error("sdrhip_fm_stream: carried tail of %lld samples exceeds the history (%lld)", (long long)tail, (long long)st->hist_n);
        return SDRHIP_ERR_STATE;
    }
    uint8_t* first = st->staged_base(sl) - 2 * tail;               // 16-byte aligned: tail and head_cap are multiples of 8 samples
    if (tail > 0) memcpy(first, st->hist.data() + 2 * (st->hist_n - tail), (size_t)(2 * tail));
    // history for the next submission: the last head_cap samples of [tail | staged] (the tail alone may not reach back far
    // enough, the staged samples alone may be fewer than head_cap)
    {
        const int64_t have = tail + n;
        const int64_t keep = have < st->head_cap ? have : st->head_cap;
        memmove(st->hist.data(), first + 2 * (have - keep), (size_t)(2 * keep));
        st->hist_n = keep;
    }

    const int64_t n_out = q_new - st->q_done;
    const bool direct = st->direct_ok && tail + n <= st->direct_samples;
    sl.n_out = 0;
    // In-place pushes alternate between two compute streams (and workspaces): a push of one source block is a few small
    // kernels, i.e. latency, and nothing push i+1 computes depends on what push i left on the device (the carried tail
    // comes from the host-side history) -- so two consecutive pushes overlap on the GPU.
    hipStream_t cs = direct ? st->compute[si] : st->compute[0];
    DevBuf& wsb_buf = direct ? st->ws[si] : st->ws[0];
    if (n_out > 0) {
        const size_t wsb = sdrhip_fm_chain_workspace_bytes(c, tail + n);
        if (wsb > wsb_buf.cap) {
            // growing frees the old buffer: nothing may still be using it
            SDRHIP_CHECK_HIP(hipStreamSynchronize(cs));
        }
        if ((rc = wsb_buf.ensure(wsb)) != SDRHIP_OK) return rc;
        if ((rc = sl.hout.ensure((size_t)n_out * 4)) != SDRHIP_OK) return rc;
    }
    if (direct) {
        // zero-copy: the kernels read the pinned staging buffer and write the pinned result buffer themselves
        if (n_out > 0) {
            if (tail + n >= st->stage_samples) {
                // ONE pass over the link into device memory on the slot's own compute stream, then the chain on device memory (round 6):
                // read in place, the one-kernel chain fetches every sample ~1.95 times over PCIe (each tile re-reads its overlap),
                // and the link is what such a push costs -- 8 ... 64 blocks per push 10-11 -> 19-24 Gsample/s.  No second stream, no
                // event: the copy and the kernels of a slot are ordered by its stream.
                DevBuf& dbuf = st->din[si];
                if ((rc = dbuf.ensure((size_t)(tail + n) * 2 + 64)) != SDRHIP_OK) return rc;
                SDRHIP_CHECK_HIP(hipMemcpyAsync(dbuf.p, first, (size_t)(tail + n) * 2, hipMemcpyHostToDevice, cs));
                rc = sdrhip_fm_chain_run(c, (void*)cs, (const uint8_t*)dbuf.p, keep_from, tail + n, (float*)sl.hout.dev, st->q_done, q_new, wsb_buf.p, wsb_buf.cap);
            } else {
                // a lone source block (a paced real-time source: the GPU is idle when it arrives): the kernel reads the pinned buffer itself
                c->input_over_link = true;
                rc = sdrhip_fm_chain_run(c, (void*)cs, (const uint8_t*)sl.hin.dev_ptr(first), keep_from, tail + n,
                                         (float*)sl.hout.dev, st->q_done, q_new, wsb_buf.p, wsb_buf.cap);
                c->input_over_link = false;
            }
            if (rc != SDRHIP_OK) return rc;
            sl.n_out = n_out;
            sl.busy = true;
        }
        // ONE event per push: the results are in pinned memory and the staging buffer is free again when the kernels are done
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, cs));
        sl.direct = true;
    } else {
        DevBuf& dbuf = st->din[si];
        if ((rc = dbuf.ensure((size_t)(tail + n) * 2 + 64)) != SDRHIP_OK) return rc;
        // slot si's device buffer was last read by the chain run of submission i-2, harvested before this slot was reopened
        SDRHIP_CHECK_HIP(hipMemcpyAsync(dbuf.p, first, (size_t)(tail + n) * 2, hipMemcpyHostToDevice, st->up));
        SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_up, st->up));
        sl.direct = false;
        if (n_out > 0) {
            SDRHIP_CHECK_HIP(hipStreamWaitEvent(st->compute[0], sl.ev_up, 0));
            if ((rc = sl.dout.ensure((size_t)n_out * 4)) != SDRHIP_OK) return rc;
            if ((rc = sdrhip_fm_chain_run(c, (void*)st->compute[0], (const uint8_t*)dbuf.p, keep_from, tail + n, (float*)sl.dout.p,
                                          st->q_done, q_new, st->ws[0].p, st->ws[0].cap)) != SDRHIP_OK) return rc;
            SDRHIP_CHECK_HIP(hipEventRecord(sl.ev_k, st->compute[0]));
            SDRHIP_CHECK_HIP(hipStreamWaitEvent(st->down, sl.ev_k, 0));
            SDRHIP_CHECK_HIP(hipMemcpyAsync(sl.hout.p, sl.dout.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st->down));
            SDRHIP_CHECK_HIP(hipEventRecord(sl.ev, st->down));
            sl.n_out = n_out;
            sl.busy = true;
        }
    }
    st->q_done = q_new;
    st->N = N1;
    st->pushes++;
    st->staged = 0;
    return st->harvest(st->cur());      // the oldest submission: its slot is the next to be filled
}

// make the current slot's staging buffer writable (its previous upload / in-place read and its download are over)
static int stream_open_slot(sdrhip_fm_stream* st)
{
    sdrhip_fm_stream::Slot& sl = st->slot[st->cur()];
    int rc;
    if (st->staged == 0) {
        if ((rc = st->harvest(st->cur())) != SDRHIP_OK) return rc;
        SDRHIP_CHECK_HIP(hipEventSynchronize(sl.direct ? sl.ev : sl.ev_up));
    }
    return sl.hin.ensure((size_t)(st->head_cap + st->capacity()) * 2);
}

extern "C" {

int sdrhip_fm_stream_set_coalesce(sdrhip_fm_stream* st, int samples)
{
    SDRHIP_REQUIRE(st != nullptr && samples >= 0, "sdrhip_fm_stream_set_coalesce");
    SDRHIP_REQUIRE(st->staged == 0, "sdrhip_fm_stream_set_coalesce: samples are staged (flush first)");
    SDRHIP_REQUIRE(st->c->block == 0 || samples % st->c->block == 0, "sdrhip_fm_stream_set_coalesce: whole source blocks only");
    for (hipStream_t s : {st->up, st->compute[0], st->compute[1], st->compute[2], st->compute[3], st->down})
        if (s) SDRHIP_CHECK_HIP(hipStreamSynchronize(s));   // staging buffers may be reallocated
    st->coalesce = samples;
    return SDRHIP_OK;
}

int sdrhip_fm_stream_set_adaptive(sdrhip_fm_stream* st, int max_samples)
{
    SDRHIP_REQUIRE(st != nullptr && max_samples >= 0, "sdrhip_fm_stream_set_adaptive");
    SDRHIP_REQUIRE(st->staged == 0, "sdrhip_fm_stream_set_adaptive: samples are staged (flush first)");
    SDRHIP_REQUIRE(st->c->block == 0 || max_samples % st->c->block == 0, "sdrhip_fm_stream_set_adaptive: whole source blocks only");
    SDRHIP_REQUIRE(max_samples == 0 || max_samples >= 2 * st->max_block, "sdrhip_fm_stream_set_adaptive: room for at least two pushes");
    for (hipStream_t s : {st->up, st->compute[0], st->compute[1], st->compute[2], st->compute[3], st->down})
        if (s) SDRHIP_CHECK_HIP(hipStreamSynchronize(s));   // staging buffers may be reallocated
    st->adaptive = max_samples;
    return SDRHIP_OK;
}

uint8_t* sdrhip_fm_stream_input_buffer(sdrhip_fm_stream* st)
{
    if (st == nullptr) { set_error("sdrhip_fm_stream_input_buffer: null stream"); return nullptr; }
    // the caller may write up to max_block samples: make room for all of them behind what is already staged
    if (st->staged + st->max_block > st->capacity() && stream_submit(st) != SDRHIP_OK) return nullptr;
    if (stream_open_slot(st) != SDRHIP_OK) return nullptr;
    return st->staged_base(st->slot[st->cur()]) + (size_t)st->staged * 2;
}

int sdrhip_fm_stream_push(sdrhip_fm_stream* st, const uint8_t* iq, int n)
{
    SDRHIP_REQUIRE(st != nullptr && iq != nullptr && n > 0 && n <= st->max_block, "sdrhip_fm_stream_push");
    SDRHIP_REQUIRE(st->c->block == 0 || n % st->c->block == 0,
                   "sdrhip_fm_stream_push: the chain reproduces the seams of `block`-sample source buffers (fm.hs:17,24)");
    int rc;
    if (st->staged + n > st->capacity() && (rc = stream_submit(st)) != SDRHIP_OK) return rc;
    if ((rc = stream_open_slot(st)) != SDRHIP_OK) return rc;
    uint8_t* dst = st->staged_base(st->slot[st->cur()]) + (size_t)st->staged * 2;
    if (iq != dst) st->copier.copy(dst, iq, (size_t)n * 2);   // else: the caller filled our staging buffer in place
    st->staged += n;
    bool submit = st->staged >= st->coalesce;
    if (st->adaptive > 0 && st->coalesce == 0 && submit) {     // an explicit set_coalesce takes precedence: fixed batches
        // a GPU that keeps up gets every push at once (lowest latency); one that is still busy with the slot this submission
        // would move on to lets the pushes pile up in the staging buffer and takes them as ONE launch when it frees up
        const bool room = st->staged + st->max_block <= st->capacity();
        submit = !room || !st->in_flight((st->cur() + 1) % st->nslots);
    }
    if (submit) {
        if ((rc = stream_submit(st)) != SDRHIP_OK) return rc;
        // a push that went out also collects whatever the GPU has finished meanwhile: a source slower than the GPU gets the
        // audio of push i at push i + 1 instead of i + nslots - 1 (staged pushes skip the query)
        if ((rc = st->harvest_done()) != SDRHIP_OK) return rc;
    }
    return st->ready();
}

int sdrhip_fm_stream_poll(sdrhip_fm_stream* st)
{
    SDRHIP_REQUIRE(st != nullptr, "sdrhip_fm_stream_poll");
    int rc = st->harvest_done();
    if (rc != SDRHIP_OK) return rc;
    return st->ready();
}

int sdrhip_fm_stream_flush(sdrhip_fm_stream* st)
{
    SDRHIP_REQUIRE(st != nullptr, "sdrhip_fm_stream_flush");
    int rc;
    if ((rc = stream_submit(st)) != SDRHIP_OK) return rc;
    const int first = st->cur();        // oldest first: the audio goes into the fifo in push order
    for (int k = 0; k < st->nslots; k++)
        if ((rc = st->harvest((first + k) % st->nslots)) != SDRHIP_OK) return rc;
    return st->ready();
}

// ---- checkpoint / resume ------------------------------------------------------------------------
// The whole state of the operator between two pushes is small and explicit (the reference keeps it in Pipe
// closures -- overlap remainder, resampler phase, last demod sample, output fill level): the stream position, the number of
// audio samples produced, the last head_cap input samples, and the audio not yet popped.  Everything else (resampler group,
// demod history, seam positions) is a closed form of the position.
namespace {
struct StreamStateHeader {
    uint32_t magic, version;
    int64_t N, q_done, head_cap, hist_n, pending;   // pending: audio floats in the fifo
    int32_t block_out, chain_block;
    int64_t chain_halo;
};
constexpr uint32_t kStateMagic = 0x53444d46u;   // "FMDS"
}  // namespace

size_t sdrhip_fm_stream_state_bytes(sdrhip_fm_stream* st)
{
    if (st == nullptr) return 0;
    // exact: drains the operator exactly as sdrhip_fm_stream_save will (0 = the drain failed, sdrhip_last_error)
    if (sdrhip_fm_stream_flush(st) < 0) return 0;
    return sizeof(StreamStateHeader) + (size_t)(2 * st->hist_n) + (st->fifo.size() - st->head) * sizeof(float);
}

int sdrhip_fm_stream_save(sdrhip_fm_stream* st, void* buf, size_t capacity, size_t* used)
{
    SDRHIP_REQUIRE(st != nullptr && buf != nullptr && used != nullptr, "sdrhip_fm_stream_save");
    int rc = sdrhip_fm_stream_flush(st);             // submits what is staged, drains both slots into the fifo
    if (rc < 0) return rc;
    StreamStateHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kStateMagic;
    h.version = 1;
    h.N = st->N;
    h.q_done = st->q_done;
    h.head_cap = st->head_cap;
    h.hist_n = st->hist_n;
    h.pending = (int64_t)(st->fifo.size() - st->head);
    h.block_out = st->block_out;
    h.chain_block = st->c->block;
    h.chain_halo = sdrhip_fm_chain_max_halo(st->c);
    const size_t need = sizeof h + (size_t)(2 * h.hist_n) + (size_t)h.pending * sizeof(float);
    if (capacity < need) {
        set_error("sdrhip_fm_stream_save: %zu bytes needed, %zu given", need, capacity);
        return SDRHIP_ERR_ARG;
    }
    unsigned char* p = (unsigned char*)buf;
    memcpy(p, &h, sizeof h);
    p += sizeof h;
    memcpy(p, st->hist.data(), (size_t)(2 * h.hist_n));
    p += 2 * h.hist_n;
    if (h.pending > 0) memcpy(p, st->fifo.data() + st->head, (size_t)h.pending * sizeof(float));
    *used = need;
    return SDRHIP_OK;
}

int sdrhip_fm_stream_restore(sdrhip_fm_stream* st, const void* buf, size_t bytes)
{
    SDRHIP_REQUIRE(st != nullptr && buf != nullptr && bytes >= sizeof(StreamStateHeader), "sdrhip_fm_stream_restore");
    SDRHIP_REQUIRE(st->N == 0 && st->staged == 0 && st->pushes == 0, "sdrhip_fm_stream_restore: only into a stream that has not been pushed to");
    StreamStateHeader h;
    memcpy(&h, buf, sizeof h);
    SDRHIP_REQUIRE(h.magic == kStateMagic && h.version == 1, "sdrhip_fm_stream_restore: not a stream state");
    SDRHIP_REQUIRE(h.block_out == st->block_out && h.chain_block == st->c->block && h.head_cap == st->head_cap &&
                       h.chain_halo == sdrhip_fm_chain_max_halo(st->c),
                   "sdrhip_fm_stream_restore: the state belongs to a stream of another geometry (chain taps / block sizes)");
    SDRHIP_REQUIRE(h.hist_n >= 0 && h.hist_n <= h.head_cap && h.pending >= 0 && h.N >= h.hist_n && h.q_done >= 0,
                   "sdrhip_fm_stream_restore: inconsistent state");
    SDRHIP_REQUIRE(bytes >= sizeof h + (size_t)(2 * h.hist_n) + (size_t)h.pending * sizeof(float), "sdrhip_fm_stream_restore: truncated state");
    const unsigned char* p = (const unsigned char*)buf + sizeof h;
    memcpy(st->hist.data(), p, (size_t)(2 * h.hist_n));
    p += 2 * h.hist_n;
    st->hist_n = h.hist_n;
    st->N = h.N;
    st->q_done = h.q_done;
    st->fifo.resize((size_t)h.pending);
    st->head = 0;
    if (h.pending > 0) memcpy(st->fifo.data(), p, (size_t)h.pending * sizeof(float));
    return st->ready();
}

int sdrhip_fm_stream_pop(sdrhip_fm_stream* st, float* out, int capacity)
{
    SDRHIP_REQUIRE(st != nullptr && out != nullptr, "sdrhip_fm_stream_pop");
    if (st->ready() <= 0) return 0;
    SDRHIP_REQUIRE(capacity >= st->block_out, "sdrhip_fm_stream_pop: capacity smaller than the block");
    memcpy(out, st->fifo.data() + st->head, (size_t)st->block_out * sizeof(float));
    st->head += (size_t)st->block_out;
    return st->block_out;
}

}  // extern "C"
