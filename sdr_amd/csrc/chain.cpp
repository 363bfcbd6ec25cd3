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
#include <vector>

#include "descriptors.hpp"

using namespace sdrhip;

static const int kStages = 5;  // decimate(+seam fix-up), fmDemod, resample, filter, gain

struct sdrhip_fm_chain {
    FirDesc decim;     // complex, factor D1
    ResampDesc resamp; // real I2/D2
    FirDesc audio;     // symmetric real
    float gain = 1.0f;
    int64_t block = 0;

    // optional per-stage timing: one set of kStages+1 events per run, on the run's stream
    bool timing = false;
    struct EvSet { hipEvent_t e[kStages + 1]; };
    std::vector<EvSet> pool;   // allocated sets
    size_t used = 0;           // sets recorded since the last read
    ~sdrhip_fm_chain()
    {
        for (auto& es : pool)
            for (auto ev : es.e) (void)hipEventDestroy(ev);
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

size_t sdrhip_fm_chain_workspace_bytes(const sdrhip_fm_chain* c, int64_t n_in)
{
    if (!c || n_in < 0) return 0;
    int64_t nk = n_in / c->decim.factor + 4;
    int64_t nm = nk * c->resamp.I / c->resamp.D + 4;
    return align_up((size_t)nk * 8, 256) + align_up((size_t)nk * 4, 256) + align_up((size_t)nm * 4, 256) + 256;
}

int sdrhip_fm_chain_run(sdrhip_fm_chain* c, void* stream, const uint8_t* d_in_iq, int64_t s0, int64_t n_in,
                        float* d_audio, int64_t q0, int64_t q1, void* d_workspace, size_t workspace_bytes)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_run");
    SDRHIP_REQUIRE(q1 >= q0 && q0 >= 0 && s0 >= 0 && n_in >= 0, "sdrhip_fm_chain_run");
    if (q1 == q0) return SDRHIP_OK;
    SDRHIP_REQUIRE(d_in_iq && d_audio && d_workspace, "sdrhip_fm_chain_run");
    hipStream_t s = (hipStream_t)stream;
    // ranges, back to front
    int64_t m0 = q0, m1 = q1 + c->audio.Lp - 1;                               // resampler outputs z[m0,m1)
    int64_t ky0 = c->resamp.in_offset(m0), ky1 = c->resamp.in_offset(m1 - 1) + c->y_reach();  // demod outputs
    int64_t kd0 = ky0 > 0 ? ky0 - 1 : 0, kd1 = ky1;                           // decimator outputs
    int64_t n_lo = kd0 * c->decim.factor, n_hi = (kd1 - 1) * c->decim.factor + c->decim.Lp;
    if (n_lo < s0 || n_hi > s0 + n_in) {
        set_error("sdrhip_fm_chain_run: outputs [%lld,%lld) need samples [%lld,%lld) but d_in holds [%lld,%lld)",
                  (long long)q0, (long long)q1, (long long)n_lo, (long long)n_hi, (long long)s0, (long long)(s0 + n_in));
        return SDRHIP_ERR_ARG;
    }
    size_t off_d = 0;
    size_t off_y = off_d + align_up((size_t)(kd1 - kd0) * 8, 256);
    size_t off_z = off_y + align_up((size_t)(ky1 - ky0) * 4, 256);
    size_t need = off_z + align_up((size_t)(m1 - m0) * 4, 256);
    if (need > workspace_bytes) {
        set_error("sdrhip_fm_chain_run: workspace too small (%zu < %zu)", workspace_bytes, need);
        return SDRHIP_ERR_ARG;
    }
    char* ws = (char*)d_workspace;
    float* d_d = (float*)(ws + off_d);
    float* d_y = (float*)(ws + off_y);
    float* d_z = (float*)(ws + off_z);
    int rc;
    sdrhip_fm_chain::EvSet* es = nullptr;
    if (c->timing) {
        if (c->used == c->pool.size()) {
            sdrhip_fm_chain::EvSet n;
            for (auto& ev : n.e) SDRHIP_CHECK_HIP(hipEventCreate(&ev));
            c->pool.push_back(n);
        }
        es = &c->pool[c->used++];
        SDRHIP_CHECK_HIP(hipEventRecord(es->e[0], s));
    }
#define STAGE_DONE(i) do { if (es) SDRHIP_CHECK_HIP(hipEventRecord(es->e[(i) + 1], s)); } while (0)
    // K1+K2: u8 -> cfloat -> decimate (convert.c:37-50 fused into decimate.c:105-113)
    if ((rc = fir_run(&c->decim, s, d_in_iq, true, s0, d_d, kd0, kd1, c->block)) != SDRHIP_OK) return rc;
    STAGE_DONE(0);
    // K3: fmDemod; at stream start the carried sample is 0 (Demod.hs:41)
    launch_fm_demod_fast(s, d_d + 2 * (ky0 - kd0), d_y, ky1 - ky0, ky0 > kd0, 0.0f, 0.0f);
    STAGE_DONE(1);
    // K4: polyphase resample
    if ((rc = resamp_run(&c->resamp, s, d_y, ky0, d_z, m0, m1, c->block)) != SDRHIP_OK) return rc;
    STAGE_DONE(2);
    // K5: symmetric audio filter
    // (+ fm.hs:40  P.map (VG.map (* 0.2)) as the kernel's epilogue: a separate f32 multiply of the rounded output)
    if ((rc = fir_run(&c->audio, s, d_z, false, m0, d_audio, q0, q1, c->block, c->gain)) != SDRHIP_OK) return rc;
    STAGE_DONE(3);
    STAGE_DONE(4);
#undef STAGE_DONE
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

int sdrhip_fm_chain_enable_timing(sdrhip_fm_chain* c, int enable)
{
    SDRHIP_REQUIRE(c != nullptr, "sdrhip_fm_chain_enable_timing");
    c->timing = enable != 0;
    c->used = 0;
    return SDRHIP_OK;
}

int sdrhip_fm_chain_read_timing(sdrhip_fm_chain* c, double* ms_sum, int* runs)
{
    SDRHIP_REQUIRE(c != nullptr && ms_sum != nullptr && runs != nullptr, "sdrhip_fm_chain_read_timing");
    for (int i = 0; i < kStages; i++) ms_sum[i] = 0.0;
    for (size_t r = 0; r < c->used; r++) {
        SDRHIP_CHECK_HIP(hipEventSynchronize(c->pool[r].e[kStages]));
        for (int i = 0; i < kStages; i++) {
            float ms = 0.0f;
            SDRHIP_CHECK_HIP(hipEventElapsedTime(&ms, c->pool[r].e[i], c->pool[r].e[i + 1]));
            ms_sum[i] += ms;
        }
    }
    *runs = (int)c->used;
    c->used = 0;
    return SDRHIP_OK;
}

}  // extern "C"
