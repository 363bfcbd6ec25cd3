// abi_device.cpp -- layer (2) of include/sdr_hip.h: descriptors + device-pointer
// `_run` calls, plus the memory/stream helpers.
#include <stdarg.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "descriptors.hpp"

namespace sdrhip {

// The "short seamed launch" scale v (sdrhip_set_small_launch_outputs; default 32768 outputs).  A seamed launch that short
// is launch-bound, so it decides its Cross outputs inside ONE launch instead of a main kernel plus fix-ups -- measured per
// stage at 8192 .. 131072 outputs:
//   real filter      <= v/2 outputs : the generic kernel           (8192: 15.7 -> 14.1 us per push; at 39k it loses 4 us)
//   real resampler   <= 2v outputs  : the generic kernel           (19.6k: 34 -> 22 us per push; 39k: 17.8 -> 11.9 us)
//   tiled decimator  <= 5v outputs  : Cross outputs in the tile kernel (1k: 2 -> 1 launch; 131k: 14.3 -> 10.3 us)
constexpr int kSmallSeamedLaunch = 32768;
// resampler launches of at least this many outputs take fmDemod into their tile loader (below, the two edge launches it adds
// cost more than the round trip of y through HBM it saves)
constexpr int kFusedDemodMinOutputs = 1 << 18;
static std::atomic<int> g_small_launch{getenv("SDRHIP_SMALL_LAUNCH") ? atoi(getenv("SDRHIP_SMALL_LAUNCH")) : kSmallSeamedLaunch};
int small_launch_outputs() { return g_small_launch.load(std::memory_order_relaxed); }

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static std::atomic<void (*)(int, const char*)> g_error_handler{nullptr};
void dropin_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (auto h = g_error_handler.load()) {
        h(code, g_err);
        throw DropinAbort{};          // the handler came back: unwind to the drop-in symbol, which returns to its caller
    }
    fprintf(stderr, "libsdr_hip: %s\n", g_err);
    abort();
}

int upload_floats(float** d, const std::vector<float>& h)
{
    *d = nullptr;
    size_t bytes = (h.size() ? h.size() : 1) * sizeof(float);
    SDRHIP_CHECK_HIP(hipMalloc((void**)d, bytes));
    if (!h.empty()) SDRHIP_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return SDRHIP_OK;
}

// First use uploads the taps.  Descriptors are immutable afterwards and may be shared by host threads, so the one
// mutation there is goes under a lock (d_taps / d_groups are published last).
static std::mutex g_upload_mu;

// A descriptor's taps live on the device that was current at its first use; using it from another device afterwards would
// hand kernels pointers into the wrong HBM: refused loudly (one descriptor per device, like one chain per device).
static int check_device(int* dev_of, const char* what)
{
    int dev = 0;
    SDRHIP_CHECK_HIP(hipGetDevice(&dev));
    if (*dev_of < 0) *dev_of = dev;
    if (*dev_of != dev) {
        set_error("%s: the descriptor's taps were uploaded to device %d, the current device is %d (create one descriptor per device)", what,
                  *dev_of, dev);
        return SDRHIP_ERR_STATE;
    }
    return SDRHIP_OK;
}

int FirDesc::ensure_device() const
{
    std::lock_guard<std::mutex> lk(g_upload_mu);
    int rc = check_device(&device, "filter / decimator");
    if (rc != SDRHIP_OK) return rc;
    if (d_taps) return SDRHIP_OK;
    if ((rc = upload_floats(&d_cross, h_plain)) != SDRHIP_OK) return rc;
    d_plain = d_cross;
    if (!h_scaled.empty() && (rc = upload_floats(&d_scaled, h_scaled)) != SDRHIP_OK) return rc;
    return upload_floats(&d_taps, h_kernel);
}
int ResampDesc::ensure_device() const
{
    std::lock_guard<std::mutex> lk(g_upload_mu);
    int rc = check_device(&device, "resampler");
    if (rc != SDRHIP_OK) return rc;
    if (d_groups) return SDRHIP_OK;
    if ((rc = upload_floats(&d_plain, h_plain)) != SDRHIP_OK) return rc;
    if (num_groups > 64) {
        // kernels.hpp ResampTable::ext: un-rotated prefix sums of the increments, then the groups' filter offsets
        std::vector<float> ext((size_t)2 * num_groups + 1);
        std::vector<int> e((size_t)2 * num_groups + 1, 0);
        for (int q = 0; q < num_groups; q++) { e[q + 1] = e[q] + increments[q]; e[num_groups + 1 + q] = offsets[q]; }
        memcpy(ext.data(), e.data(), e.size() * sizeof(int));
        float* d = nullptr;
        if ((rc = upload_floats(&d, ext)) != SDRHIP_OK) return rc;
        d_ext = reinterpret_cast<int*>(d);
    }
    return upload_floats(&d_groups, h_groups);
}

FirDesc::~FirDesc()
{
    if (d_taps) (void)hipFree(d_taps);
    if (d_cross) (void)hipFree(d_cross);
    if (d_scaled) (void)hipFree(d_scaled);
}
ResampDesc::~ResampDesc()
{
    if (d_groups) (void)hipFree(d_groups);
    if (d_plain) (void)hipFree(d_plain);
    if (d_ext) (void)hipFree(d_ext);
}

static bool order_ok(int order) { return order == SDRHIP_ORDER_SCALAR || order == SDRHIP_ORDER_SSE || order == SDRHIP_ORDER_AVX; }
static int real_lanes(int order) { return order == SDRHIP_ORDER_SCALAR ? 1 : order == SDRHIP_ORDER_SSE ? 4 : 8; }
static int cplx_lanes(int order) { return order == SDRHIP_ORDER_SCALAR ? 1 : order == SDRHIP_ORDER_SSE ? 2 : 4; }

// mkFilter / mkDecimator (Filter.hs:167-175, 282-290) and mkFilterC / mkDecimatorC
// (:196-209, 322-331): zero-pad to the SIMD multiple; complex SIMD variants get the
// taps DUPLICATED (Filter.hs:146-148) for the One kernel and plain for Cross.
// (mkFilterC's roundUp has its arguments swapped, Filter.hs:204, so the reference
// does not pad complex *filters* and over-reads the tap array when the count is not
// a SIMD multiple; we pad as mkDecimatorC does -- identical whenever the reference
// is well-defined.)
int fir_create(FirDesc* d, int order, bool cplx, int factor, const float* coeffs, int ncoeffs)
{
    SDRHIP_REQUIRE(order_ok(order), "fir_create");
    SDRHIP_REQUIRE(coeffs != nullptr && ncoeffs > 0, "fir_create");
    SDRHIP_REQUIRE(factor >= 1, "fir_create");
    d->order = order;
    d->cplx = cplx;
    d->sym = false;
    d->factor = factor;
    int mult = cplx ? cplx_lanes(order) : real_lanes(order);
    d->Lp = round_up(ncoeffs, mult);
    d->lanes = real_lanes(order);
    d->corder = order == SDRHIP_ORDER_SCALAR ? CO_SEQ : order == SDRHIP_ORDER_SSE ? CO_L2 : CO_L4;
    d->h_plain.assign(d->Lp, 0.0f);
    memcpy(d->h_plain.data(), coeffs, ncoeffs * sizeof(float));
    if (cplx && order != SDRHIP_ORDER_SCALAR) {
        d->h_kernel.resize(2 * d->Lp);
        for (int i = 0; i < d->Lp; i++) d->h_kernel[2 * i] = d->h_kernel[2 * i + 1] = d->h_plain[i];
        d->ntaps_kernel = 2 * d->Lp;
        // taps / 128 for the u8-fused tiled kernel (its loader keeps u - 128 instead of (u - 128)/128): only when the division
        // is exact for every tap, i.e. no nonzero tap below 2^-119 (nothing a filter design produces; such a descriptor simply
        // takes the convert-then-decimate route)
        bool exact = true;
        for (int i = 0; i < d->Lp; i++) {
            const float t = d->h_plain[i];
            if (t != 0.0f && !(fabsf(t) >= 0x1p-119f)) exact = false;      // NaN taps fail the test as well
        }
        if (exact) {
            d->h_scaled.resize(d->Lp);
            for (int i = 0; i < d->Lp; i++) d->h_scaled[i] = d->h_plain[i] * (1.0f / 128.0f);
        }
    } else {
        d->h_kernel = d->h_plain;
        d->ntaps_kernel = d->Lp;
    }
    return SDRHIP_OK;
}

// mkFilterSymR / mkDecimatorSymR (Filter.hs:234-245, 358-371): One kernel gets the
// half taps, Cross gets coeffs ++ reverse coeffs, numCoeffs = 2 * nhalf.
int fir_sym_create(FirDesc* d, int order, int factor, const float* half, int nhalf)
{
    SDRHIP_REQUIRE(order == SDRHIP_ORDER_SSE || order == SDRHIP_ORDER_AVX, "fir_sym_create (at least SSE4.2 required)");
    SDRHIP_REQUIRE(half != nullptr && nhalf > 0, "fir_sym_create");
    int lanes = real_lanes(order);
    SDRHIP_REQUIRE(nhalf % lanes == 0, "fir_sym_create: half-tap count must be a multiple of the SIMD width");
    d->order = order;
    d->cplx = false;
    d->sym = true;
    d->factor = factor;
    d->lanes = lanes;
    d->Lp = 2 * nhalf;
    d->ntaps_kernel = nhalf;
    d->h_kernel.assign(half, half + nhalf);
    d->h_plain.resize(2 * nhalf);
    for (int i = 0; i < nhalf; i++) {
        d->h_plain[i] = half[i];
        d->h_plain[2 * nhalf - 1 - i] = half[i];
    }
    return SDRHIP_OK;
}

int fir_run(const FirDesc* d, hipStream_t s, const void* d_in, bool in_u8, int64_t in_base, float* d_out,
            int64_t k_begin, int64_t k_end, int64_t seam_block, float gain)
{
    SDRHIP_REQUIRE(d != nullptr, "fir_run");
    SDRHIP_REQUIRE(k_end >= k_begin && k_end - k_begin < (int64_t)0x7fffffff, "fir_run");
    SDRHIP_REQUIRE(k_begin * d->factor >= in_base, "fir_run: first window starts before d_in");
    SDRHIP_REQUIRE(seam_block <= 0 || seam_block >= d->Lp, "fir_run: seam block shorter than the filter (Filter.hs:544,586)");
    if (k_end == k_begin) return SDRHIP_OK;
    SDRHIP_REQUIRE(d_in != nullptr && d_out != nullptr, "fir_run");
    {
        int rc = d->ensure_device();
        if (rc != SDRHIP_OK) return rc;
    }
    Geom g;
    g.in_base = in_base;
    g.k_begin = k_begin;
    g.count = (int)(k_end - k_begin);
    g.I = 1;
    g.D = d->factor;
    g.Lp = d->Lp;
    g.seamBI = seam_block;
    if (d->cplx) {
        if (d->corder == CO_L4 &&
            (!in_u8 || d->d_scaled) &&
            launch_decimate_c4_fast(s, g, in_u8 ? d->d_scaled : d->d_plain, d->Lp, d->d_cross, d_in, in_u8, d_out,
                                    in_u8 && (int)d->h_plain.size() == d->Lp && d->h_plain[d->Lp - 1] == 0.0f)) {
            // specialised kernel took it
        } else if (d->corder != CO_L4 && !d->sym && (!in_u8 || d->d_scaled) &&
                   launch_decimate_c_orders_fast(s, g, d->corder, in_u8 ? d->d_scaled : d->d_plain, d->Lp, d->d_cross, d_in, in_u8, d_out)) {
            // the same tiled kernel with the SSE / RC2 partial-sum layout took it
        } else if (d->corder == CO_L4 && !in_u8 && !d->sym &&
                   launch_filter_c4_tile(s, g, d->d_plain, d->Lp, d->d_cross, (const float*)d_in, d_out)) {
            // complex filter of exactly 128 / 64 taps: the tiled decimator with D = 1, eight outputs per thread
        } else if (d->corder == CO_L4 && !in_u8 &&
                   launch_filter_cplx4_fast(s, g, d->d_taps, d->Lp, d->d_cross, (const float*)d_in, d_out)) {
            // LDS-tiled complex filter took it
        } else if (!in_u8 && launch_fir_split(s, g, true, 0, d->corder, d->sym, d->sym ? d->d_taps : d->d_plain,
                                              d->sym ? d->ntaps_kernel : d->Lp, d->d_cross, (const float*)d_in, d_out, 1.0f, false)) {
            // lane-split tiled kernel took it (any factor / tap count / SIMD order)
        } else if (in_u8) {
            launch_fir_cplx_u8(s, g, d->corder, d->d_taps, d->ntaps_kernel, d->d_cross, (const uint8_t*)d_in, d_out);
        } else {
            launch_fir_cplx(s, g, d->corder, false, d->d_taps, d->ntaps_kernel, d->d_cross, (const float*)d_in, d_out);
        }
        if (gain != 1.0f) launch_scale(s, gain, d_out, d_out, 2 * (int64_t)g.count);
    } else {
        SDRHIP_REQUIRE(!in_u8, "fir_run: u8 input is IQ data, complex stages only");
        // A seamed launch this short is one host block passing through a Pipe: the generic kernel does it in ONE launch (Cross
        // outputs decided per output) where the tiled kernels need a second one for the seams, and launches are what such a
        // push costs (measured, 8192-float blocks: 15.7 -> 14.1 us per push).
        const bool small = g.seamBI > 0 && g.count <= small_launch_outputs() / 2;
        if (small) {
            launch_fir_real(s, g, d->lanes, d->sym, d->d_taps, d->ntaps_kernel, d->d_cross, (const float*)d_in, d_out);
            if (gain != 1.0f) launch_scale(s, gain, d_out, d_out, g.count);
        } else if ((d->lanes == 8 || d->lanes == 4) &&
            launch_fir_real8_fast(s, g, d->sym, d->d_taps, d->ntaps_kernel, d->d_cross, (const float*)d_in, d_out, gain, gain != 1.0f, d->lanes)) {
            // LDS-tiled kernel took it (gain fused): AVX order, and the SSE order when the tap count is a multiple of 8
        } else if (launch_decimate_real16_fast(s, g, d->lanes, d->sym ? d->d_taps : d->d_plain, d->sym ? d->ntaps_kernel : d->Lp, d->d_cross,
                                               (const float*)d_in, d_out, gain, gain != 1.0f, 0, d->sym)) {
            // real decimator by 2 / 4 / 8 / 16: 16 / D outputs per thread, scalar-loaded taps (gain fused)
        } else if (launch_fir_split(s, g, false, d->lanes, CO_SEQ, d->sym, d->sym ? d->d_taps : d->d_plain,
                                    d->sym ? d->ntaps_kernel : d->Lp, d->d_cross, (const float*)d_in, d_out, gain, gain != 1.0f)) {
            // lane-split tiled kernel took it (decimators, SSE order; gain fused)
        } else {
            launch_fir_real(s, g, d->lanes, d->sym, d->d_taps, d->ntaps_kernel, d->d_cross, (const float*)d_in, d_out);
            if (gain != 1.0f) launch_scale(s, gain, d_out, d_out, g.count);
        }
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

// prepareCoeffs, FilterInternal.hs:297-319
void prepare_coeffs(int n, int I, int D, const float* coeffs, int ncoeffs, int& num_coeffs, int& row_stride,
                    std::vector<int>& increments, std::vector<int>& offsets, std::vector<float>& groups)
{
    increments.clear();
    offsets.clear();
    int off = 0, maxlen = 0;
    do {
        int len = ncoeffs > off ? (ncoeffs - off + I - 1) / I : 0;
        if (len > maxlen) maxlen = len;
        offsets.push_back(off);
        increments.push_back((D - off - 1) / I + 1);
        off = I - 1 - (D - off - 1) % I;
    } while (off != 0 && (int)offsets.size() < I);
    num_coeffs = maxlen;
    row_stride = round_up(maxlen, n);
    groups.assign((size_t)offsets.size() * row_stride, 0.0f);
    for (size_t g = 0; g < offsets.size(); g++) {
        int j = 0;
        for (int i = offsets[g]; i < ncoeffs; i += I) groups[g * row_stride + j++] = coeffs[i];
    }
}

// mkResampler / mkResamplerC, Filter.hs:408-446 + FilterInternal.hs:335-373.
// "Only works if decimation > interpolation" (Filter.hs:641).
int resamp_create(ResampDesc* r, int order, bool cplx, int I, int D, const float* coeffs, int ncoeffs)
{
    SDRHIP_REQUIRE(order_ok(order), "resamp_create");
    SDRHIP_REQUIRE(coeffs != nullptr && ncoeffs > 0, "resamp_create");
    SDRHIP_REQUIRE(I >= 1 && D > I, "resamp_create: needs decimation > interpolation (Filter.hs:641)");
    r->order = order;
    r->cplx = cplx;
    r->I = I;
    r->D = D;
    r->ntaps = ncoeffs;
    int simd = real_lanes(order);  // sizeMultiple 1 / 4 / 8 for both real and complex (Filter.hs:448-502)
    r->lanes = simd;
    r->corder = order == SDRHIP_ORDER_SCALAR ? CO_SEQ : order == SDRHIP_ORDER_SSE ? CO_X2 : CO_X4;
    r->Lp = round_up(ncoeffs, I * simd);
    prepare_coeffs(simd, I, D, coeffs, ncoeffs, r->num_coeffs, r->row_stride, r->increments, r->offsets, r->h_groups);
    r->num_groups = (int)r->offsets.size();
    r->nloop = round_up(r->num_coeffs, simd);
    r->lut.assign(I, -1);
    for (int g = 0; g < r->num_groups; g++) r->lut[r->offsets[g]] = g;
    r->h_plain.assign(coeffs, coeffs + ncoeffs);
    return SDRHIP_OK;
}

int resamp_run(const ResampDesc* r, hipStream_t s, const float* d_in, int64_t in_base, float* d_out,
               int64_t k_begin, int64_t k_end, int64_t seam_block, int64_t out_block)
{
    return resamp_run_demod(r, s, nullptr, false, 0, d_in, in_base, d_out, k_begin, k_end, seam_block, out_block, nullptr);
}

// d_iq != nullptr: the inputs are still the decimator's complex output (d_iq = the sample whose phase step is input `in_base`,
// y_count inputs from there) and d_in is the buffer fmDemod would fill: the FM chain's shape gets fmDemod fused into the
// resampler's tile loader (*demod_fused = true; d_in is then written only where the lead / tail / seam kernels read it),
// every other shape a stand-alone fmDemod launch first.
int resamp_run_demod(const ResampDesc* r, hipStream_t s, const float* d_iq, bool iq_has_prev, int64_t y_count, const float* d_in,
                     int64_t in_base, float* d_out, int64_t k_begin, int64_t k_end, int64_t seam_block, int64_t out_block,
                     bool* demod_fused)
{
    if (demod_fused) *demod_fused = false;
    SDRHIP_REQUIRE(r != nullptr, "resamp_run");
    SDRHIP_REQUIRE(k_end >= k_begin && k_end - k_begin < (int64_t)0x7fffffff, "resamp_run");
    SDRHIP_REQUIRE(k_begin >= 0, "resamp_run");
    SDRHIP_REQUIRE(seam_block <= 0 || seam_block * r->I >= r->Lp, "resamp_run: seam block shorter than the filter (Filter.hs:691)");
    // With a padded filter shorter than the decimation step the reference's Pipe drops more inputs than a buffer holds
    // (`VG.drop usedInput bufIn`, Filter.hs:702-703) and silently loses its place in the stream at every buffer
    // boundary: there is no stream result to reproduce, so blocked streams refuse the configuration.
    SDRHIP_REQUIRE(seam_block <= 0 || r->Lp >= r->D, "resamp_run: padded filter shorter than the decimation step: the reference Pipe "
                                                      "mis-steps at buffer boundaries (Filter.hs:702-709); use seam_block = 0");
    if (k_end == k_begin) return SDRHIP_OK;
    SDRHIP_REQUIRE(d_in != nullptr && d_out != nullptr, "resamp_run");
    {
        int rc = r->ensure_device();
        if (rc != SDRHIP_OK) return rc;
    }
    int64_t p0 = r->in_offset(k_begin);
    SDRHIP_REQUIRE(p0 >= in_base, "resamp_run: first window starts before d_in");
    Geom g;
    g.in_base = in_base;
    g.k_begin = k_begin;
    g.count = (int)(k_end - k_begin);
    g.I = r->I;
    g.D = r->D;
    g.Lp = r->Lp;
    g.seamBI = seam_block < 0 ? -1 : seam_block * r->I;
    g.outB = out_block > 0 ? out_block : 0;
    ResampTable t;
    t.ngroups = r->num_groups;
    t.group0 = r->group(k_begin);
    t.pos0 = p0 - in_base;
    int acc = 0;
    for (int q = 0; q < r->num_groups; q++) {
        if (q < 64) t.pre[q] = acc;
        acc += r->increments[(t.group0 + q) % r->num_groups];
    }
    t.period = acc;
    t.row_stride = r->row_stride;
    t.nloop = r->nloop;
    t.ntaps_plain = r->ntaps;
    t.force_seq = 0;
    for (int q = 0; q < r->num_groups && q < 64; q++) t.fo[q] = r->offsets[q];
    if (r->num_groups > 64) t.ext = r->d_ext;       // per-group tables in device memory (ensure_device)
    // as in fir_run: one launch instead of up to four (lead-in, tiles, tail, seams) for a single host block
    // (configs[3]'s 65536-float blocks: 34 -> 22 us per push)
    const int64_t small_generic_r = 2 * (int64_t)small_launch_outputs();
    if (d_iq != nullptr) {
        SDRHIP_REQUIRE(!r->cplx && y_count > 0, "resamp_run_demod: fmDemod feeds a real resampler");
        const bool small = g.seamBI > 0 && g.count <= small_generic_r;
        if (!small && r->lanes == 8 && g.count >= kFusedDemodMinOutputs &&
            launch_resample_3_10_fast(s, g, t, r->increments.data(), r->d_groups, r->d_plain, d_in, d_out, d_iq, iq_has_prev, y_count)) {
            if (demod_fused) *demod_fused = true;
            SDRHIP_CHECK_HIP(hipGetLastError());
            return SDRHIP_OK;
        }
        launch_fm_demod_fast(s, d_iq, const_cast<float*>(d_in), y_count, iq_has_prev, 0.0f, 0.0f);
    }
    if (r->cplx) {
        if (launch_resample3c_fast(s, g, r->corder, t, r->increments.data(), r->d_groups, r->d_plain, d_in, d_out)) {
            // specialised complex 3-group kernel took it
        } else if (launch_resample_cycle_fast(s, g, r->lanes, t, r->increments.data(), r->d_groups, r->d_plain, d_in, d_out, true, r->corder)) {
            // thread-per-cycle kernel took it (odd decimations, "RC2" orders)
        } else if (!launch_resample_split(s, g, true, r->lanes, r->corder, t, r->d_groups, r->d_plain, d_in, d_out))
            launch_resample_cplx(s, g, r->corder, t, r->d_groups, r->d_plain, d_in, d_out);
    } else if (g.seamBI > 0 && g.count <= small_generic_r) {
        launch_resample_real(s, g, r->lanes, t, r->d_groups, r->d_plain, d_in, d_out);
    } else if ((r->lanes == 8 || r->lanes == 4) &&
               launch_resample_3_10_fast(s, g, t, r->increments.data(), r->d_groups, r->d_plain, d_in, d_out, nullptr, false, 0, r->lanes)) {
        // specialised 3-group kernel took it
    } else if (g.I == 1 && t.ext == nullptr && t.group0 == 0 &&
               launch_decimate_real16_fast(s, g, r->lanes, r->d_groups, t.nloop, r->d_plain, d_in, d_out, 1.0f, false, t.ntaps_plain)) {
        // interpolation 1: a decimator by 2 / 4 / 8 / 16 in the resampler's clothes (one group, the same lane order)
    } else if (launch_resample_cycle_fast(s, g, r->lanes, t, r->increments.data(), r->d_groups, r->d_plain, d_in, d_out)) {
        // thread-per-cycle kernel took it (odd decimations)
    } else if (launch_resample_split(s, g, false, r->lanes, r->corder, t, r->d_groups, r->d_plain, d_in, d_out)) {
        // lane-split tiled kernel took it (any I/D, SSE order)
    } else launch_resample_real(s, g, r->lanes, t, r->d_groups, r->d_plain, d_in, d_out);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

}  // namespace sdrhip

using namespace sdrhip;

extern "C" {

int sdrhip_set_small_launch_outputs(int outputs)
{
    return g_small_launch.exchange(outputs < 0 ? kSmallSeamedLaunch : outputs);
}

const char* sdrhip_version(void) { return "sdr_hip 0.1 (gfx950)"; }
const char* sdrhip_last_error(void) { return get_error(); }
void sdrhip_set_error_handler(void (*handler)(int code, const char* message)) { g_error_handler.store(handler); }

int sdrhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int sdrhip_set_device(int dev)
{
    SDRHIP_CHECK_HIP(hipSetDevice(dev));
    return SDRHIP_OK;
}
int sdrhip_device_name(char* buf, int buflen)
{
    SDRHIP_REQUIRE(buf != nullptr && buflen > 0, "sdrhip_device_name");
    int dev = 0;
    SDRHIP_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SDRHIP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SDRHIP_OK;
}

int sdrhip_malloc(void** dptr, size_t bytes)
{
    SDRHIP_REQUIRE(dptr != nullptr, "sdrhip_malloc");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return SDRHIP_ERR_NOMEM;
    }
    return SDRHIP_OK;
}
int sdrhip_free(void* dptr)
{
    if (dptr) SDRHIP_CHECK_HIP(hipFree(dptr));
    return SDRHIP_OK;
}
int sdrhip_malloc_host(void** hptr, size_t bytes)
{
    SDRHIP_REQUIRE(hptr != nullptr, "sdrhip_malloc_host");
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return SDRHIP_ERR_NOMEM;
    }
    return SDRHIP_OK;
}
int sdrhip_free_host(void* hptr)
{
    if (hptr) SDRHIP_CHECK_HIP(hipHostFree(hptr));
    return SDRHIP_OK;
}
int sdrhip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    SDRHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SDRHIP_OK;
}
int sdrhip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    SDRHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SDRHIP_OK;
}
int sdrhip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    SDRHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SDRHIP_OK;
}
int sdrhip_stream_create(void** stream)
{
    SDRHIP_REQUIRE(stream != nullptr, "sdrhip_stream_create");
    hipStream_t s;
    SDRHIP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return SDRHIP_OK;
}
int sdrhip_stream_destroy(void* stream)
{
    if (stream) SDRHIP_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return SDRHIP_OK;
}
int sdrhip_stream_sync(void* stream)
{
    SDRHIP_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SDRHIP_OK;
}

// ---- Filter -----------------------------------------------------------------
int sdrhip_filter_create(sdrhip_filter** f, int order, int data_complex, const float* coeffs, int ncoeffs)
{
    SDRHIP_REQUIRE(f != nullptr, "sdrhip_filter_create");
    *f = nullptr;
    sdrhip_filter* d = new sdrhip_filter();
    int rc = fir_create(d, order, data_complex != 0, 1, coeffs, ncoeffs);
    if (rc != SDRHIP_OK) { delete d; return rc; }
    *f = d;
    return SDRHIP_OK;
}
int sdrhip_filter_sym_create(sdrhip_filter** f, int order, const float* half, int nhalf)
{
    SDRHIP_REQUIRE(f != nullptr, "sdrhip_filter_sym_create");
    *f = nullptr;
    sdrhip_filter* d = new sdrhip_filter();
    int rc = fir_sym_create(d, order, 1, half, nhalf);
    if (rc != SDRHIP_OK) { delete d; return rc; }
    *f = d;
    return SDRHIP_OK;
}
int sdrhip_filter_num_coeffs(const sdrhip_filter* f) { return f ? f->Lp : SDRHIP_ERR_ARG; }
void sdrhip_filter_destroy(sdrhip_filter* f) { delete f; }
int sdrhip_filter_run(const sdrhip_filter* f, void* stream, const float* d_in, int64_t in_base, float* d_out,
                      int64_t k_begin, int64_t k_end, int64_t seam_block)
{
    return fir_run(f, (hipStream_t)stream, d_in, false, in_base, d_out, k_begin, k_end, seam_block);
}

// ---- Decimator ----------------------------------------------------------------
int sdrhip_decimator_create(sdrhip_decimator** d, int order, int data_complex, int factor, const float* coeffs,
                            int ncoeffs)
{
    SDRHIP_REQUIRE(d != nullptr, "sdrhip_decimator_create");
    *d = nullptr;
    sdrhip_decimator* p = new sdrhip_decimator();
    int rc = fir_create(p, order, data_complex != 0, factor, coeffs, ncoeffs);
    if (rc != SDRHIP_OK) { delete p; return rc; }
    *d = p;
    return SDRHIP_OK;
}
int sdrhip_decimator_sym_create(sdrhip_decimator** d, int order, int factor, const float* half, int nhalf)
{
    SDRHIP_REQUIRE(d != nullptr, "sdrhip_decimator_sym_create");
    *d = nullptr;
    sdrhip_decimator* p = new sdrhip_decimator();
    int rc = fir_sym_create(p, order, factor, half, nhalf);
    if (rc != SDRHIP_OK) { delete p; return rc; }
    *d = p;
    return SDRHIP_OK;
}
int sdrhip_decimator_num_coeffs(const sdrhip_decimator* d) { return d ? d->Lp : SDRHIP_ERR_ARG; }
int sdrhip_decimator_factor(const sdrhip_decimator* d) { return d ? d->factor : SDRHIP_ERR_ARG; }
void sdrhip_decimator_destroy(sdrhip_decimator* d) { delete d; }
int sdrhip_decimator_run(const sdrhip_decimator* d, void* stream, const float* d_in, int64_t in_base, float* d_out,
                         int64_t k_begin, int64_t k_end, int64_t seam_block)
{
    return fir_run(d, (hipStream_t)stream, d_in, false, in_base, d_out, k_begin, k_end, seam_block);
}
int sdrhip_decimator_run_u8(const sdrhip_decimator* d, void* stream, const uint8_t* d_in_iq, int64_t in_base,
                            float* d_out, int64_t k_begin, int64_t k_end, int64_t seam_block)
{
    SDRHIP_REQUIRE(d != nullptr && d->cplx, "sdrhip_decimator_run_u8: complex decimators only");
    return fir_run(d, (hipStream_t)stream, d_in_iq, true, in_base, d_out, k_begin, k_end, seam_block);
}

// ---- Resampler ------------------------------------------------------------------
int sdrhip_resampler_create(sdrhip_resampler** r, int order, int data_complex, int interpolation, int decimation,
                            const float* coeffs, int ncoeffs)
{
    SDRHIP_REQUIRE(r != nullptr, "sdrhip_resampler_create");
    *r = nullptr;
    sdrhip_resampler* p = new sdrhip_resampler();
    int rc = resamp_create(p, order, data_complex != 0, interpolation, decimation, coeffs, ncoeffs);
    if (rc != SDRHIP_OK) { delete p; return rc; }
    *r = p;
    return SDRHIP_OK;
}
int sdrhip_resampler_num_coeffs(const sdrhip_resampler* r) { return r ? r->Lp : SDRHIP_ERR_ARG; }
int sdrhip_resampler_num_groups(const sdrhip_resampler* r) { return r ? r->num_groups : SDRHIP_ERR_ARG; }
void sdrhip_resampler_destroy(sdrhip_resampler* r) { delete r; }
int64_t sdrhip_resampler_in_offset(const sdrhip_resampler* r, int64_t m) { return r ? r->in_offset(m) : -1; }
int sdrhip_resampler_filter_offset(const sdrhip_resampler* r, int64_t m) { return r ? r->filter_offset(m) : SDRHIP_ERR_ARG; }
int sdrhip_resampler_group(const sdrhip_resampler* r, int64_t m) { return r ? r->group(m) : SDRHIP_ERR_ARG; }
int sdrhip_resampler_run(const sdrhip_resampler* r, void* stream, const float* d_in, int64_t in_base, float* d_out,
                         int64_t k_begin, int64_t k_end, int64_t seam_block, int64_t out_block)
{
    return resamp_run(r, (hipStream_t)stream, d_in, in_base, d_out, k_begin, k_end, seam_block, out_block);
}

// ---- element-wise -----------------------------------------------------------------
int sdrhip_convert_u8_run(void* stream, const uint8_t* d_in, float* d_out, int64_t n_bytes)
{
    SDRHIP_REQUIRE(n_bytes >= 0, "sdrhip_convert_u8_run");
    launch_convert_u8((hipStream_t)stream, d_in, d_out, n_bytes);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}
int sdrhip_convert_i16_run(void* stream, const int16_t* d_in, float* d_out, int64_t n)
{
    SDRHIP_REQUIRE(n >= 0, "sdrhip_convert_i16_run");
    launch_convert_i16((hipStream_t)stream, d_in, d_out, n);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}
int sdrhip_scale_run(void* stream, float factor, const float* d_in, float* d_out, int64_t n)
{
    SDRHIP_REQUIRE(n >= 0, "sdrhip_scale_run");
    launch_scale((hipStream_t)stream, factor, d_in, d_out, n);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}
int sdrhip_fm_demod_run(void* stream, const float* d_in_iq, int64_t in_base, float* d_out, int64_t k_begin,
                        int64_t k_end, float last_re, float last_im)
{
    SDRHIP_REQUIRE(k_end >= k_begin && k_begin >= in_base, "sdrhip_fm_demod_run");
    bool has_prev = k_begin > in_base;
    launch_fm_demod_fast((hipStream_t)stream, d_in_iq + 2 * (k_begin - in_base), d_out, k_end - k_begin, has_prev, last_re,
                         last_im);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

long long sdrhip_debug_tiled_launches(void) { return split_launch_count(); }

size_t sdrhip_dc_blocker_workspace_bytes(int64_t n) { return dc_blocker_workspace_bytes(n); }

int sdrhip_dc_blocker_run(void* stream, const float* d_in, float* d_out, int64_t n, float last_sample, float last_output,
                          float* d_final, void* d_workspace, size_t workspace_bytes, int run_in)
{
    SDRHIP_REQUIRE(n >= 0 && d_final != nullptr && run_in >= 0, "sdrhip_dc_blocker_run");
    if (n == 0) return SDRHIP_OK;
    SDRHIP_REQUIRE(d_in != nullptr && d_out != nullptr && d_in != d_out, "sdrhip_dc_blocker_run: in-place is not supported");
    SDRHIP_REQUIRE(d_workspace == nullptr || workspace_bytes >= dc_blocker_workspace_bytes(n),
                   "sdrhip_dc_blocker_run: workspace smaller than sdrhip_dc_blocker_workspace_bytes(n)");
    launch_dc_blocker((hipStream_t)stream, n, last_sample, last_output, d_in, d_out, d_final, d_workspace, run_in);
    SDRHIP_CHECK_HIP(hipGetLastError());
    return SDRHIP_OK;
}

}  // extern "C"
