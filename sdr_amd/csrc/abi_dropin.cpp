// abi_dropin.cpp -- layer (1) of include/sdr_hip.h: the reference's native entry
// points re-implemented on the GPU.  HOST pointers in/out, synchronous, no state
// retained past the call (the reference's FFI contract: VS.unsafeWith pointers are
// only valid during the call, FilterInternal.hs:66-71).
//
// Each symbol keeps the summation order of the x86 variant it replaces, so a
// Haskell program that resolves its `foreign import ccall` against this library
// produces the same bits whichever variant SDR.CPUID.featureSelect picks.
//
// Every call leases a scratch context (one HIP stream, pinned staging, device buffers, the taps it uploaded last) from
// a per-device pool, so concurrent pipeline threads run side by side instead of queueing on one mutex and one stream.
// Calls that move at most kDirectBytes each way -- the reference's own block sizes (8192 samples, fm.hs:17) -- run in
// place: the input is copied into pinned memory by the host, the kernel reads it and writes the pinned result over PCIe,
// nothing goes through the copy engines (a pageable hipMemcpyAsync pair costs more than such a kernel).  Larger calls use
// hipMemcpyAsync from / to the caller's buffers.  Taps are re-uploaded only when their bytes differ from the last call's.
#include <string.h>

#include <mutex>
#include <string>

#include "common.hpp"
#include "kernels.hpp"
#include "scratch_pool.hpp"

using namespace sdrhip;

namespace {

constexpr size_t kDirectBytes = 512 << 10;

using Ctx = ScratchCtx;

void die_if(int rc, const char* what)
{
    if (rc != SDRHIP_OK) {
        const std::string why = sdrhip_last_error();
        dropin_fail(rc, "%s: %s", what, why.c_str());
    }
}

// a leased context or dropin_fail(): the drop-in symbols cannot report
struct Lease {
    ScratchLease l;
    Ctx* c;
    Lease() : c(l.c)
    {
        if (!c) {
            const std::string why = sdrhip_last_error();
            dropin_fail(SDRHIP_ERR_HIP, "%s", why.c_str());
        }
    }
};

bool fits_direct(size_t in_bytes, size_t out_bytes) { return in_bytes <= kDirectBytes && out_bytes <= kDirectBytes; }

// device-visible copy of the caller's input
const void* stage_in(Ctx& c, bool direct, const void* h, size_t bytes)
{
    if (direct) {
        die_if(c.hin.ensure(bytes + 64), "pinned buffer");
        if (bytes) memcpy(c.hin.p, h, bytes);
        return c.hin.dev;
    }
    die_if(c.in.ensure(bytes + 64), "device buffer");
    if (bytes) SDRHIP_DIE_HIP(hipMemcpyAsync(c.in.p, h, bytes, hipMemcpyHostToDevice, c.stream));
    return c.in.p;
}
void* stage_out(Ctx& c, bool direct, size_t bytes)
{
    if (direct) {
        die_if(c.hout.ensure(bytes + 64), "pinned buffer");
        return c.hout.dev;
    }
    die_if(c.out.ensure(bytes + 64), "device buffer");
    return c.out.p;
}
void finish(Ctx& c, bool direct, void* h, size_t bytes)
{
    SDRHIP_DIE_HIP(hipGetLastError());
    if (direct) {
        SDRHIP_DIE_HIP(hipStreamSynchronize(c.stream));
        if (bytes) memcpy(h, c.hout.p, bytes);
        return;
    }
    if (bytes) SDRHIP_DIE_HIP(hipMemcpyAsync(h, c.out.p, bytes, hipMemcpyDeviceToHost, c.stream));
    SDRHIP_DIE_HIP(hipStreamSynchronize(c.stream));
}
// the taps on the device; uploaded only when they differ from what the buffer holds
const float* taps_on_device(Ctx& c, DevBuf& b, std::vector<unsigned char>& now, const void* h, size_t bytes)
{
    if (b.p != nullptr && now.size() == bytes && (bytes == 0 || memcmp(now.data(), h, bytes) == 0)) return (const float*)b.p;
    die_if(b.ensure(bytes + 64), "device buffer");
    now.assign((const unsigned char*)h, (const unsigned char*)h + bytes);
    // `now` outlives the call, so the copy may still be in flight when this returns; every entry point ends on a sync
    if (bytes) SDRHIP_DIE_HIP(hipMemcpyAsync(b.p, now.data(), bytes, hipMemcpyHostToDevice, c.stream));
    return (const float*)b.p;
}

Geom flat_geom(int num, int D, int Lp)
{
    Geom g;
    g.in_base = 0;
    g.k_begin = 0;
    g.count = num;
    g.I = 1;
    g.D = D;
    g.Lp = Lp;
    g.seamBI = 0;
    return g;
}

// real taps, real data
void fir_real(int lanes, bool sym, int num, int factor, int numCoeffs, float* coeffs, float* in, float* out)
try {
    if (num <= 0) return;
    Lease l;
    Ctx& c = *l.c;
    int span = sym ? 2 * numCoeffs : numCoeffs;
    size_t nin = (size_t)(num - 1) * factor + span;
    const bool direct = fits_direct(nin * 4, (size_t)num * 4);
    const float* d_taps = taps_on_device(c, c.taps, c.taps_now, coeffs, (size_t)numCoeffs * 4);
    const float* d_in = (const float*)stage_in(c, direct, in, nin * 4);
    float* d_out = (float*)stage_out(c, direct, (size_t)num * 4);
    launch_fir_real(c.stream, flat_geom(num, factor, span), lanes, sym, d_taps, numCoeffs, nullptr, d_in, d_out);
    finish(c, direct, out, (size_t)num * 4);
} catch (const DropinAbort&) {}

// real taps, complex data.  numCoeffs is the length of the array as passed.
void fir_cplx(ComplexOrder order, bool sym, int num, int factor, int numCoeffs, float* coeffs, float* in, float* out)
try {
    if (num <= 0) return;
    Lease l;
    Ctx& c = *l.c;
    int P = (order == CO_L2 || order == CO_L4) ? numCoeffs / 2 : numCoeffs;  // complex taps walked
    int span = sym ? 2 * P : P;
    size_t nin = (size_t)(num - 1) * factor + span;
    const bool direct = fits_direct(nin * 8, (size_t)num * 8);
    const float* d_in = (const float*)stage_in(c, direct, in, nin * 8);
    float* d_out = (float*)stage_out(c, direct, (size_t)num * 8);
    Geom g = flat_geom(num, factor, span);
    bool done = false;
    std::vector<float> plain;
    if (order == CO_L4 && !sym) {
        // the LDS-tiled kernel wants plain taps; only valid when the array really is a duplicate
        bool dup = true;
        for (int i = 0; i < P && dup; i++) dup = memcmp(&coeffs[2 * i], &coeffs[2 * i + 1], 4) == 0;
        if (dup) {
            plain.resize(P);
            for (int i = 0; i < P; i++) plain[i] = coeffs[2 * i];
            const float* d_plain = taps_on_device(c, c.taps2, c.taps2_now, plain.data(), (size_t)P * 4);
            done = launch_decimate_c4_fast(c.stream, g, d_plain, P, nullptr, d_in, false, d_out);
        }
    }
    if (!done) {
        const float* d_taps = taps_on_device(c, c.taps, c.taps_now, coeffs, (size_t)numCoeffs * 4);
        launch_fir_cplx(c.stream, g, order, sym, d_taps, numCoeffs, nullptr, d_in, d_out);
    }
    finish(c, direct, out, (size_t)num * 8);
} catch (const DropinAbort&) {}

// polyphase resamplers (resample.c:34-142)
int resample_groups(bool cplx, int lanes_or_order, int buf_size, int num_coeffs, int starting_group, int num_groups,
                    int* increments, float** coeffs, float* in, float* out)
try {
    if (num_groups <= 0) return starting_group;       // the reference's loop would divide by zero: nothing to compute
    int end_group = (int)(((int64_t)starting_group + (buf_size > 0 ? buf_size : 0)) % num_groups);
    if (buf_size <= 0) return starting_group;
    Lease l;
    Ctx& c = *l.c;
    int simd;
    ComplexOrder co = CO_SEQ;
    if (cplx) {
        co = (ComplexOrder)lanes_or_order;
        simd = co == CO_SEQ ? 1 : co == CO_X2 ? 4 : 8;
    } else {
        simd = lanes_or_order;
    }
    int nloop = round_up(num_coeffs, simd);
    std::vector<float> table((size_t)num_groups * nloop);
    for (int g = 0; g < num_groups; g++) memcpy(&table[(size_t)g * nloop], coeffs[g], (size_t)nloop * 4);
    ResampTable t;
    t.ngroups = num_groups;
    t.group0 = starting_group;
    t.pos0 = 0;
    // prefix of the increments starting at starting_group; up to 64 groups it travels in the kernel arguments, beyond that as
    // a device table of the un-rotated prefix sums (kernels.hpp: ResampTable::ext) -- the reference's C takes any count
    std::vector<int> pre((size_t)num_groups);
    int acc = 0;
    for (int q = 0; q < num_groups; q++) {
        pre[q] = acc;
        acc += increments[(starting_group + q) % num_groups];
    }
    t.period = acc;
    std::vector<int> ext;
    if (num_groups <= 64) {
        for (int q = 0; q < num_groups; q++) { t.pre[q] = pre[q]; t.fo[q] = 0; }
    } else {
        ext.assign((size_t)2 * num_groups + 1, 0);
        for (int q = 0; q < num_groups; q++) ext[q + 1] = ext[q] + increments[q];
    }
    t.row_stride = nloop;
    t.nloop = nloop;
    t.ntaps_plain = 0;
    t.force_seq = 0;
    int64_t last = buf_size - 1;
    size_t nin = (size_t)((last / num_groups) * (int64_t)t.period + pre[last % num_groups] + nloop);
    size_t esz = cplx ? 8 : 4;
    const bool direct = fits_direct(nin * esz, (size_t)buf_size * esz);
    const float* d_taps = taps_on_device(c, c.taps, c.taps_now, table.data(), table.size() * 4);
    if (!ext.empty()) t.ext = (const int*)taps_on_device(c, c.taps2, c.taps2_now, ext.data(), ext.size() * 4);
    const float* d_in = (const float*)stage_in(c, direct, in, nin * esz);
    float* d_out = (float*)stage_out(c, direct, (size_t)buf_size * esz);
    Geom g = flat_geom(buf_size, 1, 0);
    if (cplx) launch_resample_cplx(c.stream, g, co, t, d_taps, nullptr, d_in, d_out);
    else launch_resample_real(c.stream, g, simd, t, d_taps, nullptr, d_in, d_out);
    finish(c, direct, out, (size_t)buf_size * esz);
    return end_group;
} catch (const DropinAbort&) { return starting_group; }

template <class Fn>
void elementwise(const void* in, size_t in_bytes, void* out, size_t out_bytes, Fn launch)
try {
    Lease l;
    Ctx& c = *l.c;
    const bool direct = fits_direct(in_bytes, out_bytes);
    const void* d_in = stage_in(c, direct, in, in_bytes);
    void* d_out = stage_out(c, direct, out_bytes);
    launch(c.stream, d_in, d_out);
    finish(c, direct, out, out_bytes);
} catch (const DropinAbort&) {}

}  // namespace

extern "C" {

// ---- convert.c ------------------------------------------------------------------
// The three u8 variants are bit-identical in the reference (exact results); unlike
// convertCSSE/AVX (convert.c:27,42) this never reads past in[num).
void convertC(int num, uint8_t* in, float* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num, out, (size_t)num * 4,
                [&](hipStream_t s, const void* di, void* dout) { launch_convert_u8(s, (const uint8_t*)di, (float*)dout, num); });
}
void convertCSSE(int num, uint8_t* in, float* out) { convertC(num, in, out); }
void convertCAVX(int num, uint8_t* in, float* out) { convertC(num, in, out); }

void convertCBladeRF(int num, int16_t* in, float* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num * 2, out, (size_t)num * 4,
                [&](hipStream_t s, const void* di, void* dout) { launch_convert_i16(s, (const int16_t*)di, (float*)dout, num); });
}
void convertCSSEBladeRF(int num, int16_t* in, float* out) { convertCBladeRF(num, in, out); }
void convertCAVXBladeRF(int num, int16_t* in, float* out) { convertCBladeRF(num, in, out); }
void convertBladeRFTransmit(int num, float* in, int16_t* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num * 4, out, (size_t)num * 2, [&](hipStream_t s, const void* di, void* dout) {
        launch_convert_f32_to_i16_bladerf(s, (const float*)di, (int16_t*)dout, num);
    });
}

// ---- scale.c ----------------------------------------------------------------------
void scale(int num, float factor, float* in_buf, float* out_buf)
{
    if (num <= 0) return;
    elementwise(in_buf, (size_t)num * 4, out_buf, (size_t)num * 4,
                [&](hipStream_t s, const void* di, void* dout) { launch_scale(s, factor, (const float*)di, (float*)dout, num); });
}
void scaleSSE(int num, float factor, float* in_buf, float* out_buf) { scale(num, factor, in_buf, out_buf); }
void scaleAVX(int num, float factor, float* in_buf, float* out_buf) { scale(num, factor, in_buf, out_buf); }

// ---- filter.c ---------------------------------------------------------------------
void filterRR(int num, int numCoeffs, float* c, float* in, float* out) { fir_real(1, false, num, 1, numCoeffs, c, in, out); }
void filterSSERR(int num, int numCoeffs, float* c, float* in, float* out) { fir_real(4, false, num, 1, numCoeffs, c, in, out); }
void filterAVXRR(int num, int numCoeffs, float* c, float* in, float* out) { fir_real(8, false, num, 1, numCoeffs, c, in, out); }
void filterSSESymmetricRR(int num, int numCoeffs, float* c, float* in, float* out) { fir_real(4, true, num, 1, numCoeffs, c, in, out); }
void filterAVXSymmetricRR(int num, int numCoeffs, float* c, float* in, float* out) { fir_real(8, true, num, 1, numCoeffs, c, in, out); }
void filterRC(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_SEQ, false, num, 1, numCoeffs, c, in, out); }
void filterSSERC(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_L2, false, num, 1, numCoeffs, c, in, out); }
void filterSSERC2(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X2, false, num, 1, numCoeffs, c, in, out); }
void filterAVXRC(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_L4, false, num, 1, numCoeffs, c, in, out); }
void filterAVXRC2(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X4, false, num, 1, numCoeffs, c, in, out); }
void filterSSESymmetricRC(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X2, true, num, 1, numCoeffs, c, in, out); }
void filterAVXSymmetricRC(int num, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X4, true, num, 1, numCoeffs, c, in, out); }

void dcBlocker(int num, float lastSample, float lastOutput, float* finalSample, float* finalOutput, float* inBuf,
               float* outBuf)
try {
    if (num <= 0) {
        *finalSample = lastSample;
        *finalOutput = lastOutput;
        return;
    }
    Lease l;
    Ctx& c = *l.c;
    const size_t ob = (size_t)num * 4 + 16;   // {finalSample, finalOutput, -, -} then the block, 16-byte aligned
    const bool direct = fits_direct((size_t)num * 4, ob);
    const float* d_in = (const float*)stage_in(c, direct, inBuf, (size_t)num * 4);
    float* dout = (float*)stage_out(c, direct, ob);
    die_if(c.work.ensure(dc_blocker_workspace_bytes(num)), "device buffer");
    launch_dc_blocker(c.stream, num, lastSample, lastOutput, d_in, dout + 4, dout, c.work.p, 0);
    SDRHIP_DIE_HIP(hipGetLastError());
    float fin[2];
    if (direct) {
        SDRHIP_DIE_HIP(hipStreamSynchronize(c.stream));
        memcpy(fin, c.hout.p, 8);
        memcpy(outBuf, (const float*)c.hout.p + 4, (size_t)num * 4);
    } else {
        SDRHIP_DIE_HIP(hipMemcpyAsync(fin, dout, 8, hipMemcpyDeviceToHost, c.stream));
        SDRHIP_DIE_HIP(hipMemcpyAsync(outBuf, dout + 4, (size_t)num * 4, hipMemcpyDeviceToHost, c.stream));
        SDRHIP_DIE_HIP(hipStreamSynchronize(c.stream));
    }
    *finalSample = fin[0];
    *finalOutput = fin[1];
} catch (const DropinAbort&) {}

// ---- decimate.c ---------------------------------------------------------------------
void decimateRR(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_real(1, false, num, factor, numCoeffs, c, in, out); }
void decimateSSERR(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_real(4, false, num, factor, numCoeffs, c, in, out); }
void decimateAVXRR(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_real(8, false, num, factor, numCoeffs, c, in, out); }
void decimateSSESymmetricRR(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_real(4, true, num, factor, numCoeffs, c, in, out); }
void decimateAVXSymmetricRR(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_real(8, true, num, factor, numCoeffs, c, in, out); }
void decimateRC(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_SEQ, false, num, factor, numCoeffs, c, in, out); }
void decimateSSERC(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_L2, false, num, factor, numCoeffs, c, in, out); }
void decimateSSERC2(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X2, false, num, factor, numCoeffs, c, in, out); }
void decimateAVXRC(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_L4, false, num, factor, numCoeffs, c, in, out); }
void decimateAVXRC2(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X4, false, num, factor, numCoeffs, c, in, out); }
void decimateSSESymmetricRC(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X2, true, num, factor, numCoeffs, c, in, out); }
void decimateAVXSymmetricRC(int num, int factor, int numCoeffs, float* c, float* in, float* out) { fir_cplx(CO_X4, true, num, factor, numCoeffs, c, in, out); }

// ---- resample.c ---------------------------------------------------------------------
// Legacy single-array resampler (resample.c:16-32): sequential order, phase
// recurrence from `filter_offset`.
void resampleRR(int buf_size, int coeff_size, int interpolation, int decimation, int filter_offset, float* coeffs,
                float* in_buf, float* out_buf)
try {
    if (buf_size <= 0) return;
    if (interpolation < 1 || decimation <= interpolation || filter_offset < 0 || filter_offset >= interpolation) {
        // outside these the reference's own recurrence (resample.c:16-32) indexes before its arrays
        dropin_fail(SDRHIP_ERR_ARG, "resampleRR: need 1 <= interpolation < decimation and 0 <= filter_offset < interpolation");
    }
    Lease l;
    Ctx& c = *l.c;
    ResampTable t;
    // walk the recurrence from filter_offset until it repeats (at most `interpolation` phases)
    std::vector<int> fo, pre;
    int off = filter_offset, ng = 0, acc = 0;
    do {
        fo.push_back(off);
        pre.push_back(acc);
        acc += (decimation - off - 1) / interpolation + 1;
        off = interpolation - 1 - (decimation - off - 1) % interpolation;
        ng++;
    } while (off != filter_offset && ng < interpolation);
    std::vector<int> ext;
    if (ng <= 64) {
        for (int q = 0; q < ng; q++) { t.fo[q] = fo[q]; t.pre[q] = pre[q]; }
    } else {
        ext.assign((size_t)2 * ng + 1, 0);
        for (int q = 0; q < ng; q++) { ext[q] = pre[q]; ext[ng + 1 + q] = fo[q]; }
        ext[ng] = acc;
    }
    t.ngroups = ng;
    t.group0 = 0;
    t.pos0 = 0;
    t.period = acc;
    t.row_stride = 0;
    t.nloop = 0;
    t.ntaps_plain = coeff_size;
    t.force_seq = 1;
    int64_t last = buf_size - 1;
    int maxlen = (coeff_size + interpolation - 1) / interpolation;
    size_t nin = (size_t)((last / ng) * (int64_t)t.period + pre[last % ng] + maxlen);
    const bool direct = fits_direct(nin * 4, (size_t)buf_size * 4);
    const float* d_taps = taps_on_device(c, c.taps, c.taps_now, coeffs, (size_t)coeff_size * 4);
    if (!ext.empty()) t.ext = (const int*)taps_on_device(c, c.taps2, c.taps2_now, ext.data(), ext.size() * 4);
    const float* d_in = (const float*)stage_in(c, direct, in_buf, nin * 4);
    float* d_out = (float*)stage_out(c, direct, (size_t)buf_size * 4);
    Geom g = flat_geom(buf_size, decimation, 0);
    g.I = interpolation;
    launch_resample_real(c.stream, g, 1, t, nullptr, d_taps, d_in, d_out);
    finish(c, direct, out_buf, (size_t)buf_size * 4);
} catch (const DropinAbort&) {}

int resample2RR(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(false, 1, n, nc, sg, ng, inc, c, in, out); }
int resampleSSERR(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(false, 4, n, nc, sg, ng, inc, c, in, out); }
int resampleAVXRR(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(false, 8, n, nc, sg, ng, inc, c, in, out); }
int resample2RC(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(true, CO_SEQ, n, nc, sg, ng, inc, c, in, out); }
int resampleSSERC(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(true, CO_X2, n, nc, sg, ng, inc, c, in, out); }
int resampleAVXRC(int n, int nc, int sg, int ng, int* inc, float** c, float* in, float* out) { return resample_groups(true, CO_X4, n, nc, sg, ng, inc, c, in, out); }

// ---- fmDemod (new FFI seam) -------------------------------------------
void fmDemodF(int num, float last_re, float last_im, const float* in_iq, float* out)
{
    if (num <= 0) return;
    elementwise(in_iq, (size_t)num * 8, out, (size_t)num * 4, [&](hipStream_t s, const void* di, void* dout) {
        launch_fm_demod_fast(s, (const float*)di, (float*)dout, num, false, last_re, last_im);
    });
}

}  // extern "C"
