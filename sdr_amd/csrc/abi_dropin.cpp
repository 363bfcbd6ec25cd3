// abi_dropin.cpp -- layer (1) of include/sdr_hip.h: the reference's native entry
// points re-implemented on the GPU.  HOST pointers in/out, synchronous, no state
// retained past the call (the reference's FFI contract: VS.unsafeWith pointers are
// only valid during the call, FilterInternal.hs:66-71).
//
// Each symbol keeps the summation order of the x86 variant it replaces, so a
// Haskell program that resolves its `foreign import ccall` against this library
// produces the same bits whichever variant SDR.CPUID.featureSelect picks.
//
// A process-wide scratch context (device in/out/tap buffers + one HIP stream)
// is guarded by a mutex: the reference calls these from a single pipeline thread,
// but nothing stops another caller.
#include <string.h>

#include <mutex>

#include "common.hpp"
#include "kernels.hpp"

using namespace sdrhip;

namespace {

struct Scratch {
    std::mutex mu;
    hipStream_t stream = nullptr;
    DevBuf in, out, taps, taps2;
    hipStream_t s()
    {
        if (!stream) SDRHIP_DIE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return stream;
    }
};
Scratch& scratch()
{
    static Scratch* sc = new Scratch();  // intentionally leaked: no destructor-order games at exit
    return *sc;
}

void die_if(int rc, const char* what)
{
    if (rc != SDRHIP_OK) {
        fprintf(stderr, "libsdr_hip: %s: %s\n", what, sdrhip_last_error());
        abort();
    }
}

void up(Scratch& sc, DevBuf& b, const void* h, size_t bytes)
{
    die_if(b.ensure(bytes), "device buffer");
    if (bytes) SDRHIP_DIE_HIP(hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, sc.s()));
}
void down(Scratch& sc, void* h, const DevBuf& b, size_t bytes)
{
    if (bytes) SDRHIP_DIE_HIP(hipMemcpyAsync(h, b.p, bytes, hipMemcpyDeviceToHost, sc.s()));
    SDRHIP_DIE_HIP(hipStreamSynchronize(sc.s()));
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
{
    if (num <= 0) return;
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
    int span = sym ? 2 * numCoeffs : numCoeffs;
    size_t nin = (size_t)(num - 1) * factor + span;
    up(sc, sc.taps, coeffs, (size_t)numCoeffs * 4);
    up(sc, sc.in, in, nin * 4);
    die_if(sc.out.ensure((size_t)num * 4), "device buffer");
    launch_fir_real(sc.s(), flat_geom(num, factor, span), lanes, sym, (const float*)sc.taps.p, numCoeffs, nullptr,
                    (const float*)sc.in.p, (float*)sc.out.p);
    SDRHIP_DIE_HIP(hipGetLastError());
    down(sc, out, sc.out, (size_t)num * 4);
}

// real taps, complex data.  numCoeffs is the length of the array as passed.
void fir_cplx(ComplexOrder order, bool sym, int num, int factor, int numCoeffs, float* coeffs, float* in, float* out)
{
    if (num <= 0) return;
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
    int P = (order == CO_L2 || order == CO_L4) ? numCoeffs / 2 : numCoeffs;  // complex taps walked
    int span = sym ? 2 * P : P;
    size_t nin = (size_t)(num - 1) * factor + span;
    up(sc, sc.taps, coeffs, (size_t)numCoeffs * 4);
    up(sc, sc.in, in, nin * 8);
    die_if(sc.out.ensure((size_t)num * 8), "device buffer");
    Geom g = flat_geom(num, factor, span);
    bool done = false;
    std::vector<float> plain;  // must outlive the stream sync in down()
    if (order == CO_L4 && !sym) {
        // the LDS-tiled kernel wants plain taps; only valid when the array really is a duplicate
        bool dup = true;
        for (int i = 0; i < P && dup; i++) dup = memcmp(&coeffs[2 * i], &coeffs[2 * i + 1], 4) == 0;
        if (dup) {
            plain.resize(P);
            for (int i = 0; i < P; i++) plain[i] = coeffs[2 * i];
            up(sc, sc.taps2, plain.data(), (size_t)P * 4);
            done = launch_decimate_c4_fast(sc.s(), g, (const float*)sc.taps2.p, P, nullptr, sc.in.p, false,
                                           (float*)sc.out.p);
        }
    }
    if (!done)
        launch_fir_cplx(sc.s(), g, order, sym, (const float*)sc.taps.p, numCoeffs, nullptr, (const float*)sc.in.p,
                        (float*)sc.out.p);
    SDRHIP_DIE_HIP(hipGetLastError());
    down(sc, out, sc.out, (size_t)num * 8);
}

// polyphase resamplers (resample.c:34-142)
int resample_groups(bool cplx, int lanes_or_order, int buf_size, int num_coeffs, int starting_group, int num_groups,
                    int* increments, float** coeffs, float* in, float* out)
{
    if (num_groups <= 0 || num_groups > 64) {
        fprintf(stderr, "libsdr_hip: resample: num_groups %d unsupported (1..64)\n", num_groups);
        abort();
    }
    int end_group = (int)(((int64_t)starting_group + (buf_size > 0 ? buf_size : 0)) % num_groups);
    if (buf_size <= 0) return starting_group;
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
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
    int acc = 0;
    for (int q = 0; q < num_groups; q++) {
        t.pre[q] = acc;
        acc += increments[(starting_group + q) % num_groups];
        t.fo[q] = 0;
    }
    t.period = acc;
    t.row_stride = nloop;
    t.nloop = nloop;
    t.ntaps_plain = 0;
    t.force_seq = 0;
    int64_t last = buf_size - 1;
    size_t nin = (size_t)((last / num_groups) * (int64_t)t.period + t.pre[last % num_groups] + nloop);
    size_t esz = cplx ? 8 : 4;
    up(sc, sc.taps, table.data(), table.size() * 4);
    up(sc, sc.in, in, nin * esz);
    die_if(sc.out.ensure((size_t)buf_size * esz), "device buffer");
    Geom g = flat_geom(buf_size, 1, 0);
    if (cplx) launch_resample_cplx(sc.s(), g, co, t, (const float*)sc.taps.p, nullptr, (const float*)sc.in.p, (float*)sc.out.p);
    else launch_resample_real(sc.s(), g, simd, t, (const float*)sc.taps.p, nullptr, (const float*)sc.in.p, (float*)sc.out.p);
    SDRHIP_DIE_HIP(hipGetLastError());
    down(sc, out, sc.out, (size_t)buf_size * esz);
    return end_group;
}

template <class Fn>
void elementwise(const void* in, size_t in_bytes, void* out, size_t out_bytes, Fn launch)
{
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
    up(sc, sc.in, in, in_bytes);
    die_if(sc.out.ensure(out_bytes), "device buffer");
    launch(sc.s(), sc.in.p, sc.out.p);
    SDRHIP_DIE_HIP(hipGetLastError());
    down(sc, out, sc.out, out_bytes);
}

}  // namespace

extern "C" {

// ---- convert.c ------------------------------------------------------------------
// The three u8 variants are bit-identical in the reference (exact results); unlike
// convertCSSE/AVX (convert.c:27,42) this never reads past in[num).
void convertC(int num, uint8_t* in, float* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num, out, (size_t)num * 4,
                [&](hipStream_t s, void* di, void* dout) { launch_convert_u8(s, (const uint8_t*)di, (float*)dout, num); });
}
void convertCSSE(int num, uint8_t* in, float* out) { convertC(num, in, out); }
void convertCAVX(int num, uint8_t* in, float* out) { convertC(num, in, out); }

void convertCBladeRF(int num, int16_t* in, float* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num * 2, out, (size_t)num * 4,
                [&](hipStream_t s, void* di, void* dout) { launch_convert_i16(s, (const int16_t*)di, (float*)dout, num); });
}
void convertCSSEBladeRF(int num, int16_t* in, float* out) { convertCBladeRF(num, in, out); }
void convertCAVXBladeRF(int num, int16_t* in, float* out) { convertCBladeRF(num, in, out); }
void convertBladeRFTransmit(int num, float* in, int16_t* out)
{
    if (num <= 0) return;
    elementwise(in, (size_t)num * 4, out, (size_t)num * 2, [&](hipStream_t s, void* di, void* dout) {
        launch_convert_f32_to_i16_bladerf(s, (const float*)di, (int16_t*)dout, num);
    });
}

// ---- scale.c ----------------------------------------------------------------------
void scale(int num, float factor, float* in_buf, float* out_buf)
{
    if (num <= 0) return;
    elementwise(in_buf, (size_t)num * 4, out_buf, (size_t)num * 4,
                [&](hipStream_t s, void* di, void* dout) { launch_scale(s, factor, (const float*)di, (float*)dout, num); });
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
{
    if (num <= 0) {
        *finalSample = lastSample;
        *finalOutput = lastOutput;
        return;
    }
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
    up(sc, sc.in, inBuf, (size_t)num * 4);
    die_if(sc.out.ensure((size_t)num * 4 + 16), "device buffer");
    die_if(sc.taps2.ensure(dc_blocker_workspace_bytes(num)), "device buffer");
    float* dout = (float*)sc.out.p;   // {finalSample, finalOutput, -, -} then the block, 16-byte aligned
    launch_dc_blocker(sc.s(), num, lastSample, lastOutput, (const float*)sc.in.p, dout + 4, dout, sc.taps2.p, 0);
    SDRHIP_DIE_HIP(hipGetLastError());
    float fin[2];
    SDRHIP_DIE_HIP(hipMemcpyAsync(fin, dout, 8, hipMemcpyDeviceToHost, sc.s()));
    SDRHIP_DIE_HIP(hipMemcpyAsync(outBuf, dout + 4, (size_t)num * 4, hipMemcpyDeviceToHost, sc.s()));
    SDRHIP_DIE_HIP(hipStreamSynchronize(sc.s()));
    *finalSample = fin[0];
    *finalOutput = fin[1];
}

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
{
    if (buf_size <= 0) return;
    if (interpolation < 1 || interpolation > 64 || decimation <= interpolation) {
        fprintf(stderr, "libsdr_hip: resampleRR: need 1 <= interpolation <= 64 < decimation\n");
        abort();
    }
    Scratch& sc = scratch();
    std::lock_guard<std::mutex> lk(sc.mu);
    ResampTable t;
    // walk the recurrence from filter_offset until it repeats
    int off = filter_offset, ng = 0, acc = 0;
    do {
        t.fo[ng] = off;
        t.pre[ng] = acc;
        acc += (decimation - off - 1) / interpolation + 1;
        off = interpolation - 1 - (decimation - off - 1) % interpolation;
        ng++;
    } while (off != filter_offset && ng < 64);
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
    size_t nin = (size_t)((last / ng) * (int64_t)t.period + t.pre[last % ng] + maxlen);
    up(sc, sc.taps, coeffs, (size_t)coeff_size * 4);
    up(sc, sc.in, in_buf, nin * 4);
    die_if(sc.out.ensure((size_t)buf_size * 4), "device buffer");
    Geom g = flat_geom(buf_size, decimation, 0);
    g.I = interpolation;
    launch_resample_real(sc.s(), g, 1, t, nullptr, (const float*)sc.taps.p, (const float*)sc.in.p, (float*)sc.out.p);
    SDRHIP_DIE_HIP(hipGetLastError());
    down(sc, out_buf, sc.out, (size_t)buf_size * 4);
}

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
    elementwise(in_iq, (size_t)num * 8, out, (size_t)num * 4, [&](hipStream_t s, void* di, void* dout) {
        launch_fm_demod_fast(s, (const float*)di, (float*)dout, num, false, last_re, last_im);
    });
}

}  // extern "C"
