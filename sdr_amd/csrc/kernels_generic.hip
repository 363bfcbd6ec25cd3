// kernels_generic.hip -- the correctness backbone: one thread per output, any tap
// count / factor / order / seam block.  The specialised, LDS-tiled kernels for the
// hot configurations live in kernels_fast.hip; everything they do not cover lands
// here.
//
// Arithmetic contract: IEEE binary32, separate multiply and add
// (this file MUST be compiled with -ffp-contract=off), denormals preserved,
// partial sums strided by the SIMD width of the reference variant being
// reproduced, reduced by the reference's horizontal-add tree:
//   real  4 lanes: (a0+a1)+(a2+a3)                      common.h:12-16
//   real  8 lanes: ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7))  common.h:18-29
//   cplx  2 lanes: l0+l1            4 lanes: (l0+l1)+(l2+l3)      common.h:77-90
//   "RC2": 2*CL complex partials, q_k = p_k + p_{k+CL}, then the CL-lane tree  common.h:108-155
// "Cross" outputs (window straddles a seam) use the sequential order of the
// reference's pure-Haskell kernels, FilterInternal.hs:397-423.
#include "kernels.hpp"

namespace sdrhip {

// One / Cross classification of output m (Filter.hs:536-727): virtual window [v, v + Lp), v = m*D, in the upsampled
// index.  One when the window fits the buffer it starts in; else Cross -- unless the Pipe never crosses over at that
// boundary (seam_has_crossover, kernels.hpp), in which case the output is the first One of the next buffer.
template <int L>
__device__ __forceinline__ float tree_r(const float* a)
{
    if constexpr (L == 1) return a[0];
    else if constexpr (L == 4) return (a[0] + a[1]) + (a[2] + a[3]);
    else return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// ---------------------------------------------------------------------------
// real data
// ---------------------------------------------------------------------------
template <int L, bool SYM>
__global__ void __launch_bounds__(256) k_fir_real(Geom g, const float* __restrict__ taps, int ntaps,
                                                   const float* __restrict__ xtaps,
                                                   const float* __restrict__ in, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.count) return;
    int64_t m = g.k_begin + i;
    const float* x = in + (m * g.D - g.in_base);
    float r;
    if (is_cross(g, m)) {
        r = 0.0f;
        for (int j = 0; j < g.Lp; j++) r = r + x[j] * xtaps[j];
    } else {
        float acc[L];
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = 0.0f;
        if constexpr (SYM) {
            const float* e = x + 2 * ntaps - 1;
            for (int j = 0; j < ntaps; j += L) {
#pragma unroll
                for (int l = 0; l < L; l++) acc[l] = acc[l] + taps[j + l] * (x[j + l] + e[-(j + l)]);
            }
        } else {
            for (int j = 0; j < ntaps; j += L) {
#pragma unroll
                for (int l = 0; l < L; l++) acc[l] = acc[l] + taps[j + l] * x[j + l];
            }
        }
        r = tree_r<L>(acc);
    }
    out[i] = r;
}

void launch_fir_real(hipStream_t s, const Geom& g, int lanes, bool sym, const float* d_taps, int ntaps,
                     const float* d_cross_taps, const float* d_in, float* d_out)
{
    if (g.count <= 0) return;
    dim3 grid((g.count + 255) / 256), block(256);
#define GO(L, S) hipLaunchKernelGGL((k_fir_real<L, S>), grid, block, 0, s, g, d_taps, ntaps, d_cross_taps, d_in, d_out)
    if (lanes == 1) GO(1, false);
    else if (lanes == 4) { if (sym) GO(4, true); else GO(4, false); }
    else { if (sym) GO(8, true); else GO(8, false); }
#undef GO
}

// ---------------------------------------------------------------------------
// complex data, real taps.  Loader abstracts f32-IQ vs u8-IQ input.
// ---------------------------------------------------------------------------
struct LoadF32 {
    const float* p;
    __device__ __forceinline__ float2 operator()(int64_t idx) const { return *reinterpret_cast<const float2*>(p + 2 * idx); }
};
struct LoadU8 {
    const uint8_t* p;
    // convert.c:15-20: ((float)u - 128) * (1/128); exact, so any evaluation order gives the same bits
    __device__ __forceinline__ float2 operator()(int64_t idx) const
    {
        uchar2 u = *reinterpret_cast<const uchar2*>(p + 2 * idx);
        return make_float2(((float)u.x - 128.0f) * (1.0f / 128.0f), ((float)u.y - 128.0f) * (1.0f / 128.0f));
    }
};

template <int CL>
__device__ __forceinline__ float2 tree_c(const float2* a)
{
    if constexpr (CL == 1) return a[0];
    else if constexpr (CL == 2) return make_float2(a[0].x + a[1].x, a[0].y + a[1].y);
    else return make_float2((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y));
}

// ORDER: see ComplexOrder.  taps layout: CO_L2/CO_L4 interleaved (re-tap, im-tap) per
// complex tap -- the reference's "duplicated" array, honoured even when the two
// copies differ; other orders plain.
template <int ORDER, bool SYM, class Loader>
__device__ __forceinline__ float2 dot_cplx(const float* __restrict__ taps, int ntaps, Loader ld, int64_t x0)
{
    if constexpr (ORDER == CO_SEQ) {
        float re = 0.0f, im = 0.0f;
        for (int j = 0; j < ntaps; j++) {
            float2 v = ld(x0 + j);
            re = re + v.x * taps[j];
            im = im + v.y * taps[j];
        }
        return make_float2(re, im);
    } else if constexpr (ORDER == CO_L2 || ORDER == CO_L4) {
        constexpr int CL = (ORDER == CO_L2) ? 2 : 4;
        float2 acc[CL];
#pragma unroll
        for (int l = 0; l < CL; l++) acc[l] = make_float2(0.0f, 0.0f);
        int P = ntaps / 2;
        for (int j = 0; j < P; j += CL) {
#pragma unroll
            for (int l = 0; l < CL; l++) {
                float2 v = ld(x0 + j + l);
                acc[l].x = acc[l].x + taps[2 * (j + l)] * v.x;
                acc[l].y = acc[l].y + taps[2 * (j + l) + 1] * v.y;
            }
        }
        return tree_c<CL>(acc);
    } else {
        constexpr int CL = (ORDER == CO_X2) ? 2 : 4;
        constexpr int M = 2 * CL;
        float2 p[M];
#pragma unroll
        for (int l = 0; l < M; l++) p[l] = make_float2(0.0f, 0.0f);
        for (int j = 0; j < ntaps; j += M) {
#pragma unroll
            for (int l = 0; l < M; l++) {
                float2 v = ld(x0 + j + l);
                if constexpr (SYM) {
                    float2 w = ld(x0 + 2 * ntaps - 1 - (j + l));
                    v.x = v.x + w.x;
                    v.y = v.y + w.y;
                }
                p[l].x = p[l].x + taps[j + l] * v.x;
                p[l].y = p[l].y + taps[j + l] * v.y;
            }
        }
        float2 q[CL];
#pragma unroll
        for (int l = 0; l < CL; l++) q[l] = make_float2(p[l].x + p[l + CL].x, p[l].y + p[l + CL].y);
        return tree_c<CL>(q);
    }
}

template <class Loader>
__device__ __forceinline__ float2 dot_cplx_seq(const float* __restrict__ taps, int n, Loader ld, int64_t x0)
{
    float re = 0.0f, im = 0.0f;
    for (int j = 0; j < n; j++) {
        float2 v = ld(x0 + j);
        re = re + v.x * taps[j];
        im = im + v.y * taps[j];
    }
    return make_float2(re, im);
}

template <int ORDER, bool SYM, class Loader>
__global__ void __launch_bounds__(256) k_fir_cplx(Geom g, const float* __restrict__ taps, int ntaps,
                                                   const float* __restrict__ xtaps, Loader ld,
                                                   float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.count) return;
    int64_t m = g.k_begin + i;
    int64_t x0 = m * g.D - g.in_base;
    float2 r;
    if (is_cross(g, m)) r = dot_cplx_seq(xtaps, g.Lp, ld, x0);
    else r = dot_cplx<ORDER, SYM>(taps, ntaps, ld, x0);
    *reinterpret_cast<float2*>(out + 2 * (int64_t)i) = r;
}

template <class Loader>
static void launch_fir_cplx_t(hipStream_t s, const Geom& g, ComplexOrder order, bool sym, const float* d_taps,
                              int ntaps, const float* d_cross_taps, Loader ld, float* d_out)
{
    if (g.count <= 0) return;
    dim3 grid((g.count + 255) / 256), block(256);
#define GO(O, S) hipLaunchKernelGGL((k_fir_cplx<O, S, Loader>), grid, block, 0, s, g, d_taps, ntaps, d_cross_taps, ld, d_out)
    switch (order) {
        case CO_SEQ: GO(CO_SEQ, false); break;
        case CO_L2: GO(CO_L2, false); break;
        case CO_L4: GO(CO_L4, false); break;
        case CO_X2: if (sym) GO(CO_X2, true); else GO(CO_X2, false); break;
        case CO_X4: if (sym) GO(CO_X4, true); else GO(CO_X4, false); break;
    }
#undef GO
}

void launch_fir_cplx(hipStream_t s, const Geom& g, ComplexOrder order, bool sym, const float* d_taps, int ntaps,
                     const float* d_cross_taps, const float* d_in, float* d_out)
{
    launch_fir_cplx_t(s, g, order, sym, d_taps, ntaps, d_cross_taps, LoadF32{d_in}, d_out);
}

void launch_fir_cplx_u8(hipStream_t s, const Geom& g, ComplexOrder order, const float* d_taps, int ntaps,
                        const float* d_cross_taps, const uint8_t* d_in, float* d_out)
{
    launch_fir_cplx_t(s, g, order, false, d_taps, ntaps, d_cross_taps, LoadU8{d_in}, d_out);
}

// ---------------------------------------------------------------------------
// polyphase resampler (resample.c:34-142; phase closed form Filter.hs:613-641)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void resamp_locate(const ResampTable& t, int i, int& group, int64_t& pos)
{
    int cyc = i / t.ngroups;
    int r = i - cyc * t.ngroups;
    group = t.group0 + r;
    if (group >= t.ngroups) group -= t.ngroups;
    int pre;
    if (t.ext != nullptr) {
        // prefix of the increments starting at group0, from the un-rotated prefix sums
        const int a = t.group0 + r, base = t.ext[t.group0];
        pre = a <= t.ngroups ? t.ext[a] - base : t.period - base + t.ext[a - t.ngroups];
    } else {
        pre = t.pre[r];
    }
    pos = t.pos0 + (int64_t)cyc * t.period + pre;
}
__device__ __forceinline__ int resamp_filter_offset(const ResampTable& t, int group)
{
    return t.ext != nullptr ? t.ext[t.ngroups + 1 + group] : t.fo[group];
}

template <int L>
__global__ void __launch_bounds__(256) k_resample_real(Geom g, ResampTable t, const float* __restrict__ groups,
                                                        const float* __restrict__ plain,
                                                        const float* __restrict__ in, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.count) return;
    int group;
    int64_t pos;
    resamp_locate(t, i, group, pos);
    const float* x = in + pos;
    int64_t m = g.k_begin + i;
    float r;
    if (t.force_seq || is_cross(g, m)) {
        // FilterInternal.hs:410-423 / resample.c:16-32: stride I (drop filterOffset coeffs), sequential
        int fo = resamp_filter_offset(t, group);
        r = 0.0f;
        for (int l = 0, j = fo; j < t.ntaps_plain; l++, j += g.I) r = r + x[l] * plain[j];
    } else {
        const float* c = groups + (size_t)group * t.row_stride;
        float acc[L];
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = 0.0f;
        for (int j = 0; j < t.nloop; j += L) {
#pragma unroll
            for (int l = 0; l < L; l++) acc[l] = acc[l] + c[j + l] * x[j + l];
        }
        r = tree_r<L>(acc);
    }
    out[i] = r;
}

void launch_resample_real(hipStream_t s, const Geom& g, int lanes, const ResampTable& t, const float* d_groups,
                          const float* d_plain_taps, const float* d_in, float* d_out)
{
    if (g.count <= 0) return;
    dim3 grid((g.count + 255) / 256), block(256);
    if (lanes == 1) hipLaunchKernelGGL((k_resample_real<1>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
    else if (lanes == 4) hipLaunchKernelGGL((k_resample_real<4>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
    else hipLaunchKernelGGL((k_resample_real<8>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
}

template <int ORDER>
__global__ void __launch_bounds__(256) k_resample_cplx(Geom g, ResampTable t, const float* __restrict__ groups,
                                                        const float* __restrict__ plain,
                                                        const float* __restrict__ in, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.count) return;
    int group;
    int64_t pos;
    resamp_locate(t, i, group, pos);
    int64_t m = g.k_begin + i;
    LoadF32 ld{in};
    float2 r;
    if (t.force_seq || is_cross(g, m)) {
        int fo = resamp_filter_offset(t, group);
        float re = 0.0f, im = 0.0f;
        for (int l = 0, j = fo; j < t.ntaps_plain; l++, j += g.I) {
            float2 v = ld(pos + l);
            re = re + v.x * plain[j];
            im = im + v.y * plain[j];
        }
        r = make_float2(re, im);
    } else {
        r = dot_cplx<ORDER, false>(groups + (size_t)group * t.row_stride, t.nloop, ld, pos);
    }
    *reinterpret_cast<float2*>(out + 2 * (int64_t)i) = r;
}

void launch_resample_cplx(hipStream_t s, const Geom& g, ComplexOrder order, const ResampTable& t,
                          const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out)
{
    if (g.count <= 0) return;
    dim3 grid((g.count + 255) / 256), block(256);
    if (order == CO_SEQ) hipLaunchKernelGGL((k_resample_cplx<CO_SEQ>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
    else if (order == CO_X2) hipLaunchKernelGGL((k_resample_cplx<CO_X2>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
    else hipLaunchKernelGGL((k_resample_cplx<CO_X4>), grid, block, 0, s, g, t, d_groups, d_plain_taps, d_in, d_out);
}

// ---------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------
// convert.c:15-50.  The kernel is write-dominated (2 B in, 8 B out per sample), so the STORES are
// what must coalesce: each lane converts one dword (4 bytes) into one float4, so a wave's store
// instruction writes 1 KiB of consecutive floats; four such dword/float4 pairs per thread per trip,
// a whole workgroup-stride apart, keep enough loads in flight.
__global__ void __launch_bounds__(256) k_convert_u8(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n)
{
    const int64_t nvec = n >> 2;                                    // dwords in / float4s out
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint32_t* in4 = reinterpret_cast<const uint32_t*>(in);
    float4* out4 = reinterpret_cast<float4*>(out);
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v + 3 * stride < nvec; v += 4 * stride) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = in4[v + k * stride];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float4 f;
            f.x = ((float)(w[k] & 0xff) - 128.0f) * (1.0f / 128.0f);
            f.y = ((float)((w[k] >> 8) & 0xff) - 128.0f) * (1.0f / 128.0f);
            f.z = ((float)((w[k] >> 16) & 0xff) - 128.0f) * (1.0f / 128.0f);
            f.w = ((float)(w[k] >> 24) - 128.0f) * (1.0f / 128.0f);
            out4[v + k * stride] = f;
        }
    }
    for (; v < nvec; v += stride) {
        const uint32_t w = in4[v];
        float4 f;
        f.x = ((float)(w & 0xff) - 128.0f) * (1.0f / 128.0f);
        f.y = ((float)((w >> 8) & 0xff) - 128.0f) * (1.0f / 128.0f);
        f.z = ((float)((w >> 16) & 0xff) - 128.0f) * (1.0f / 128.0f);
        f.w = ((float)(w >> 24) - 128.0f) * (1.0f / 128.0f);
        out4[v] = f;
    }
    // tail (< 4 bytes)
    if (blockIdx.x == 0) {
        int64_t t = (nvec << 2) + threadIdx.x;
        if (t < n) out[t] = ((float)in[t] - 128.0f) * (1.0f / 128.0f);
    }
}

__global__ void __launch_bounds__(256) k_convert_u8_unaligned(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        out[t] = ((float)in[t] - 128.0f) * (1.0f / 128.0f);
}

static inline int grid_for(int64_t work_items, int cap = 256 * 8)
{
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

void launch_convert_u8(hipStream_t s, const uint8_t* d_in, float* d_out, int64_t n)
{
    if (n <= 0) return;
    bool aligned = ((reinterpret_cast<uintptr_t>(d_in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0);
    if (aligned) hipLaunchKernelGGL(k_convert_u8, dim3(grid_for((n >> 4) + 1, 256 * 16)), dim3(256), 0, s, d_in, d_out, n);
    else hipLaunchKernelGGL(k_convert_u8_unaligned, dim3(grid_for(n)), dim3(256), 0, s, d_in, d_out, n);
}

// convert.c:52-85
__global__ void __launch_bounds__(256) k_convert_i16(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        out[t] = (float)in[t] * (1.0f / 2048.0f);
}
void launch_convert_i16(hipStream_t s, const int16_t* d_in, float* d_out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_convert_i16, dim3(grid_for(n)), dim3(256), 0, s, d_in, d_out, n);
}

// convert.c:87-101.  (int16_t)val truncates toward zero; values are in [0,4096) for
// inputs in [-1,1) so the saturating branches never fire for in-range data, but
// out-of-range floats wrap through int16 exactly like the C cast on x86
// (cvttss2si -> low 16 bits).
__global__ void __launch_bounds__(256) k_convert_tx(const float* __restrict__ in, int16_t* __restrict__ out, int64_t n)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        float val = in[t];
        val = val + 1.0f;
        val = val * 2048.0f;
        int32_t wide;
        // x86 cvttss2si returns INT_MIN for NaN / out-of-int32-range
        if (!(val > -2147483904.0f && val < 2147483648.0f)) wide = (int32_t)0x80000000;
        else wide = (int32_t)val;
        int16_t res = (int16_t)(uint16_t)(uint32_t)wide;
        res = (int16_t)(res - 2048);
        if (res > 2047) res = 2047;
        if (res < -2048) res = -2048;
        out[t] = res;
    }
}
void launch_convert_f32_to_i16_bladerf(hipStream_t s, const float* d_in, int16_t* d_out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_convert_tx, dim3(grid_for(n)), dim3(256), 0, s, d_in, d_out, n);
}

// scale.c:15-36
__global__ void __launch_bounds__(256) k_scale(float factor, const float* __restrict__ in, float* __restrict__ out, int64_t n)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = in[t] * factor;
}
void launch_scale(hipStream_t s, float factor, const float* d_in, float* d_out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, s, factor, d_in, d_out, n);
}

// ---------------------------------------------------------------------------
// fmDemod.  Demod.hs:21-46 with GHC base's Data.Complex / RealFloat atan2
//  and the host libm's atanf.  The image's glibc 2.35 atanf
// is the fdlibm f32 algorithm; device_atanf below evaluates the same operations in
// the same order in f32 (no contraction) and matches it bit-for-bit on all 2^32
// inputs (tests/test_oracle_demod.py sweeps the C twin of this function).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float device_atanf(float x)
{
    const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    uint32_t hx = __float_as_uint(x);
    uint32_t ix = hx & 0x7fffffffu;
    bool neg = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {
        if (ix > 0x7f800000u) return x + x;
        float r = hi[3] + lo[3];
        return neg ? -r : r;
    }
    int id;
    if (ix < 0x3ee00000u) {
        if (ix < 0x31000000u) return x;
        id = -1;
    } else {
        x = __uint_as_float(ix);
        if (ix < 0x3f980000u) {
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    float hv = id == 0 ? hi[0] : id == 1 ? hi[1] : id == 2 ? hi[2] : hi[3];
    float lv = id == 0 ? lo[0] : id == 1 ? lo[1] : id == 2 ? lo[2] : lo[3];
    z = hv - ((x * (s1 + s2) - lv) - x);
    return neg ? -z : z;
}

__device__ __forceinline__ bool is_neg_zero(float v) { return __float_as_uint(v) == 0x80000000u; }

// GHC RealFloat default atan2 (recursion unrolled: the recursive call always has y >= +0)
__device__ __forceinline__ float ghc_atan2(float y, float x)
{
    const float pi = 3.14159274101257324f;
    if (x > 0.0f) return device_atanf(y / x);
    if (x == 0.0f && y > 0.0f) return pi / 2.0f;
    if (x < 0.0f && y > 0.0f) return pi + device_atanf(y / x);
    if ((x <= 0.0f && y < 0.0f) || (x < 0.0f && is_neg_zero(y)) || (is_neg_zero(x) && is_neg_zero(y))) {
        float yn = -y;  // > 0 or +0
        float r;
        if (x == 0.0f && yn > 0.0f) r = pi / 2.0f;
        else if (x < 0.0f && yn > 0.0f) r = pi + device_atanf(yn / x);
        else if (yn == 0.0f && (x < 0.0f || is_neg_zero(x))) r = pi;
        else if (x == 0.0f && yn == 0.0f) r = yn;
        else r = x + yn;
        return -r;
    }
    if (y == 0.0f && (x < 0.0f || is_neg_zero(x))) return pi;
    if (x == 0.0f && y == 0.0f) return y;
    return x + y;
}

__device__ __forceinline__ float fm_phase(float2 cur, float2 prev)
{
    // sample * conjugate last; conjugate (c:+d) = c:+(-d); (a:+b)*(c:+e) = (a*c-b*e):+(a*e+b*c)
    float nd = -prev.y;
    float re = cur.x * prev.x - cur.y * nd;
    float im = cur.x * nd + cur.y * prev.x;
    if (re == 0.0f && im == 0.0f) return 0.0f;
    return ghc_atan2(im, re);
}

__global__ void __launch_bounds__(256) k_fm_demod(const float* __restrict__ in, float* __restrict__ out, int64_t count,
                                                   int has_prev, float last_re, float last_im)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        float2 cur = *reinterpret_cast<const float2*>(in + 2 * i);
        float2 prev;
        if (i > 0 || has_prev) prev = *reinterpret_cast<const float2*>(in + 2 * (i - 1));
        else prev = make_float2(last_re, last_im);
        out[i] = fm_phase(cur, prev);
    }
}

void launch_fm_demod(hipStream_t s, const float* d_in_iq, float* d_out, int64_t count, bool has_prev,
                     float last_re, float last_im)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_fm_demod, dim3(grid_for(count)), dim3(256), 0, s, d_in_iq, d_out, count, has_prev ? 1 : 0,
                       last_re, last_im);
}

}  // namespace sdrhip
