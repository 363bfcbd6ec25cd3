// tools/k2lab/sys_lab.hip -- development bench (round 4): the decimate-by-8, 128-tap complex decimator as a REGISTER-RESIDENT
// SYSTOLIC walk whose multiplies run on the matrix pipe.  Not part of the product; every variant is checked bit for bit
// against the production kernel (decimate_tile.hpp).
//
// Idea.  out[o] = (L0+L1)+(L2+L3), L_k = sum_{j = k (mod 4)} h[j] x[8o+j], every L_k from +0 in increasing j with separate
// multiply and add (decimate.c:105-113, common.h:58-90).  Write j = 8b + r (b = 0..15, r = 0..7):
//   * lane l of a wave keeps the 32 samples 32l .. 32l+31 of its strip in registers (sample 8c + r, c = 0..3) and never
//     moves them;
//   * the 32 partial sums of the output group q = (outputs 4q .. 4q+3) x (4 partials) x (re, im) TRAVEL: they start in lane
//     q, and after every stage of 32 taps move one lane up as the DPP operand of the stage's first add
//     (v_add_f32_dpp wave_shr:1).  In stage t lane l works on group l - t: output 4(l-t) + i meets sample 8c + r of lane l
//     under tap b = 4t + c - i.  Five stages (t = 0..4) complete a group; lanes 0..3 of a wave only warm the pipe up
//     (a wave turns 2048 samples into 240 outputs, consecutive waves overlap by 128 samples);
//   * the PRODUCTS come from v_mfma_f32_4x4x1_16b_f32 with C = 0: D[i][lane] = fma(A[quad lane i], B[lane], +0) =
//     round(A * B), the bits of v_mul_f32 (tools/mfma_mul_bench.hip).  B = the lane's own sample component, A = the four
//     taps h[8(m - i) + r], i = 0..3 (m = 4t + c), which sit in the four lanes of every quad: one instruction = the lane's
//     sample against the taps of its group's four outputs.  The VALU is left with the additions only -- half of today's
//     VALU work -- in exactly the reference's order (per partial: b ascending, r = k before r = k + 4).
//
//   F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isdr_amd/csrc"
//   hipcc $F -c tools/k2lab/sys_lab_prod.hip -o /tmp/sys_lab_prod.o && hipcc $F -fno-slp-vectorize -c tools/k2lab/sys_lab.hip -o /tmp/sys_lab.o &&
//   hipcc --offload-arch=gfx950 /tmp/sys_lab.o /tmp/sys_lab_prod.o -o tools/k2lab/sys_lab -lpthread
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "../power_sampler.hpp"
#include "sys_lab.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// wave-uniform tap rows by scalar loads, pinned in program order (decimate_tile.hpp:load_tap_chunk)
template <int TC> struct TapVec;
template <> struct TapVec<8> { typedef float type __attribute__((ext_vector_type(8))); };
template <int TC>
__device__ __forceinline__ typename TapVec<TC>::type load_tap_chunk(const float* taps, int c)
{
    typedef const __attribute__((address_space(4))) typename TapVec<TC>::type* ctapp;
    uint64_t a = reinterpret_cast<uint64_t>(taps) + (4u * TC) * (uint32_t)c;
    asm volatile("" : "+s"(a));
    return *reinterpret_cast<ctapp>(a);
}

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef SYS_WPE
#define SYS_WPE 3
#endif

constexpr int SYS_OUTS = 240;      // outputs per wave-strip
constexpr int SYS_STEP = 1920;     // samples between strips
constexpr int SYS_M = 19;          // m = 4 t + c = 0 .. 18

__device__ __forceinline__ float dpp_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// tap table in LDS: TT[m][p][r] = h[8 (m - p) + r] (0 outside the filter): lane p (mod 4) of every quad supplies row p of A
__device__ __forceinline__ void build_tap_table(float* TT, const float* __restrict__ taps)
{
    for (int idx = threadIdx.x; idx < SYS_M * 32; idx += blockDim.x) {
        const int m = idx >> 5, p = (idx >> 3) & 3, r = idx & 7, b = m - p;
        TT[idx] = (b >= 0 && b < 16) ? taps[8 * b + r] : 0.0f;
    }
}

// One wave-strip.  S[8c + r][re/im] are the lane's samples, acc[i][k][re/im] the travelling partial sums.
template <bool U8, int PSKIP>
__global__ void __launch_bounds__(256, SYS_WPE) k_sys(const void* __restrict__ in, int64_t x0, int nstrips, const float* __restrict__ taps,
                                                     float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float TT[SYS_M * 32];
    build_tap_table(TT, taps);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip >= nstrips) return;

    float S[32][2];
    const int64_t s0 = x0 + (int64_t)SYS_STEP * strip + 32 * lane;
    if constexpr (U8) {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + 2 * s0);
        uint4 raw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) raw[q] = src[q];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t w[4] = {raw[q].x ^ 0x80808080u, raw[q].y ^ 0x80808080u, raw[q].z ^ 0x80808080u, raw[q].w ^ 0x80808080u};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                S[8 * q + 2 * k][0] = (float)(signed char)(w[k] & 0xff);
                S[8 * q + 2 * k][1] = (float)(signed char)((w[k] >> 8) & 0xff);
                S[8 * q + 2 * k + 1][0] = (float)(signed char)((w[k] >> 16) & 0xff);
                S[8 * q + 2 * k + 1][1] = (float)(signed char)(w[k] >> 24);
            }
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + 2 * s0);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float4 v = src[q];
            S[2 * q][0] = v.x; S[2 * q][1] = v.y; S[2 * q + 1][0] = v.z; S[2 * q + 1][1] = v.w;
        }
    }

    float acc[4][4][2];
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* trow = TT + 8 * (lane & 3);
#pragma unroll
    for (int m = 0; m < SYS_M; m++) {
        const int t = m >> 2, c = m & 3;
        const f4 ta = *reinterpret_cast<const f4*>(trow + 32 * m), tb = *reinterpret_cast<const f4*>(trow + 32 * m + 4);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float tap = r < 4 ? ta[r] : tb[r - 4];
            const f4 pre = __builtin_amdgcn_mfma_f32_4x4x1f32(tap, S[8 * c + r][0], zero, 0, 0, 0);
            const f4 pim = __builtin_amdgcn_mfma_f32_4x4x1f32(tap, S[8 * c + r][1], zero, 0, 0, 0);
            const int k = r & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = m - i;
                if (b < 0 || b > 15) continue;
                if (PSKIP && 8 * b + r >= 128 - PSKIP) continue;       // the zero tap that pads 127 to 128: (+-0) * finite adds nothing
                if (b == 0 && r < 4) {
                    acc[i][k][0] = pre[i];      // 0 + p: p is never -0 here (fma(a, b, +0)), so this IS the first addition
                    acc[i][k][1] = pim[i];
                } else if (t > 0 && c == 0 && r < 4) {
                    acc[i][k][0] = dpp_shr1(acc[i][k][0]) + pre[i];   // the group moves one lane up as it enters the stage
                    acc[i][k][1] = dpp_shr1(acc[i][k][1]) + pim[i];
                } else {
                    acc[i][k][0] = acc[i][k][0] + pre[i];
                    acc[i][k][1] = acc[i][k][1] + pim[i];
                }
            }
        }
    }
    // output 0 of a group finished one stage early, one lane below: fold it there and move the result up
    float res[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 2; e++) res[i][e] = (acc[i][0][e] + acc[i][1][e]) + (acc[i][2][e] + acc[i][3][e]);
    res[0][0] = dpp_shr1(res[0][0]);
    res[0][1] = dpp_shr1(res[0][1]);
    if (lane >= 4) {
        float4* dst = reinterpret_cast<float4*>(out + 2 * ((int64_t)SYS_OUTS * strip + 4 * (lane - 4)));
        dst[0] = make_float4(res[0][0], res[0][1], res[1][0], res[1][1]);
        dst[1] = make_float4(res[2][0], res[2][1], res[3][0], res[3][1]);
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void load_strip_samples_u8(const void* __restrict__ in, int64_t s0, f2 (&S)[32])
{
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + 2 * s0);
    uint4 raw[4];
#pragma unroll
    for (int q = 0; q < 4; q++) raw[q] = src[q];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t w[4] = {raw[q].x ^ 0x80808080u, raw[q].y ^ 0x80808080u, raw[q].z ^ 0x80808080u, raw[q].w ^ 0x80808080u};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            S[8 * q + 2 * k] = f2{(float)(signed char)(w[k] & 0xff), (float)(signed char)((w[k] >> 8) & 0xff)};
            S[8 * q + 2 * k + 1] = f2{(float)(signed char)((w[k] >> 16) & 0xff), (float)(signed char)(w[k] >> 24)};
        }
    }
}

// cfloat strips: a lane's 256 contiguous bytes arrive by coalesced 16-byte loads (8 lanes = one 128-byte half row) and are
// transposed through a wave-private LDS buffer, half a strip (c = 2h, 2h+1 of every lane) at a time: 64 rows of 128 + 16 B.
// No barrier: the buffer belongs to the wave (lgkmcnt orders its own writes and reads).
constexpr int CF_ROW = 36;                  // dwords per half row (32 + 4 of padding): lane l reads row l, conflict-free per 16 lanes
constexpr int CF_WAVE_DW = 64 * CF_ROW;     // 9216 B per wave
template <bool NTL = false>
__device__ __forceinline__ void load_strip_samples_cf(const float* __restrict__ in, int64_t strip_s0, float* __restrict__ wbuf, int lane, f2 (&S)[32])
{
    const float* base = in + 2 * strip_s0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float* p = base + 64 * (8 * j + (lane >> 3)) + 32 * h + 4 * (lane & 7);
            if constexpr (NTL) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
                v[j] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v[j] = *reinterpret_cast<const float4*>(p);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<float4*>(wbuf + CF_ROW * (8 * j + (lane >> 3)) + 4 * (lane & 7)) = v[j];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 t = *reinterpret_cast<const float4*>(wbuf + CF_ROW * lane + 4 * q);
            S[16 * h + 2 * q] = f2{t.x, t.y};
            S[16 * h + 2 * q + 1] = f2{t.z, t.w};
        }
    }
}

// The same walk with the products on the VALU (VERDICT r03 "next" #1): taps are wave-uniform (every lane is in the same stage:
// tap b = 4t + c - i), so they are SGPR operands of v_pk_mul_f32 exactly as in the tile kernel; u8 input needs no LDS at all,
// no barrier, no lgkmcnt wait other than the scalar tap loads'.
// CFMODE (cfloat input): 0 = every lane reads its own 256 bytes with sixteen 16-byte loads, 1 = coalesced loads + LDS transpose
template <bool U8, int PSKIP, int CFMODE, bool MASK>
__global__ void __launch_bounds__(256, SYS_WPE) k_sysv(const void* __restrict__ in, int64_t x0, int nstrips, const float* __restrict__ taps,
                                                      float* __restrict__ out, ClkProbe* pr)
{
    __shared__ __attribute__((aligned(16))) float tbuf[(!U8 && CFMODE >= 1) ? 4 * CF_WAVE_DW : 4];
    const int lane = threadIdx.x & 63;
    // XCD-aware order (as the tile kernel's): within every 64 workgroups XCD x takes 8 consecutive ones
    const int b = blockIdx.x;
    const int wg = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    const int strip = wg * 4 + (threadIdx.x >> 6);
    if (strip >= nstrips) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    f2 S[32];
    const int64_t strip_s0 = x0 + (int64_t)SYS_STEP * strip;
    if constexpr (U8) {
        load_strip_samples_u8(in, strip_s0 + 32 * lane, S);
    } else if constexpr (CFMODE == 1) {
        load_strip_samples_cf<false>(reinterpret_cast<const float*>(in), strip_s0, tbuf + CF_WAVE_DW * (threadIdx.x >> 6), lane, S);
    } else if constexpr (CFMODE == 2) {
        load_strip_samples_cf<true>(reinterpret_cast<const float*>(in), strip_s0, tbuf + CF_WAVE_DW * (threadIdx.x >> 6), lane, S);
    } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + 2 * (strip_s0 + 32 * lane));
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float4 v = src[q];
            S[2 * q] = f2{v.x, v.y};
            S[2 * q + 1] = f2{v.z, v.w};
        }
    }
    f2 acc[4][4];
    typename TapVec<8>::type tc[16];        // the 16 rows of 8 taps; a row is loaded one m ahead of its first use
    tc[0] = load_tap_chunk<8>(taps, 0);
    // MASK: in stage t only lanes t .. 59 + t hold a group that completes inside this wave (lanes below t would continue a
    // group of the previous strip, lanes above 59 + t start one that falls off the top).  The others are switched off (EXEC):
    // under the power cap idle lanes cost time but no energy.  The stage's first four tap columns carry the DPP shift, whose
    // SOURCE lane (t - 1) has to be enabled too.
    auto stage_part = [&](auto mc, auto r_lo, auto r_hi) {
        constexpr int m = decltype(mc)::value, t = m >> 2, c = m & 3;
#pragma unroll
        for (int r = decltype(r_lo)::value; r < decltype(r_hi)::value; r++) {
            const int k = r & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int bb = m - i;
                if (bb < 0 || bb > 15) continue;
                if (PSKIP && 8 * bb + r >= 128 - PSKIP) continue;
                const float h = tc[bb][r];
                const f2 p = S[8 * c + r] * h;
                if (bb == 0 && r < 4) {
                    acc[i][k] = f2{0.f, 0.f} + p;
                } else if (t > 0 && c == 0 && r < 4) {
                    acc[i][k] = f2{dpp_shr1(acc[i][k].x) + p.x, dpp_shr1(acc[i][k].y) + p.y};
                } else {
                    acc[i][k] = acc[i][k] + p;
                }
            }
        }
    };
    auto do_m = [&](auto mc) {
        constexpr int m = decltype(mc)::value, t = m >> 2, c = m & 3;
        if (m + 1 < 16) tc[m + 1] = load_tap_chunk<8>(taps, m + 1);
        else asm volatile("" ::: "memory");
        if constexpr (MASK) {
            if (c == 0 && t > 0) {
                if (lane >= t - 1 && lane <= 59 + t) stage_part(mc, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                if (lane >= t && lane <= 59 + t) stage_part(mc, std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
            } else {
                if (lane >= t && lane <= 59 + t) stage_part(mc, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            }
        } else {
            stage_part(mc, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        }
    };
    [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (do_m(std::integral_constant<int, Ms>{}), ...); }(std::make_integer_sequence<int, SYS_M>{});
    f2 res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    res[0] = f2{dpp_shr1(res[0].x), dpp_shr1(res[0].y)};
    if (lane >= 4) {
        float4* dst = reinterpret_cast<float4*>(out + 2 * ((int64_t)SYS_OUTS * strip + 4 * (lane - 4)));
        dst[0] = make_float4(res[0].x, res[0].y, res[1].x, res[1].y);
        dst[1] = make_float4(res[2].x, res[2].y, res[3].x, res[3].y);
    }
    probe_end(pr, c0, r0);
}

// The same walk with NC * 8 samples and NC outputs per lane (NC = 4 is k_sysv): fewer stage transitions per output (8 (NS - 1) DPP
// additions per output), fewer warm-up lanes (NS - 1 of 64), at the price of 24 NC registers (2 waves per SIMD for NC = 8).  u8 only.
template <int NC, int PSKIP, int WPE>
__global__ void __launch_bounds__(256, WPE) k_sysn(const void* __restrict__ in, int64_t x0, int nstrips, const float* __restrict__ taps,
                                                  float* __restrict__ out, ClkProbe* pr)
{
    constexpr int MMAX = 15 + NC - 1, NS = MMAX / NC + 1, OUTS = (64 - (NS - 1)) * NC, STEP = OUTS * 8;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    const int wg = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    const int strip = wg * 4 + (threadIdx.x >> 6);
    if (strip >= nstrips) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    f2 S[8 * NC];
    {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + 2 * (x0 + (int64_t)STEP * strip + 8 * NC * lane));
        uint4 raw[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) raw[q] = src[q];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const uint32_t w[4] = {raw[q].x ^ 0x80808080u, raw[q].y ^ 0x80808080u, raw[q].z ^ 0x80808080u, raw[q].w ^ 0x80808080u};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                S[8 * q + 2 * k] = f2{(float)(signed char)(w[k] & 0xff), (float)(signed char)((w[k] >> 8) & 0xff)};
                S[8 * q + 2 * k + 1] = f2{(float)(signed char)((w[k] >> 16) & 0xff), (float)(signed char)(w[k] >> 24)};
            }
        }
    }
    f2 acc[NC][4];
    typename TapVec<8>::type tc[16];
    tc[0] = load_tap_chunk<8>(taps, 0);
    auto do_m = [&](auto mc) {
        constexpr int m = decltype(mc)::value, t = m / NC, c = m % NC;
        if constexpr (m + 1 < 16) tc[m + 1] = load_tap_chunk<8>(taps, m + 1);
        else asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = r & 3;
#pragma unroll
            for (int i = 0; i < NC; i++) {
                const int bb = (m - i) & 15;
                if (m - i < 0 || m - i > 15) continue;
                if (PSKIP && 8 * bb + r >= 128 - PSKIP) continue;
                const f2 p = S[8 * c + r] * tc[bb][r];
                if (bb == 0 && r < 4) acc[i][k] = f2{0.f, 0.f} + p;
                else if (t > 0 && c == 0 && r < 4) acc[i][k] = f2{dpp_shr1(acc[i][k].x) + p.x, dpp_shr1(acc[i][k].y) + p.y};
                else acc[i][k] = acc[i][k] + p;
            }
        }
    };
    [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (do_m(std::integral_constant<int, Ms>{}), ...); }(std::make_integer_sequence<int, MMAX + 1>{});
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(acc[i][k]));
    f2 res[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        res[i] = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
        // output i is complete after stage (15 + i) / NC: the early ones sit one lane below the others
        if ((15 + i) / NC < NS - 1) res[i] = f2{dpp_shr1(res[i].x), dpp_shr1(res[i].y)};
    }
    static_assert(NC == 4 || NC == 6 || NC == 8, "early outputs are exactly one stage early for these");
    if (lane >= NS - 1) {
        float2* dst = reinterpret_cast<float2*>(out + 2 * ((int64_t)OUTS * strip + NC * (lane - (NS - 1))));
#pragma unroll
        for (int i = 0; i < NC; i += 2) *reinterpret_cast<float4*>(dst + i) = make_float4(res[i].x, res[i].y, res[i + 1].x, res[i + 1].y);
    }
    probe_end(pr, c0, r0);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double us(F f, int reps, int warm = 2)
    {
        for (int i = 0; i < warm; i++) f();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        return ms * 1e3 / reps;
    }
};

static uint64_t sm64(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 27;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int rounds = argc > 3 ? atoi(argv[3]) : 1;
    const double sustain_s = argc > 4 ? atof(argv[4]) : 2.0;
    const bool with_mfma = argc > 5 ? atoi(argv[5]) != 0 : true;
    constexpr int P = 128;
    const int64_t n = (int64_t)1 << log2n;
    const int64_t nout_all = n / 8;
    const int nstrips = (int)(nout_all / SYS_OUTS) / 4 * 4;         // whole workgroups of four wave-strips
    const int64_t nout = (int64_t)nstrips * SYS_OUTS;               // outputs both kernels produce
    const int64_t n_alloc = n + 8192;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; n = 2^%d samples, %d wave-strips, %lld outputs compared\n", prop.name, prop.multiProcessorCount, log2n, nstrips, (long long)nout);

    std::vector<float> ht(P, 0.0f), ht128(P, 0.0f);
    for (int j = 0; j < 127; j++) {
        const double m = j - 63.0, fc = 1.0 / 16.0;
        const double s = m == 0 ? 2 * fc : sin(2 * M_PI * fc * m) / (M_PI * m);
        ht[j] = (float)(s * (0.54 - 0.46 * cos(2 * M_PI * j / 126.0)));
        ht128[j] = ht[j] / 128.0f;
    }
    std::vector<uint8_t> hu((size_t)2 * n_alloc);
    std::vector<float> hx((size_t)2 * n_alloc);
    uint64_t seed = 1002;
    for (auto& v : hu) v = (uint8_t)(sm64(seed) >> 56);
    for (auto& v : hx) v = (float)((double)(sm64(seed) >> 11) * (2.0 / 9007199254740992.0) - 1.0);
    uint8_t* du;
    float *dx, *dt, *dt128, *dref, *dout;
    CK(hipMalloc(&du, hu.size()));
    CK(hipMalloc(&dx, hx.size() * 4));
    CK(hipMalloc(&dt, P * 4));
    CK(hipMalloc(&dt128, P * 4));
    CK(hipMalloc(&dref, (size_t)nout_all * 8 + 4096));
    CK(hipMalloc(&dout, (size_t)nout_all * 8 + 4096));
    CK(hipMemcpy(du, hu.data(), hu.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt128, ht128.data(), P * 4, hipMemcpyHostToDevice));
    ClkProbe* dpr;
    CK(hipMalloc(&dpr, sizeof(ClkProbe)));
    Timer tm;
    std::vector<uint64_t> href((size_t)nout), hout((size_t)nout);

    const int gsys = ((nstrips / 4 + 63) / 64) * 64;
    ClkProbe* pr = nullptr;      // set to dpr for probed launches
    auto prod_u8 = [&](float* o) { lab_prod_launch(true, du, nout, dt128, o, pr); };
    auto prod_cf = [&](float* o) { lab_prod_launch(false, dx, nout, dt, o, pr); };
    auto sys_u8 = [&](float* o) { hipLaunchKernelGGL((k_sys<true, 1>), dim3(nstrips / 4), dim3(256), 0, 0, (const void*)du, (int64_t)0, nstrips, dt128, o); };
    auto sysv_u8 = [&](float* o) { hipLaunchKernelGGL((k_sysv<true, 1, 0, false>), dim3(gsys), dim3(256), 0, 0, (const void*)du, (int64_t)0, nstrips, dt128, o, pr); };
    auto sysv_cf2 = [&](float* o) { hipLaunchKernelGGL((k_sysv<false, 0, 2, false>), dim3(gsys), dim3(256), 0, 0, (const void*)dx, (int64_t)0, nstrips, dt, o, pr); };
    auto sysm_u8 = [&](float* o) { hipLaunchKernelGGL((k_sysv<true, 1, 0, true>), dim3(gsys), dim3(256), 0, 0, (const void*)du, (int64_t)0, nstrips, dt128, o, pr); };
    auto sysm_cf1 = [&](float* o) { hipLaunchKernelGGL((k_sysv<false, 0, 1, true>), dim3(gsys), dim3(256), 0, 0, (const void*)dx, (int64_t)0, nstrips, dt, o, pr); };
    auto sysn = [&](auto kern, int outs_per_strip, float* o) {
        const int ns = (int)(nout / outs_per_strip) / 4 * 4;
        hipLaunchKernelGGL(kern, dim3(((ns / 4 + 63) / 64) * 64), dim3(256), 0, 0, (const void*)du, (int64_t)0, ns, dt128, o, pr);
        return (int64_t)ns * outs_per_strip;
    };
    auto sysv_cf0 = [&](float* o) { hipLaunchKernelGGL((k_sysv<false, 0, 0, false>), dim3(gsys), dim3(256), 0, 0, (const void*)dx, (int64_t)0, nstrips, dt, o, pr); };
    auto sysv_cf1 = [&](float* o) { hipLaunchKernelGGL((k_sysv<false, 0, 1, false>), dim3(gsys), dim3(256), 0, 0, (const void*)dx, (int64_t)0, nstrips, dt, o, pr); };

    auto fetch = [&](float* d, std::vector<uint64_t>& h) { CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, (size_t)nout * 8, hipMemcpyDeviceToHost)); };
    auto check = [&](const char* name) {
        fetch(dout, hout);
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < (size_t)nout; i++) if (hout[i] != href[i]) { if (!bad) first = i; bad++; }
        printf("    check %-28s %s (%zu of %lld outputs differ", name, bad ? "MISMATCH" : "bit-exact", bad, (long long)nout);
        if (bad) printf("; first at %zu: want %016llx got %016llx", first, (unsigned long long)href[first], (unsigned long long)hout[first]);
        printf(")\n");
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
    };
    auto report = [&](const char* name, double us, double bytes_in) {
        printf("%-40s %8.1f us  %7.1f Gsamp/s  read %5.3f TB/s  %6.2f TFLOP/s (64 flop/sample)\n", name, us, n / us / 1e3, bytes_in * n / us / 1e6, 64.0 * n / us / 1e6);
        fflush(stdout);
    };
    for (int round = 0; round < rounds; round++) {
        printf("---- round %d\n", round);
        prod_u8(dref);
        fetch(dref, href);
        report("u8: production tile kernel (FULL)", tm.us([&] { prod_u8(dref); }, reps), 2);
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        if (with_mfma) {
            report("u8: systolic, MFMA products", tm.us([&] { sys_u8(dout); }, reps), 2);
            check("u8 systolic MFMA");
        }
        report("u8: systolic, VALU products", tm.us([&] { sysv_u8(dout); }, reps), 2);
        check("u8 systolic VALU");
        for (int nc : {6, 8}) {
            CK(hipMemset(dout, 0xff, (size_t)nout * 8));
            int64_t covered = 0;
            const double us = tm.us([&] { covered = nc == 6 ? sysn(k_sysn<6, 1, 3>, 61 * 6, dout) : sysn(k_sysn<8, 1, 2>, 62 * 8, dout); }, reps);
            report(nc == 6 ? "u8: systolic, 48 samples per lane (3 w/SIMD)" : "u8: systolic, 64 samples per lane (2 w/SIMD)", us * (double)nout / (double)covered, 2);
            fetch(dout, hout);
            size_t bad = 0;
            for (int64_t q = 0; q < covered; q++) bad += hout[q] != href[q];
            printf("    check NC=%d: %s (%zu of %lld outputs differ)\n", nc, bad ? "MISMATCH" : "bit-exact", bad, (long long)covered);
        }
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        report("u8: systolic VALU, idle lanes masked", tm.us([&] { sysm_u8(dout); }, reps), 2);
        check("u8 systolic VALU masked");
        prod_cf(dref);
        fetch(dref, href);
        report("cfloat: production tile kernel (FULL)", tm.us([&] { prod_cf(dref); }, reps), 8);
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        report("cfloat: systolic VALU, direct loads", tm.us([&] { sysv_cf0(dout); }, reps), 8);
        check("cfloat systolic VALU direct");
        report("cfloat: systolic VALU, LDS transpose", tm.us([&] { sysv_cf1(dout); }, reps), 8);
        check("cfloat systolic VALU transposed");
        report("cfloat: systolic, transposed, nt loads", tm.us([&] { sysv_cf2(dout); }, reps), 8);
        check("cfloat systolic transposed nt");
        report("cfloat: systolic VALU, transposed, masked", tm.us([&] { sysm_cf1(dout); }, reps), 8);
        check("cfloat systolic VALU transposed masked");
    }
    // sustained rows: every variant for `sustain_s` seconds with the socket power and sclk sampled every 10 ms (hwmon), then a
    // short probed run (shader cycles per wave-strip / workgroup and the shader clock from s_memtime : s_memrealtime)
    if (sustain_s > 0) {
        char bdf[64] = {0};
        CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
        PowerSampler ps(bdf);
        printf("---- sustained, %.1f s per row; power telemetry of %s %s\n", sustain_s, bdf, ps.available() ? "from hwmon" : "NOT available");
        auto sustained = [&](const char* name, auto f, double us_guess) {
            pr = nullptr;
            const int reps2 = (int)(sustain_s * 1e6 / us_guess) + 1;
            ps.start();
            const double us = tm.us(f, reps2, 0);
            const PowerStats st = ps.finish();
            pr = dpr;
            CK(hipMemset(dpr, 0, sizeof(ClkProbe)));
            const double usp = tm.us(f, 50, 0);
            ClkProbe h;
            CK(hipMemcpy(&h, dpr, sizeof h, hipMemcpyDeviceToHost));
            pr = nullptr;
            printf("%-38s %7.1f us %6.1f Gsamp/s | %3d smp %4.0f W (%4.0f..%4.0f) sclk %4.0f (%4.0f..%4.0f) MHz | probed %7.1f us: %4.0f MHz, %6.0f cyc per wave\n", name, us,
                   n / us / 1e3, st.n, st.mean_w, st.min_w, st.max_w, st.mean_sclk_mhz, st.min_sclk_mhz, st.max_sclk_mhz, usp,
                   h.rt ? (double)h.cyc / (double)h.rt * 100.0 : 0.0, h.n ? (double)h.cyc / h.n : 0.0);
            fflush(stdout);
        };
        for (int ab = 0; ab < 2; ab++) {
            sustained("u8: production tile kernel", [&] { prod_u8(dref); }, 200);
            sustained("u8: systolic VALU", [&] { sysv_u8(dout); }, 200);
            sustained("u8: systolic VALU, masked", [&] { sysm_u8(dout); }, 200);
            sustained("u8: systolic NC=6 (x covered/all)", [&] { sysn(k_sysn<6, 1, 3>, 61 * 6, dout); }, 200);
            sustained("u8: systolic NC=8 (x covered/all)", [&] { sysn(k_sysn<8, 1, 2>, 62 * 8, dout); }, 200);
            sustained("cfloat: production tile kernel", [&] { prod_cf(dref); }, 260);
            sustained("cfloat: systolic VALU, LDS transpose", [&] { sysv_cf1(dout); }, 260);
            sustained("cfloat: systolic VALU, transp, masked", [&] { sysm_cf1(dout); }, 260);
            sustained("cfloat: systolic, transposed, nt loads", [&] { sysv_cf2(dout); }, 260);
        }
    }
    return 0;
}
