// kernels_systolic.hip -- K2 as a register-resident systolic walk (round 4): the decimate-by-8, 128-tap complex decimator of the
// FM chain (decimateAVXRC, c_sources/decimate.c:105-113 -> avx_dotprod_R common.h:58-72 -> avx_hadd_C common.h:82-90) without
// LDS in its multiply-add loop.
//
//   out[o] = (L0 + L1) + (L2 + L3),   L_k = sum_{j = k (mod 4)} h[j] * x[8o + j],  each L_k from +0 in increasing j, separate
//   multiply and add.  Write j = 8b + r (b = 0..15, r = 0..7).
//
// * A wave owns a STRIP of 2048 input samples.  Lane l keeps samples 32l .. 32l+31 (sample 8c + r, c = 0..3) in 64 VGPRs and
//   never moves them.  u8 IQ: four 16-byte loads per lane straight from global memory (a wave reads 4 KiB contiguously),
//   xor 0x80 + signed-byte convert, taps pre-scaled by 1/128 (exactness: decimate_tile.hpp, Stage::store_regs).  cfloat IQ:
//   coalesced 16-byte loads (8 lanes = one 128-byte half row), transposed once through a wave-private LDS buffer -- the only
//   LDS traffic of the kernel (16 ds_write_b128 + 16 ds_read_b128 per lane and 240 outputs; the tile kernel: 24 + 138).
// * The 32 partial sums of the output GROUP q = outputs 4q .. 4q+3 (x 4 partials x re/im) TRAVEL: they start in lane q and
//   move one lane up at every stage of 32 taps, as the DPP operand of the stage's first addition (v_add_f32_dpp wave_shr:1 --
//   the move costs no instruction of its own).  In stage t lane l works on group l - t: output 4(l-t) + i meets the lane's
//   sample 8c + r under tap b = 4t + c - i, which is the same for every lane -> taps are SGPR operands (s_load_dwordx8 one row
//   of 8 ahead), exactly as in the tile kernel.  Five stages complete a group.  Partial k still sees taps k, k+4, k+8, .. in
//   increasing order from +0 (b ascending, r = k before r = k + 4): the SAME BITS as the tile kernel and the reference.
// * Lanes 0..3 of a wave only warm the pipe up: a wave turns 2048 samples into 240 outputs, consecutive strips overlap by 128
//   samples (the tile kernel's tiles overlap by 120).  Output 0 of a group is complete one stage (= one lane) early; its
//   folded result is moved up by one more DPP so that a lane stores 4 consecutive outputs (two 16-byte stores).
// * No barrier, no LDS wait in the loop, 4 waves per SIMD (102..122 VGPRs).
//
// Why: under the 1400 W socket cap the tile kernel runs at 2.0 GHz (u8) / 1.63 GHz (cfloat) and this form at 2.13..2.18 / 1.6
// GHz with fewer instructions per output: 168 vs 178 us (u8) and 227 vs 239 us (cfloat) per 2^27 samples, sustained, same process
// (tools/k2lab/sys_lab.hip; profiles/k2lab/r04_systolic.txt).  DESIGN.md section 5 has the energy accounting.
//
// The launch covers ALL `count` outputs: whole strips take the fast body, the ragged end (a strip whose samples or outputs
// run past the launch) a guarded one in the same launch.  Cross outputs (seam straddlers) are rewritten afterwards by the
// caller's fix-up kernel, as with the tile kernel.  Compiled with -fno-slp-vectorize (build.py): the packed operations are
// written out as 2-vectors, and the vectoriser would turn the scalar DPP additions into v_mov_dpp + v_pk_add.
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "demod_forms.hpp"
#include "lab.hpp"

#ifndef SDRHIP_SYSTOLIC_ROTCOST
#define SDRHIP_SYSTOLIC_ROTCOST 0
#endif

namespace sdrhip {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));

constexpr int kStripOuts = 240;        // outputs per wave-strip
constexpr int kStripStep = 1920;       // samples between strips (= 8 * 240)
constexpr int kStripSpan = 2048;       // samples a strip reads
constexpr int kM = 19;                 // m = 4t + c = 0 .. 18
constexpr int kWavesPerWg = 4;

__device__ __forceinline__ float dpp_shr1(float v)
{
    // wave_shr:1, lanes without a source read 0 (bound_ctrl:0): lets the DPP fold into the consuming v_add_f32
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

template <int... Ms, class F>
__device__ __forceinline__ void for_each_m(std::integer_sequence<int, Ms...>, F&& f)
{
    (f(std::integral_constant<int, Ms>{}), ...);
}

// one row of 8 taps by a scalar load, pinned in program order (decimate_tile.hpp:load_tap_chunk)
__device__ __forceinline__ f8 load_tap_row(const float* taps, int b)
{
    typedef const __attribute__((address_space(4))) f8* ctapp;
    uint64_t a = reinterpret_cast<uint64_t>(taps) + 32u * (uint32_t)b;
    asm volatile("" : "+s"(a));
    return *reinterpret_cast<ctapp>(a);
}

__device__ __forceinline__ void convert_u8x16(const uint4 raw, f2* S)
{
    const uint32_t w[4] = {raw.x ^ 0x80808080u, raw.y ^ 0x80808080u, raw.z ^ 0x80808080u, raw.w ^ 0x80808080u};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        S[2 * k] = f2{(float)(signed char)(w[k] & 0xff), (float)(signed char)((w[k] >> 8) & 0xff)};
        S[2 * k + 1] = f2{(float)(signed char)((w[k] >> 16) & 0xff), (float)(signed char)(w[k] >> 24)};
    }
}

// cfloat transpose buffer of one wave: 64 half rows of 128 B + 16 B of padding (lane l reads row l: 16 lanes hit 16 distinct
// 16-byte bank groups)
constexpr int kCfRow = 36;
constexpr int kCfWaveDw = 64 * kCfRow;

// `avail`: samples that exist from the strip's first one on (>= kStripSpan for a whole strip); beyond them zeros (u8: 128)
template <bool U8, bool WHOLE>
__device__ __forceinline__ void load_strip(const void* __restrict__ in, int64_t strip_s0, int64_t avail, float* __restrict__ wbuf, int lane, f2 (&S)[32])
{
    if constexpr (U8) {
        const uint8_t* base = reinterpret_cast<const uint8_t*>(in) + 2 * (strip_s0 + 32 * lane);
        uint4 raw[4];
        if constexpr (WHOLE) {
#pragma unroll
            for (int q = 0; q < 4; q++) raw[q] = reinterpret_cast<const uint4*>(base)[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t s = 32 * lane + 8 * q;               // first sample of the vector, relative to the strip
                if (s + 8 <= avail) {
                    raw[q] = reinterpret_cast<const uint4*>(base)[q];
                } else {
                    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                    for (int e = 0; e < 16; e++)
                        if (s + e / 2 < avail) w[e >> 2] = (w[e >> 2] & ~(0xffu << (8 * (e & 3)))) | ((uint32_t)base[16 * q + e] << (8 * (e & 3)));
                    raw[q] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) convert_u8x16(raw[q], &S[8 * q]);
    } else {
        const float* base = reinterpret_cast<const float*>(in) + 2 * strip_s0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int row = 8 * j + (lane >> 3);                // = the lane whose samples these are
                const int s = 32 * row + 16 * h + 2 * (lane & 7);   // first of the two samples of the vector
                const float* p = base + 2 * s;
                if (WHOLE || s + 2 <= avail) {
                    // non-temporal: every byte is read once (bar the 6 % strip overlap): 226 -> 221 us per 2^27 samples, and the
                    // copy-only stream of this shape gains 15 % from the same hint
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v t4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
                    v[j] = make_float4(t4.x, t4.y, t4.z, t4.w);
                } else {
                    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (s < avail) { v[j].x = p[0]; v[j].y = p[1]; }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) *reinterpret_cast<float4*>(wbuf + kCfRow * (8 * j + (lane >> 3)) + 4 * (lane & 7)) = v[j];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float4 t = *reinterpret_cast<const float4*>(wbuf + kCfRow * lane + 4 * q);
                S[16 * h + 2 * q] = f2{t.x, t.y};
                S[16 * h + 2 * q + 1] = f2{t.z, t.w};
            }
        }
    }
}

// PSKIP: the last PSKIP taps are the zero padding (Filter.hs:146-148) and every sample is finite (u8 input): their MACs are
// skipped -- exact, see decimate_tile.hpp:mac_window.
// DEMOD (round 4, K2 + K3 in one kernel): the decimator's outputs never reach HBM -- the lane that holds outputs o .. o+3 demodulates
// them in place (fmDemod, Demod.hs:21-46: y[k] = phase(d[k] * conj(d[k-1])); output o's predecessor comes from the lane below by one
// more DPP) and stores y.  Strips then advance by 239 outputs: the first output of a strip is only the predecessor of the
// second (the previous strip stores its y), exactly the role decimator output kd0 = ky0 - 1 plays for a whole launch.
// `count` = decimator outputs of the launch, `yshift` = 1 when output 0 is such a predecessor (else y[0] = phase(d[0] * conj(0)),
// the stream's very first sample, Demod.hs:41).
template <bool U8, int PSKIP, bool WHOLE, bool DEMOD>
__device__ __forceinline__ void systolic_strip(const void* __restrict__ in, int64_t x0, int strip, int count, const float* __restrict__ taps,
                                               float* __restrict__ out, float* __restrict__ wbuf, int lane, int yshift)
{
    constexpr int kOuts = DEMOD ? kStripOuts - 1 : kStripOuts;      // outputs a strip advances by
    constexpr int kStep = 8 * kOuts;
    f2 S[32];
    const int64_t strip_s0 = x0 + (int64_t)kStep * strip;
    // samples of the launch: (count - 1) * 8 + 128 from x0 on
    const int64_t avail = WHOLE ? kStripSpan : ((int64_t)(count - 1) * 8 + 128) - (int64_t)kStep * strip;
    load_strip<U8, WHOLE>(in, strip_s0, avail, wbuf, lane, S);

    f2 acc[4][4];
    f8 tc[16];
    tc[0] = load_tap_row(taps, 0);
    auto do_m = [&](auto mc) {
        constexpr int m = decltype(mc)::value, t = m >> 2, c = m & 3;
        if constexpr (m + 1 < 16) tc[m + 1] = load_tap_row(taps, m + 1);
        else asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = r & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = (m - i) & 15;                        // tap row of output i (only used when m - i is in 0 .. 15)
                if (m - i < 0 || m - i > 15) continue;
                if (PSKIP && 8 * b + r >= 128 - PSKIP) continue;
                const f2 p = S[8 * c + r] * tc[b][r];
                if (b == 0 && r < 4) {
                    acc[i][k] = f2{0.f, 0.f} + p;                  // the first addition of the partial: +0 + product
                } else if (t > 0 && c == 0 && r < 4) {
#if SDRHIP_SYSTOLIC_ROTCOST
                    // MEASUREMENT ONLY (LABNOTES round 5, VERDICT r04 "next" 4): what a rotating walk would have to add -- the partial sums
                    // that leave lane 63 entering lane 0 of the next strip -- costs at least one lane-0 patch per value and stage boundary:
                    // a v_mov_b32_dpp wave_ror:1 of the value (here of the accumulator itself; the result is thrown away, the
                    // instruction is not)
                    {
                        float rx, ry;
                        asm volatile("v_mov_b32_dpp %0, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 wave_ror:1 row_mask:0xf bank_mask:0xf"
                                     : "=&v"(rx), "=&v"(ry) : "v"(acc[i][k].x), "v"(acc[i][k].y));
                    }
#endif
                    acc[i][k] = f2{dpp_shr1(acc[i][k].x) + p.x, dpp_shr1(acc[i][k].y) + p.y};   // the group enters the stage one lane up
                } else {
                    acc[i][k] = acc[i][k] + p;
                }
            }
        }
    };
    for_each_m(std::make_integer_sequence<int, kM>{}, do_m);

    // the last stage's arithmetic must not sink into the `lane >= 4` block below: its first additions carry the DPP move, which
    // needs every lane -- sunk, each becomes a v_mov_dpp outside plus an addition inside (26 extra instructions per strip; measured
    // in one process: decimate stage 0.681-0.683 ms with this fence against 0.689-0.693 without)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(acc[i][k]));
    f2 res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    res[0] = f2{dpp_shr1(res[0].x), dpp_shr1(res[0].y)};           // output 0 of the group sat one lane below
    if constexpr (!DEMOD) {
        if (lane >= 4) {
            const int o = kStripOuts * strip + 4 * (lane - 4);
            if (WHOLE || o + 4 <= count) {
                float4* dst = reinterpret_cast<float4*>(out + 2 * (int64_t)o);
                dst[0] = make_float4(res[0].x, res[0].y, res[1].x, res[1].y);
                dst[1] = make_float4(res[2].x, res[2].y, res[3].x, res[3].y);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (o + i < count) *reinterpret_cast<float2*>(out + 2 * (int64_t)(o + i)) = make_float2(res[i].x, res[i].y);
            }
        }
    } else {
        // the predecessor of the lane's first output: the last output of the lane below (lane 4: of the previous strip -- not here)
        const f2 below = f2{dpp_shr1(res[3].x), dpp_shr1(res[3].y)};
        if (lane >= 4) {
            const int o = 4 * (lane - 4);                          // strip-local output of res[0]
            const int j = kOuts * strip + o;                       // launch-local decimator output
            const bool first_of_stream = strip == 0 && o == 0 && yshift == 0;
            const f2 p0 = first_of_stream ? f2{0.f, 0.f} : below;
            float y[4];
            y[0] = fm_phase_sel(make_float2(res[0].x, res[0].y), make_float2(p0.x, p0.y));
#pragma unroll
            for (int i = 1; i < 4; i++) y[i] = fm_phase_sel(make_float2(res[i].x, res[i].y), make_float2(res[i - 1].x, res[i - 1].y));
            float* dst = out + ((int64_t)j - yshift);              // y of output j
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            if (o > 0 && (WHOLE || j + 4 <= count)) {
                *reinterpret_cast<f4u*>(dst) = f4u{y[0], y[1], y[2], y[3]};      // 4-byte aligned 16-byte store
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool mine = (o + i > 0) || first_of_stream;           // a strip's output 0 belongs to the previous strip
                    if (mine && j + i < count) dst[i] = y[i];
                }
            }
        }
    }
}

template <bool U8, int PSKIP, bool DEMOD = false>
__global__ void __launch_bounds__(64 * kWavesPerWg, 4) k_decimate_systolic(const void* __restrict__ in, int64_t x0 /* sample of output 0's window in `in` */,
                                                                          int count, const float* __restrict__ taps, float* __restrict__ out,
                                                                          int nwhole /* strips [0, nwhole) are whole */, int nstrips, int yshift)
{
    __shared__ __attribute__((aligned(16))) float tbuf[U8 ? 4 : kWavesPerWg * kCfWaveDw];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order (the tile kernel's): workgroup b runs on XCD b % 8; within every 64 consecutive workgroups XCD x takes 8
    // consecutive ones, so 7 of 8 strip-to-strip overlaps of a workgroup's neighbours hit in the same L2
    const int b = blockIdx.x;
    const int wg = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    const int strip = wg * kWavesPerWg + wave;
    if (strip >= nstrips) return;
    float* wbuf = tbuf + (U8 ? 0 : kCfWaveDw * wave);
    if (strip < nwhole) systolic_strip<U8, PSKIP, true, DEMOD>(in, x0, strip, count, taps, out, wbuf, lane, yshift);
    else systolic_strip<U8, PSKIP, false, DEMOD>(in, x0, strip, count, taps, out, wbuf, lane, yshift);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Seams of the fused decimate + fmDemod launch.  At a buffer boundary E (a multiple of the reference's block) the 15 decimator outputs
// whose windows straddle E are computed by the reference in sequential order (decimateCrossHighLevel, FilterInternal.hs:397-402); they
// change 16 demodulated outputs, y[E/8 - 15 .. E/8].  One group of 32 threads per seam: the 256 samples around E are staged once (one
// 16-byte load per thread), threads 0..16 compute d[E/8 - 16 .. E/8] -- the two ends in the SIMD order (they are ordinary One outputs,
// needed as neighbours), the 15 in between sequentially -- and threads 0..15 demodulate and overwrite the 16 y.
constexpr int kFixSeamsPerWg = 8;
constexpr int kFixRow = 256 + 256 / 8;      // one float2 of padding after every 8 samples
__global__ void __launch_bounds__(32 * kFixSeamsPerWg) k_decimate_demod_crossfix(const uint8_t* __restrict__ in, int64_t in_base, int64_t kd0, int count,
                                                                                int yshift, const float* __restrict__ xtaps, float* __restrict__ y,
                                                                                int64_t first_seam, int nseams, int64_t seam)
{
    __shared__ float2 rows[kFixSeamsPerWg][kFixRow];
    __shared__ float2 dd[kFixSeamsPerWg][17];
    const int g = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int si = blockIdx.x * kFixSeamsPerWg + g;
    const bool live = si < nseams;
    const int64_t E = (first_seam + si) * seam;                                 // global sample index of the boundary
    const int64_t lo = kd0 * 8, hi = (kd0 + count - 1) * 8 + 128;               // samples the launch may read
    if (live) {
        const int64_t s_first = E - 128 + 8 * t;                                // this thread's 8 samples
        float2 smp[8];
        if (s_first >= lo && s_first + 8 <= hi) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + 2 * (s_first - in_base));
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                smp[2 * k] = make_float2(((float)(w[k] & 0xff) - 128.0f) * (1.0f / 128.0f), ((float)((w[k] >> 8) & 0xff) - 128.0f) * (1.0f / 128.0f));
                smp[2 * k + 1] = make_float2(((float)((w[k] >> 16) & 0xff) - 128.0f) * (1.0f / 128.0f), ((float)(w[k] >> 24) - 128.0f) * (1.0f / 128.0f));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int64_t sk = s_first + k;
                smp[k] = make_float2(0.f, 0.f);
                if (sk >= lo && sk < hi) {
                    const uchar2 u = *reinterpret_cast<const uchar2*>(in + 2 * (sk - in_base));
                    smp[k] = make_float2(((float)u.x - 128.0f) * (1.0f / 128.0f), ((float)u.y - 128.0f) * (1.0f / 128.0f));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) rows[g][9 * t + k] = smp[k];                 // sample 8t + k sits at 8t + k + t
    }
    __syncthreads();
    if (live && t < 17) {
        const float2* w = &rows[g][9 * t];                                      // window of candidate m = E/8 - 16 + t starts at sample 8t
        float2 r;
        if (t == 0 || t == 16) {
            float2 L[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll 4
            for (int j = 0; j < 128; j++) {
                const float2 x = w[j + j / 8];
                const float h = xtaps[j];
                L[j & 3].x = L[j & 3].x + x.x * h;
                L[j & 3].y = L[j & 3].y + x.y * h;
            }
            r = make_float2((L[0].x + L[1].x) + (L[2].x + L[3].x), (L[0].y + L[1].y) + (L[2].y + L[3].y));
        } else {
            float re = 0.0f, im = 0.0f;
#pragma unroll 4
            for (int j = 0; j < 128; j++) {
                const float2 x = w[j + j / 8];
                const float h = xtaps[j];
                re = re + x.x * h;
                im = im + x.y * h;
            }
            r = make_float2(re, im);
        }
        dd[g][t] = r;
    }
    __syncthreads();
    if (live && t < 16) {
        const int64_t k = E / 8 - 15 + t;                                       // y[k] = phase(d[k] conj d[k-1])
        const int64_t j = k - kd0;
        if (j >= 0 && j < count && (j > 0 || yshift == 0)) {
            const float2 prev = j > 0 ? dd[g][t] : make_float2(0.f, 0.f);
            y[j - yshift] = fm_phase_sel(dd[g][t + 1], prev);
        }
    }
}

std::atomic<int>& systolic_flag()
{
    static std::atomic<int> f{getenv("SDRHIP_SYSTOLIC") ? atoi(getenv("SDRHIP_SYSTOLIC")) : 1};
    return f;
}
std::atomic<long long> g_systolic_launches{0};

}  // namespace

// How a launch of `count` outputs is cut into wave-strips (host arithmetic, testable without a GPU: sdrhip_debug_systolic_plan).
// plain: strip t covers outputs 240 t .. 240 t + 239 and reads samples 1920 t .. 1920 t + 2047; demod: strips advance by 239
// outputs (strip t covers 239 t .. 239 t + 239, its first output being only a predecessor).  Strips [0, nwhole) have all their
// outputs wanted and all their samples inside the launch's (count - 1) * 8 + 128.
void systolic_plan(int count, bool demod, int* nstrips, int* nwhole)
{
    if (demod) {
        constexpr int kOuts = kStripOuts - 1;
        *nstrips = count > 1 ? (count - 1 + kOuts - 1) / kOuts : 1;
        *nwhole = count >= kStripOuts + 1 ? (count - 1 - kStripOuts) / kOuts + 1 : 0;
    } else {
        *nstrips = (count + kStripOuts - 1) / kStripOuts;
        int w = count / kStripOuts;
        while (w > 0 && (int64_t)kStripStep * (w - 1) + kStripSpan > (int64_t)(count - 1) * 8 + 128) w--;
        *nwhole = w;
    }
}

void set_systolic(int on) { systolic_flag().store(on); }
long long systolic_launch_count() { return g_systolic_launches.load(); }

// The SIMD ("One") outputs of a decimate-by-8, 128-tap, AVX-order launch.  False = not this kernel's shape or too small to be
// worth it (the tile kernel computes its Cross outputs in place for launch-bound sizes); the caller then takes the tile kernel.
bool launch_decimate_c4_systolic(hipStream_t s, const Geom& g, const float* d_taps, int P, const void* d_in, bool in_is_u8, float* d_out,
                                 bool last_tap_zero)
{
    if (!systolic_flag().load(std::memory_order_relaxed)) return false;
    if (g.I != 1 || g.D != 8 || P != 128 || g.Lp != 128 || g.count < 64 * kStripOuts * kWavesPerWg) return false;
    const int64_t x0 = g.k_begin * g.D - g.in_base;
    const uintptr_t base = reinterpret_cast<uintptr_t>(d_in);
    if (((base + (in_is_u8 ? 2 : 8) * (uintptr_t)x0) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return false;
    // strip n is whole when its 240 outputs are wanted and its 2048 samples exist: 1920 n + 2048 <= (count - 1) * 8 + 128
    int nstrips, nwhole;
    systolic_plan(g.count, false, &nstrips, &nwhole);
    const int nwg = (nstrips + kWavesPerWg - 1) / kWavesPerWg;
    const dim3 grid(((nwg + 63) / 64) * 64), block(64 * kWavesPerWg);
    if (in_is_u8) {
        if (last_tap_zero) hipLaunchKernelGGL((k_decimate_systolic<true, 1>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, 0);
        else hipLaunchKernelGGL((k_decimate_systolic<true, 0>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, 0);
    } else {
        hipLaunchKernelGGL((k_decimate_systolic<false, 0>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, 0);
    }
    g_systolic_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// K2 + K3 of the FM chain in one launch (+ the seam launch): u8 IQ -> decimate by 8 (128 taps, AVX order) -> fmDemod, the decimated
// stream never written.  Decimator outputs [kd0, kd1) are computed, demodulated outputs [ky0, kd1) stored at d_y[k - ky0], where
// ky0 = kd0 + 1 (output kd0 is only y[ky0]'s predecessor) or ky0 = kd0 = 0 (the stream's first sample, predecessor 0).
// false = not this shape / too small: the caller runs the two stages on their own.
bool launch_decimate_demod_systolic(hipStream_t s, const uint8_t* d_in, int64_t in_base, int64_t kd0, int64_t kd1, int64_t ky0,
                                    const float* d_scaled_taps, const float* d_plain_taps, int P, bool last_tap_zero, int64_t seam_block,
                                    float* d_y)
{
    if (!systolic_flag().load(std::memory_order_relaxed)) return false;
    const int64_t n = kd1 - kd0;
    if (P != 128 || n < 64 * kStripOuts * kWavesPerWg || n >= (int64_t)0x7fffffff || seam_block < 0) return false;
    if (!(ky0 == kd0 + 1 || (ky0 == kd0 && kd0 == 0))) return false;
    if (seam_block != 0 && (seam_block % 8 != 0 || seam_block < 256)) return false;
    const int64_t x0 = kd0 * 8 - in_base;
    if (x0 < 0 || ((reinterpret_cast<uintptr_t>(d_in) + 2 * (uintptr_t)x0) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_y) & 3) != 0) return false;
    const int count = (int)n, yshift = (int)(ky0 - kd0);
    // strip t is whole when all of its outputs 239 t .. 239 t + 239 exist (then so do its 2048 samples)
    int nstrips, nwhole;
    systolic_plan(count, true, &nstrips, &nwhole);
    const int nwg = (nstrips + kWavesPerWg - 1) / kWavesPerWg;
    const dim3 grid(((nwg + 63) / 64) * 64), block(64 * kWavesPerWg);
    if (last_tap_zero) hipLaunchKernelGGL((k_decimate_systolic<true, 1, true>), grid, block, 0, s, (const void*)d_in, x0, count, d_scaled_taps, d_y, nwhole, nstrips, yshift);
    else hipLaunchKernelGGL((k_decimate_systolic<true, 0, true>), grid, block, 0, s, (const void*)d_in, x0, count, d_scaled_taps, d_y, nwhole, nstrips, yshift);
    g_systolic_launches.fetch_add(1, std::memory_order_relaxed);
    if (seam_block != 0) {
        const int64_t v_lo = kd0 * 8, v_hi = (kd1 - 1) * 8 + 128;
        const int64_t first = v_lo / seam_block + 1, last = (v_hi - 1) / seam_block;     // boundaries strictly inside the launch's samples
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            hipLaunchKernelGGL(k_decimate_demod_crossfix, dim3((nseams + kFixSeamsPerWg - 1) / kFixSeamsPerWg), dim3(32 * kFixSeamsPerWg), 0, s, d_in, in_base,
                               kd0, count, yshift, d_plain_taps, d_y, first, nseams, seam_block);
        }
    }
    return true;
}

}  // namespace sdrhip
