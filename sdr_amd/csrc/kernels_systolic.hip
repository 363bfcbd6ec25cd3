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

#include "kernels.hpp"
#include "decimate_tile.hpp"      // decimate_c_crossfix_wg: the seam fix-up's workgroup body (round 6: interleaved into this kernel's launch)

namespace sdrhip {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));

// PT = rows of 8 taps (16: the 128-tap filter this file was written for; 8: up to 64 taps -- round 6, the reference example's own
// 51 -> 52-tap RF decimator with the 12 taps of padding skipped).  A group's partial sums travel (PT + 2) / 4 stages, so that many
// lanes at the bottom of a wave only warm the pipe up.
template <int PT>
struct Sys {
    static_assert(PT == 16 || PT == 8, "tap rows: a multiple of four (output 0 of a group then completes one stage before the others)");
    static constexpr int kWarm = (PT + 2) / 4;              // 4 / 2 lanes without a finished group
    static constexpr int kOuts = 4 * (64 - kWarm);          // outputs per wave-strip: 240 / 248
    static constexpr int kStep = 8 * kOuts;                 // samples between strips: 1920 / 1984
    static constexpr int kM = PT + 3;                       // m = 4t + c = 0 .. PT + 2
};
constexpr int kStripOuts = Sys<16>::kOuts;      // the 128-tap kernel's figures (host arithmetic, tests)
constexpr int kStripSpan = 2048;       // samples a strip reads
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
template <bool U8, bool WHOLE, bool NTL>
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
                    if constexpr (NTL) {
                        const f4v t4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
                        v[j] = make_float4(t4.x, t4.y, t4.z, t4.w);
                    } else {
                        v[j] = *reinterpret_cast<const float4*>(p);
                    }
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
// (Round 4 also built fmDemod into this kernel's epilogue -- K2 + K3 in one launch, the decimated stream never written -- bit-equal and
// slower: the pair 0.91 ms against 0.69 + 0.16.  tools/lab_variants/decimate_demod_systolic.hip, LABNOTES.)
// FIX (round 6): the launch computes its own Cross outputs.  The 15 outputs in front of every buffer boundary (window straddling a
// multiple of the reference's block, FilterInternal.hs:397-402: sequential order) used to be rewritten by a fix-up launch AFTER this
// kernel -- 15 us of load -> barrier -> 128-step dependent chains on an otherwise idle chip (6.7 % of BASELINE configs[1]'s period).
// Now the fix-up's workgroups are part of THIS launch, spread through it in blocks of 64 (one block every `period_g` groups of 64
// workgroups: the XCD-aware order of the strips is kept), and run beside the strips; a strip does not store the Cross outputs it
// computed in SIMD order (which of a strip's outputs are Cross is scalar arithmetic: k mod seamK >= seamK - 15), so the two kinds of
// workgroup write disjoint outputs and need no order between them.
struct SystolicFix {
    Geom g;                 // the launch's geometry as the fix-up kernel takes it
    const float* xtaps;     // plain taps of the sequential outputs
    int64_t first_seam;
    int nseams;
    int nfix_groups;        // blocks of 64 fix-up workgroups
    int period_g;           // group G (= blockIdx.x >> 6) is a fix-up block when G % period_g == period_g - 1 and G / period_g < nfix_groups
    int seamK;              // outputs per buffer (seam_block / 8), >= 256
    int k0mod;              // k_begin mod seamK
    int nx;                 // Cross outputs in front of a boundary: (Lp - 1) / 8 = 15 for 128 taps, 6 for 52
    int lp;                 // the filter's own length (128, or 52 on the 64-tap instantiation): the launch's samples end at (count - 1) * 8 + lp
};
constexpr int kFixPer = 16, kFixSpw = 16;      // 16 candidate slots per seam, 16 seams per workgroup (k_decimate_c_crossfix's shape)
constexpr int kFixLdsFloats = 2 * kFixSpw * crossfix_row_float2<8, 128, kFixPer>();

template <bool U8, int PSKIP, bool WHOLE, bool NTL, bool FIX, int PT>
__device__ __forceinline__ void systolic_strip(const void* __restrict__ in, int64_t x0, int strip, int count, const float* __restrict__ taps,
                                               float* __restrict__ out, float* __restrict__ wbuf, int lane, const SystolicFix& fx)
{
    constexpr int kOuts = Sys<PT>::kOuts, kStep = Sys<PT>::kStep, kWarm = Sys<PT>::kWarm;
    const int seamK = fx.seamK, k0mod = fx.k0mod, nx = fx.nx;
    f2 S[32];
    const int64_t strip_s0 = x0 + (int64_t)kStep * strip;
    // samples of the launch: (count - 1) * 8 + lp from x0 on
    const int64_t avail = WHOLE ? kStripSpan : ((int64_t)(count - 1) * 8 + fx.lp) - (int64_t)kStep * strip;
    load_strip<U8, WHOLE, NTL>(in, strip_s0, avail, wbuf, lane, S);

    f2 acc[4][4];
    f8 tc[PT];
    tc[0] = load_tap_row(taps, 0);
    auto do_m = [&](auto mc) {
        constexpr int m = decltype(mc)::value, t = m >> 2, c = m & 3;
        if constexpr (m + 1 < PT) tc[m + 1] = load_tap_row(taps, m + 1);
        else asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = r & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = (m - i) & (PT - 1);                  // tap row of output i (only used when m - i is in 0 .. PT - 1)
                if (m - i < 0 || m - i > PT - 1) continue;
                if (PSKIP && 8 * b + r >= 8 * PT - PSKIP) {
                    // a skipped tap at a stage's entry: the partial sum still has to move up a lane (only the 64-tap instantiation
                    // skips that far into a row: taps 56 .. 59 of output 1 meet m = 8)
                    if (t > 0 && c == 0 && r < 4) acc[i][k] = f2{dpp_shr1(acc[i][k].x), dpp_shr1(acc[i][k].y)};
                    continue;
                }
                const f2 p = S[8 * c + r] * tc[b][r];
                if (b == 0 && r < 4) {
                    acc[i][k] = f2{0.f, 0.f} + p;                  // the first addition of the partial: +0 + product
                } else if (t > 0 && c == 0 && r < 4) {
                    acc[i][k] = f2{dpp_shr1(acc[i][k].x) + p.x, dpp_shr1(acc[i][k].y) + p.y};   // the group enters the stage one lane up
                } else {
                    acc[i][k] = acc[i][k] + p;
                }
            }
        }
    };
    for_each_m(std::make_integer_sequence<int, Sys<PT>::kM>{}, do_m);

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
    // FIX: position of the strip's first output inside its buffer (wave-uniform: scalar arithmetic), and whether any of its 240 (248)
    // outputs is one of the buffer's last nx (seamK >= 256 > 248: the strip wraps the buffer grid at most once)
    int km_strip = 0;
    bool strip_cross = false;
    if constexpr (FIX) {
        const int su = __builtin_amdgcn_readfirstlane(strip);
        km_strip = (int)(((unsigned)k0mod + (unsigned)((kOuts * (int64_t)su) % seamK)) % (unsigned)seamK);
        strip_cross = km_strip + (kOuts - 1) >= seamK - nx;
    }
    if (lane >= kWarm) {
        const int o = kOuts * strip + 4 * (lane - kWarm);
        bool cross[4] = {false, false, false, false};
        bool any_cross = false;
        if constexpr (FIX) {
            if (strip_cross) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int km = km_strip + 4 * (lane - kWarm) + i;
                    if (km >= seamK) km -= seamK;
                    cross[i] = km >= seamK - nx;
                    any_cross |= cross[i];
                }
            }
        }
        if ((WHOLE || o + 4 <= count) && !any_cross) {
            float4* dst = reinterpret_cast<float4*>(out + 2 * (int64_t)o);
            dst[0] = make_float4(res[0].x, res[0].y, res[1].x, res[1].y);
            dst[1] = make_float4(res[2].x, res[2].y, res[3].x, res[3].y);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if ((WHOLE || o + i < count) && !cross[i]) *reinterpret_cast<float2*>(out + 2 * (int64_t)(o + i)) = make_float2(res[i].x, res[i].y);
        }
    }
}

template <bool U8, int PSKIP, bool NTL = true, bool FIX = false, int PT = 16>
__global__ void __launch_bounds__(64 * kWavesPerWg, 4) k_decimate_systolic(const void* __restrict__ in, int64_t x0 /* sample of output 0's window in `in` */,
                                                                          int count, const float* __restrict__ taps, float* __restrict__ out,
                                                                          int nwhole /* strips [0, nwhole) are whole */, int nstrips, SystolicFix fx)
{
    constexpr int kStripLds = U8 ? 4 : kWavesPerWg * kCfWaveDw;
    __shared__ __attribute__((aligned(16))) float tbuf[FIX && kFixLdsFloats > kStripLds ? kFixLdsFloats : kStripLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b = blockIdx.x;
    if constexpr (FIX) {
        const int G = b >> 6, q = G / fx.period_g;
        if (q < fx.nfix_groups && G - q * fx.period_g == fx.period_g - 1) {
            decimate_c_crossfix_wg<U8, 8, (PT == 16 ? 128 : 52), kFixPer, kFixSpw>(reinterpret_cast<float2*>(tbuf), 64 * q + (b & 63), fx.g, fx.xtaps, in, out, fx.first_seam, fx.nseams);
            return;
        }
        b -= 64 * (q < fx.nfix_groups ? q : fx.nfix_groups);          // fix-up blocks in front of this group
    }
    // XCD-aware order (the tile kernel's): workgroup b runs on XCD b % 8; within every 64 consecutive workgroups XCD x takes 8
    // consecutive ones, so 7 of 8 strip-to-strip overlaps of a workgroup's neighbours hit in the same L2
    const int wg = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    const int strip = wg * kWavesPerWg + wave;
    if (strip >= nstrips) return;
    float* wbuf = tbuf + (U8 ? 0 : kCfWaveDw * wave);
    if (strip < nwhole) systolic_strip<U8, PSKIP, true, NTL, FIX, PT>(in, x0, strip, count, taps, out, wbuf, lane, fx);
    else systolic_strip<U8, PSKIP, false, NTL, FIX, PT>(in, x0, strip, count, taps, out, wbuf, lane, fx);
}

std::atomic<int>& systolic_flag()
{
    static std::atomic<int> f{getenv("SDRHIP_SYSTOLIC") ? atoi(getenv("SDRHIP_SYSTOLIC")) : 2};
    return f;
}
std::atomic<long long> g_systolic_launches{0};

}  // namespace

// How a launch of `count` outputs is cut into wave-strips (host arithmetic, testable without a GPU: sdrhip_debug_systolic_plan).
// Strip t covers outputs 240 t .. 240 t + 239 and reads samples 1920 t .. 1920 t + 2047.  Strips [0, nwhole) have all their
// outputs wanted and all their samples inside the launch's (count - 1) * 8 + 128.
static void systolic_plan_for(int count, int outs, int lp, int* nstrips, int* nwhole)
{
    *nstrips = (count + outs - 1) / outs;
    int w = count / outs;
    while (w > 0 && (int64_t)8 * outs * (w - 1) + kStripSpan > (int64_t)(count - 1) * 8 + lp) w--;
    *nwhole = w;
}
void systolic_plan(int count, int* nstrips, int* nwhole) { systolic_plan_for(count, kStripOuts, 128, nstrips, nwhole); }

void set_systolic(int mode) { systolic_flag().store(mode); }
long long systolic_launch_count() { return g_systolic_launches.load(); }

// The SIMD ("One") outputs of a decimate-by-8, 128-tap, AVX-order launch.  False = not this kernel's shape or too small to be
// worth it (the tile kernel computes its Cross outputs in place for launch-bound sizes); the caller then takes the tile kernel.
//
// cfloat input, plain or non-temporal loads (round 6, tools/route_sweep_fine.py; profiles/r06/route_sweep_fine*.txt): the launch reads
// 64 B per output.  Between ~64 MB and ~285 MB the input of a back-to-back caller -- or of a pipeline whose producer has just written
// it -- sits in the 256 MB Infinity Cache: plain loads are served from there (B = 2048 blocks: 33.8 us against 41.2 with
// non-temporal loads and 39.0 for the tile kernel), a non-temporal load of a cached line is the slow case.  Past that size the
// stream comes from HBM whatever the hint says and the non-temporal form wins by 2-4 % (it does not evict the taps and the outputs'
// lines); below it both forms are within noise of each other and of launch latency.
constexpr int64_t kPlainLoadMinOutputs = (int64_t)1000 * 1024;      // ~64 MB of cfloat input
constexpr int64_t kPlainLoadMaxOutputs = (int64_t)4400 * 1024;      // ~285 MB
constexpr int64_t kFixInsideMaxOutputs = (int64_t)1 << 23;          // the seam fix-up's workgroups inside the launch up to this size

bool launch_decimate_c4_systolic(hipStream_t s, const Geom& g, const float* d_taps, int P, const void* d_in, bool in_is_u8, float* d_out,
                                 bool last_tap_zero, const float* d_cross_taps, bool* seams_done)
{
    if (seams_done) *seams_done = false;
    const int mode = systolic_flag().load(std::memory_order_relaxed);
    if (mode == 0) return false;
    // 128 taps (either input), or -- u8 input, the library's own choice only -- the reference example's 52-tap filter on the 64-tap
    // instantiation (12 taps of padding skipped; cfloat input and other lengths stay on the tile kernel)
    const bool p52 = in_is_u8 && mode == 2 && P == 52 && g.Lp == 52;
    if (g.I != 1 || g.D != 8 || !((P == 128 && g.Lp == 128) || p52) || g.count < 64 * kStripOuts * kWavesPerWg) return false;
    const int64_t x0 = g.k_begin * g.D - g.in_base;
    const uintptr_t base = reinterpret_cast<uintptr_t>(d_in);
    if (((base + (in_is_u8 ? 2 : 8) * (uintptr_t)x0) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return false;
    // strip n is whole when its 240 outputs are wanted and its 2048 samples exist: 1920 n + 2048 <= (count - 1) * 8 + 128
    int nstrips, nwhole;
    systolic_plan_for(g.count, p52 ? Sys<8>::kOuts : Sys<16>::kOuts, g.Lp, &nstrips, &nwhole);
    const int nwg = (nstrips + kWavesPerWg - 1) / kWavesPerWg;
    int groups = (nwg + 63) / 64;
    // the launch's own seams (those strictly inside its samples), as the fix-up launch of kernels_fast.hip counts them
    SystolicFix fx = {};
    fx.lp = g.Lp;
    fx.nx = (g.Lp - 1) / 8;
    bool fix = false;
    // Inside the launch up to 2^23 outputs (2^26 samples): there the fix-up is latency -- a second launch, 10-15 us of dependent chains
    // on an idle chip -- and hiding it gains 3-30 % (2^22 ... 2^25 samples: 15.6 -> 10.8, 24.3 -> 18.0, 34.1 -> 28.6, 59.7 -> 57.9 us
    // cfloat; u8 13.2 -> 9.8 ... 45.9 -> 42.3).  Past that the chip is issue- and power-bound for the whole launch and the fix-up's
    // instructions cost the same wherever they run, plus what they disturb: 1.5-4 % SLOWER inside at 2^27 ... 2^29 samples
    // (profiles/r06/k2_fix_inside_ab_sizes.txt) -- those launches keep the second launch.
    if (mode == 2 && g.count <= kFixInsideMaxOutputs && seams_done && g.seamBI > 0 && g.seamBI % 8 == 0 && g.seamBI / 8 >= 256 && g.seamBI / 8 < (1 << 30) && d_cross_taps != nullptr) {
        const int64_t v_lo = g.k_begin * g.D, v_hi = (g.k_begin + g.count - 1) * g.D + g.Lp;
        const int64_t first = v_lo / g.seamBI + 1, last = (v_hi - 1) / g.seamBI;
        if (last >= first && last - first + 1 < (int64_t)1 << 30) {
            fx.g = g;
            fx.xtaps = d_cross_taps;
            fx.first_seam = first;
            fx.nseams = (int)(last - first + 1);
            fx.nfix_groups = ((fx.nseams + kFixSpw - 1) / kFixSpw + 63) / 64;
            fx.period_g = (groups + fx.nfix_groups) / fx.nfix_groups;
            fx.seamK = (int)(g.seamBI / 8);
            fx.k0mod = (int)(g.k_begin % fx.seamK);
            fix = fx.period_g >= 2;
        } else if (last < first) {
            *seams_done = true;          // no boundary inside the launch: nothing to fix up
        }
    }
    if (fix) {
        groups += fx.nfix_groups;
        *seams_done = true;
    }
    const dim3 grid(groups * 64), block(64 * kWavesPerWg);
    const bool plain = !in_is_u8 && mode == 2 && g.count >= kPlainLoadMinOutputs && g.count <= kPlainLoadMaxOutputs;
#define SYS(U8V, PSKIPV, NTLV)                                                                                                                      \
    do {                                                                                                                                            \
        if (fix) hipLaunchKernelGGL((k_decimate_systolic<U8V, PSKIPV, NTLV, true>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, fx); \
        else hipLaunchKernelGGL((k_decimate_systolic<U8V, PSKIPV, NTLV, false>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, fx); \
    } while (0)
    if (p52) {
        if (fix) hipLaunchKernelGGL((k_decimate_systolic<true, 12, true, true, 8>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, fx);
        else hipLaunchKernelGGL((k_decimate_systolic<true, 12, true, false, 8>), grid, block, 0, s, d_in, x0, g.count, d_taps, d_out, nwhole, nstrips, fx);
    } else if (in_is_u8 && last_tap_zero) SYS(true, 1, true);
    else if (in_is_u8) SYS(true, 0, true);
    else if (plain) SYS(false, 0, false);
    else SYS(false, 0, true);
#undef SYS
    g_systolic_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

}  // namespace sdrhip
