// kernels_resample_stream.hip -- round 5: fmDemod (Demod.hs:21-46) + the 3/10 polyphase resampler (resampleAVXRR,
// resample.c:70-87) as a STREAMING kernel: a workgroup walks a contiguous run of tiles and the complex samples of tile
// T + 1 are in flight (registers, 16-byte loads) while tile T is demodulated and resampled.
//
// Why (VERDICT r04 "weak" 3): in k_resample3_fast<.., DEMOD> (kernels_chain.hip) a workgroup loads, waits, computes and exits;
// with both of its roofs equally near (~0.10 ms of VALU, ~0.11 ms of HBM traffic per 2^26 inputs) the two only overlap across
// workgroups, and did so poorly (VALU 55-59 % busy, SQ_WAIT_ANY 1.27 x active).  Here every wave carries its own loads across
// its own arithmetic:
//   * a tile is kNT = 256 polyphase cycles: 768 outputs from inputs [2560 T, 2560 T + 2621) of the launch;
//   * the 61 inputs a tile shares with its successor are demodulated ONCE: the phases y live in a two-segment LDS ring and
//     the last 64 of a segment are copied to the head of the other one (after the barrier, from registers), so a tile
//     demodulates exactly 2560 new samples = 5 aligned pairs per thread -- no ragged sixth round, no 2.4 % redone;
//   * pairs are 16-byte loads (global_load_dwordx4), a quarter of the 8-byte load instructions of the tile kernel, which
//     read every sample twice (itself and its predecessor): the predecessor of a pair's first sample is the second sample of
//     the lane below (DPP wave_shr:1), and lane 0 of a wave takes it from a wave-uniform 8-byte load;
//   * one LDS-only barrier per tile (s_waitcnt lgkmcnt(0); s_barrier): the loads in flight stay in flight across it.
// Arithmetic: fm_phase_common_tbl + wave vote + fm_phase_sel (demod.hpp) and the packed-pair walk of k_resample3_fast --
// the same operations in the same order, so the same bits (tests/test_gpu_chain.py::test_chain_demod_fusion_is_invisible,
// tests/test_gpu_stream.py, tests/test_gpu_resample_stream.py).
//
// MEASURED (round 5, LABNOTES.md): both variants in this file -- prefetch in registers (three waves per SIMD) and in LDS by
// global_load_lds (four) -- take 0.233-0.236 ms per 2^26 inputs where the tile kernel takes 0.189-0.198.  The loads are hidden (the
// ablation shows it); what loses is the lock step of demodulator and resampler phases inside a persistent workgroup.  Kept, switched
// off (sdrhip_debug_set_resample_demod_stream), under test.
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "lab.hpp"
#include "demod_forms.hpp"

#ifndef SDRHIP_RSTREAM_MINW
#define SDRHIP_RSTREAM_MINW 3          // waves per SIMD the register budget is set for (= workgroups per CU)
#endif
#ifndef SDRHIP_RSTREAM_BATCH
#define SDRHIP_RSTREAM_BATCH 0        // register pairs of the window per batch of the resampler's walk (36 pairs in all)
#endif
#ifndef SDRHIP_RSTREAM_ILP
#define SDRHIP_RSTREAM_ILP 2
#endif
#ifndef SDRHIP_RSTREAM_PAIR
#define SDRHIP_RSTREAM_PAIR 0
#endif
#ifndef SDRHIP_RSTREAM_DEBUG
#define SDRHIP_RSTREAM_DEBUG 0        // measurements only (wrong results): 1 no fmDemod arithmetic, 2 no global loads after the first tile, 4 no resampler walk
#endif
#ifndef SDRHIP_RSTREAM_WGS
#define SDRHIP_RSTREAM_WGS SDRHIP_RSTREAM_MINW   // workgroups per CU the grid is sized for
#endif

namespace sdrhip {

namespace {

constexpr int kNT = 256;                       // threads per workgroup = polyphase cycles per tile
constexpr int kTileIn = 10 * kNT;              // new inputs per tile
constexpr int kCarry = 64;                     // inputs a tile's windows reach past its own 2560 (61), in whole pairs
constexpr int kSeg = kTileIn + kCarry;         // floats per LDS segment
constexpr int kRounds = kTileIn / (2 * kNT);   // pairs per thread and tile

// (the four arrays are kernel parameters of their own: `__restrict__` on a parameter is what lets the compiler read the taps with
// scalar loads -- as members of this struct they came in through 48 vector registers)
struct StreamArgs {
    int64_t origin;        // sample index of segment position 0 of tile 0: the first cycle's window starts at origin + e; may be -1
    int ntiles, ncycles, tiles_per_wg, has_prev;
    int64_t y_count;
    int row_stride;
    int64_t y_abs0;        // absolute stream index of sample 0
    int yseam, ykeep, nedge;
};

// wave_shr:1 -- lane l takes lane l - 1's `src`; lane 0, which has no source, keeps `old`
__device__ __forceinline__ float dpp_shr1_or(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf,
                                                                 0xf, false));
}

__device__ __forceinline__ void lds_barrier()
{
    // __syncthreads() would also drain vmcnt: the next tile's loads are meant to stay in flight across this
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The packed-pair walk of k_resample3_fast (kernels_chain.hip) over a window that starts E floats into an 8-byte aligned row:
// 8 lane partials per output as four v_pk_mul_f32 (SGPR tap pair) + v_pk_add_f32 pairs; a group whose first sample is the high
// half of its register pair pairs the partials (1,2)(3,4)(5,6)(7,0), taps 0 and 63 single.  Every partial adds its products in
// increasing tap order from +0 and the tree is ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)) (avx_dotprod_R / avx_hadd_R, common.h:18-29,58-72).
template <int E>
__device__ __forceinline__ void resample3_window(const float* __restrict__ row, const float* __restrict__ groups, int row_stride,
                                                 float (&res)[3])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NLOOP = 64;
    constexpr int PRE[3] = {E, 4 + E, 7 + E};
    constexpr int NPAIR = (PRE[2] + NLOOP + 1) / 2;
#if SDRHIP_RSTREAM_BATCH == 0
    // group after group over the whole window in registers (the walk of k_resample3_fast)
    f2 W2[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; i++) {
        const float2 q = *reinterpret_cast<const float2*>(row + 2 * i);
        W2[i] = f2{q.x, q.y};
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float* c = groups + g * row_stride;
        float acc[8];
        if (PRE[g] % 2 == 0) {
            f2 A[4];
#pragma unroll
            for (int p = 0; p < 4; p++) A[p] = f2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NLOOP; j += 2) A[(j % 8) / 2] = A[(j % 8) / 2] + W2[(PRE[g] + j) / 2] * f2{c[j], c[j + 1]};
#pragma unroll
            for (int p = 0; p < 4; p++) { acc[2 * p] = A[p].x; acc[2 * p + 1] = A[p].y; }
        } else {
            f2 B[4];                                                                   // B[p] = (partial 2p + 1, partial 2p + 2 mod 8)
#pragma unroll
            for (int p = 0; p < 4; p++) B[p] = f2{0.0f, 0.0f};
            B[3].y = 0.0f + c[0] * W2[PRE[g] / 2].y;
#pragma unroll
            for (int j = 1; j + 1 < NLOOP; j += 2) B[((j % 8) - 1) / 2] = B[((j % 8) - 1) / 2] + W2[(PRE[g] + j) / 2] * f2{c[j], c[j + 1]};
            B[3].x = B[3].x + c[NLOOP - 1] * W2[(PRE[g] + NLOOP - 1) / 2].x;
            acc[0] = B[3].y; acc[7] = B[3].x;
#pragma unroll
            for (int p = 0; p < 3; p++) { acc[2 * p + 1] = B[p].x; acc[2 * p + 2] = B[p].y; }
        }
        res[g] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
#else
    // ONE pass over the row's register pairs, the three groups side by side: a pair is read when the first group needs it and is
    // dead when the last one has used it (seven pairs later), so the walk holds 24 partials and a handful of pairs instead of the
    // whole 72-float window -- the registers that buys carry the next tile's samples.  Each partial still sees its own products in
    // increasing tap order.
    f2 P[3][4];
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int p = 0; p < 4; p++) P[g][p] = f2{0.0f, 0.0f};
    // batches of NB pairs, the next batch's LDS reads issued before the current batch's arithmetic and nothing allowed across the
    // batch boundaries (left alone the scheduler hoists all 36 reads to the top: the whole window live again)
    constexpr int NB = SDRHIP_RSTREAM_BATCH;
    static_assert(NPAIR % NB == 0, "whole batches");
    f2 W[2][NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const float2 q = *reinterpret_cast<const float2*>(row + 2 * i);
        W[0][i] = f2{q.x, q.y};
    }
#pragma unroll
    for (int b = 0; b < NPAIR / NB; b++) {
        if (b + 1 < NPAIR / NB) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const float2 q = *reinterpret_cast<const float2*>(row + 2 * ((b + 1) * NB + i));
                W[(b + 1) & 1][i] = f2{q.x, q.y};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int k = b * NB + i;
            const f2 w = W[b & 1][i];
#pragma unroll
            for (int g = 0; g < 3; g++) {
                const float* c = groups + g * row_stride;
                const int j = 2 * k - PRE[g];                          // the tap the pair's low half meets
                if (PRE[g] % 2 == 0) {
                    // P[g][p] = (partial 2p, partial 2p + 1)
                    if (j >= 0 && j < NLOOP) P[g][(j % 8) / 2] = P[g][(j % 8) / 2] + w * f2{c[j], c[j + 1]};
                } else {
                    // P[g][p] = (partial 2p + 1, partial 2p + 2 mod 8): P[g][3] = (partial 7, partial 0); taps 0 and 63 are single
                    if (j == -1) P[g][3].y = 0.0f + c[0] * w.y;
                    else if (j == NLOOP - 1) P[g][3].x = P[g][3].x + c[NLOOP - 1] * w.x;
                    else if (j >= 1 && j < NLOOP - 1) P[g][((j % 8) - 1) / 2] = P[g][((j % 8) - 1) / 2] + w * f2{c[j], c[j + 1]};
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        float acc[8];
        if (PRE[g] % 2 == 0) {
#pragma unroll
            for (int p = 0; p < 4; p++) { acc[2 * p] = P[g][p].x; acc[2 * p + 1] = P[g][p].y; }
        } else {
            acc[0] = P[g][3].y; acc[7] = P[g][3].x;
#pragma unroll
            for (int p = 0; p < 3; p++) { acc[2 * p + 1] = P[g][p].x; acc[2 * p + 2] = P[g][p].y; }
        }
        res[g] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
#endif
}

__device__ __forceinline__ bool keep_pos(int m, int yseam, int ykeep) { return m < ykeep || m >= yseam - ykeep; }

// One guarded tile (the launch's first when nothing precedes it, its last ones): every position of the segment from scratch, one
// sample at a time, the full form of fmDemod; positions without a sample hold 0.  Never runs with loads of the streaming path in flight.
__device__ __forceinline__ void guarded_tile(const StreamArgs& a, const float* __restrict__ in, float* __restrict__ y_out, int64_t tile0, float* sg,
                                             int tid)
{
    const float2* z = reinterpret_cast<const float2*>(in);
    for (int L = tid; L < kSeg; L += kNT) {
        const int64_t n = tile0 + L;
        float yv = 0.0f;
        if (n >= 0 && n < a.y_count) {
            const float2 c = z[n];
            const float2 pv = (n > 0 || a.has_prev) ? z[n - 1] : make_float2(0.0f, 0.0f);
            yv = fm_phase_sel(c, pv);
            if (a.yseam > 0 && keep_pos((int)((a.y_abs0 + n) % a.yseam), a.yseam, a.ykeep)) y_out[n] = yv;
        }
        sg[L] = yv;
    }
}

// FULL: every cycle of the tile exists -- the store is then unconditional, which is what lets the compiler count it: vmcnt returns in
// order, and behind a store that MAY have been issued the wait for the next tile's samples becomes a wait for the store's acknowledgement
template <int E, bool FULL>
__device__ __forceinline__ void resample_tile(const StreamArgs& a, const float* __restrict__ groups, float* __restrict__ out, int T, const float* sg,
                                              int tid)
{
    const int cyc0 = T * kNT;
    if (FULL || cyc0 + tid < a.ncycles) {
        float res[3];
        if (SDRHIP_RSTREAM_DEBUG & 4) {
            res[0] = sg[tid * 10]; res[1] = sg[tid * 10 + 4]; res[2] = sg[tid * 10 + 7];
        } else
        resample3_window<E>(sg + tid * 10, groups, a.row_stride, res);
        struct __attribute__((packed, aligned(4))) f3 { float a, b, c; };
        const f3 v = {res[0], res[1], res[2]};
        float* o = out + (int64_t)cyc0 * 3;                                   // wave-uniform base, 32-bit lane offset
        *reinterpret_cast<f3*>(reinterpret_cast<char*>(o) + (unsigned)tid * 12u) = v;
    }
}

template <int E>
__global__ void __launch_bounds__(kNT, SDRHIP_RSTREAM_MINW) k_resample3_demod_stream(const float* __restrict__ in, const float* __restrict__ groups,
                                                                                          float* __restrict__ out, float* __restrict__ y_out, StreamArgs a)
{
    // in: decimator output, sample n at in[2n], in[2n + 1], n in [0, y_count) (and n = -1 when has_prev); groups: the three tap rows;
    // out[3 c .. 3 c + 2]: the outputs of cycle c; y_out[n]: the phases other kernels still read (seam fix-up, lead / tail launches)
    __shared__ __attribute__((aligned(16))) float seg[2][kSeg];
    __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
    const int tid = threadIdx.x;
    const int wave_t0 = __builtin_amdgcn_readfirstlane(tid) & ~63;            // first thread of this wave: wave-uniform
    atan_table_fill(atbl, tid);

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = t_begin + a.tiles_per_wg < a.ntiles ? t_begin + a.tiles_per_wg : a.ntiles;
    const float2* __restrict__ z = reinterpret_cast<const float2*>(in);
    // a tile of 256 whole cycles whose 2560 new samples (and the pair slack behind them) all exist takes the streaming path
    auto fast = [&](int T) { return T < t_end && (T + 1) * kNT <= a.ncycles && a.origin + (int64_t)T * kTileIn + kSeg <= a.y_count; };
    // position of a tile's first NEW sample (segment position kCarry) in the seam grid
    int m_new = 0;
    if (a.yseam > 0) m_new = (int)((a.y_abs0 + a.origin + (int64_t)t_begin * kTileIn + kCarry) % a.yseam);
    auto advance = [&]() {
        if (a.yseam > 0) {
            m_new += kTileIn;
            if (m_new >= a.yseam) m_new -= a.yseam;
        }
    };
    int T = t_begin, s = 0;
    lds_barrier();                                                             // the table
    // the first tile of a run has no carried phases: its 64 head positions are demodulated in the streaming path's prologue when
    // their samples (and the predecessor of the first) exist; otherwise the whole tile goes the guarded way
    bool carried = false;
    {
        const int64_t tile0 = a.origin + (int64_t)T * kTileIn;
        if (T < t_end && !(tile0 >= 1 || (tile0 == 0 && a.has_prev))) {
            guarded_tile(a, in, y_out, tile0, seg[s], tid);
            lds_barrier();
            if (tid < kCarry / 2) reinterpret_cast<float2*>(seg[s ^ 1])[tid] = reinterpret_cast<const float2*>(seg[s])[kTileIn / 2 + tid];
            resample_tile<E, false>(a, groups, out, T, seg[s], tid);
            advance();
            T++, s ^= 1, carried = true;
        }
    }

    // ---- the streaming path ----
    // a wave owns kRounds * 64 consecutive pairs of the tile (round i = pairs [64 i, 64 i + 64) of them): the predecessor of a
    // pair's first sample is the second sample of the lane below, or, for lane 0, of lane 63 in the round before -- only round 0
    // needs a load of its own (the sample before the wave's first pair: a wave-uniform address)
    const int lp0 = kRounds * wave_t0 + (tid & 63);                            // this thread's pair in round 0, relative to the tile's first new pair
    const unsigned off16 = (unsigned)lp0 * 16u, off8 = (unsigned)lp0 * 8u;
    // `live` false (no next tile): the same instructions on one valid 16-byte address for every lane.  The loads must not sit in a
    // branch: the compiler's wait-count pass merges the two paths at the join, takes the shorter queue (nothing issued) and makes the
    // demodulator wait for ALL BUT FOUR of whatever is outstanding -- the loads just issued -- instead of for last tile's (measured:
    // 0.25 ms per 2^26 inputs with `if (more) issue(..)`, the whole HBM latency exposed once per tile)
    auto issue = [&](int Tn, bool live, float4 (&v)[kRounds], float2& p) {
        const int64_t n0 = a.origin + (int64_t)Tn * kTileIn + kCarry;         // an even offset from `origin`: 16-byte aligned
        const char* src = reinterpret_cast<const char*>(in + 2 * n0);          // wave-uniform base, 32-bit lane offsets
        const unsigned o = live ? off16 : 0u;
        const unsigned step = live ? 1024u : 0u;
#pragma unroll
        for (int i = 0; i < kRounds; i++) v[i] = *reinterpret_cast<const float4*>(src + (o + step * i));
        // the wave's first predecessor: a wave-uniform address, but it must NOT become a scalar load -- SMEM returns out of order,
        // so with an s_load of HBM data outstanding every LDS wait of the demodulator turns into lgkmcnt(0) and sits out the whole
        // memory latency; the offset goes through a VGPR the compiler cannot see through and the load counts on vmcnt like the rest
        unsigned po = live ? (unsigned)(2 * kRounds * wave_t0) * 8u : 0u;
        asm volatile("" : "+v"(po));
        p = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(z + (n0 - 1)) + po);
    };
    // one tile: demodulate `cur` (its wave's first predecessor in `pl`), with the next tile's samples on their way into `nxt` / `npl`.
    // Called with the two register sets swapped for alternate tiles (a `cur = nxt` at the end of a single loop body costs twenty
    // v_mov and, worse, a wait for the loads just issued at the very place they are meant to be in flight)
    auto tile = [&](const float4 (&cur)[kRounds], const float2& pl, float4 (&nxt)[kRounds], float2& npl) {
        const int64_t tile0 = a.origin + (int64_t)T * kTileIn;                // sample index of segment position 0
        float2* seg2 = reinterpret_cast<float2*>(seg[s]);
        const bool more = fast(T + 1);
        if (SDRHIP_RSTREAM_DEBUG & 2) {
            npl = pl;
#pragma unroll
            for (int i = 0; i < kRounds; i++) nxt[i] = cur[i];
        } else
        issue(more ? T + 1 : T, more, nxt, npl);
        if (!carried && tid < kCarry / 2) {
            const int64_t n = tile0 + 2 * tid;
            const float2 pv = z[n - 1], A = z[n], B = z[n + 1];
            const float2 y = make_float2(fm_phase_sel(A, pv), fm_phase_sel(B, A));
            seg2[tid] = y;
            if (a.yseam > 0) {
                const int m0 = (int)((a.y_abs0 + n) % a.yseam), m1 = m0 + 1 == a.yseam ? 0 : m0 + 1;
                if (keep_pos(m0, a.yseam, a.ykeep)) y_out[n] = y.x;
                if (keep_pos(m1, a.yseam, a.ykeep)) y_out[n + 1] = y.y;
            }
        }
        carried = true;
        // the sample before pair i's first: lane l - 1's second sample; lane 0: lane 63's of the round before, or the loaded one
        auto pred = [&](int i) {
            float2 o = pl;
            if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].z), 63)),
                                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].w), 63)));
            return make_float2(dpp_shr1_or(o.x, cur[i].z), dpp_shr1_or(o.y, cur[i].w));
        };
        float2 y[kRounds];
        bool rare = false;
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
            const float2 pv = pred(i);
            bool q0;
            if (SDRHIP_RSTREAM_DEBUG & 1) {
                y[i] = make_float2(A.x + pv.y, B.x + A.y);
                continue;
            }
#if SDRHIP_RSTREAM_PAIR
            // the pair as ONE packed evaluation (demod.hpp: fm_phase_common_tbl2): measured slower, see SDRHIP_LOADER_PAIR in kernels_chain.hip
            y[i] = fm_phase_common_tbl2(A, pv, B, A, q0, atbl);
            rare |= q0;
            if (SDRHIP_RSTREAM_ILP <= 2 || (2 * (i + 1)) % SDRHIP_RSTREAM_ILP == 0) __builtin_amdgcn_sched_barrier(0);
#else
            bool q1;
            y[i].x = fm_phase_common_tbl(A, pv, q0, atbl);
            if (SDRHIP_RSTREAM_ILP == 1) __builtin_amdgcn_sched_barrier(0);
            y[i].y = fm_phase_common_tbl(B, A, q1, atbl);
            rare |= q0 | q1;
            // SDRHIP_RSTREAM_ILP samples' dependent chains interleaved (1: sample after sample; 2: a pair; 0: the scheduler's choice)
            if (SDRHIP_RSTREAM_ILP == 1 || (SDRHIP_RSTREAM_ILP >= 2 && (2 * (i + 1)) % SDRHIP_RSTREAM_ILP == 0)) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // stored before the vote: with the phases needed only after it, the compiler moves most of the arithmetic behind the
        // branch and keeps the samples' lane masks alive across it
#pragma unroll
        for (int i = 0; i < kRounds; i++) seg2[kCarry / 2 + lp0 + 64 * i] = y[i];
        if (__any(rare)) {
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
                const float2 pv = pred(i);
                y[i] = make_float2(fm_phase_sel(A, pv), fm_phase_sel(B, A));
                seg2[kCarry / 2 + lp0 + 64 * i] = y[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the phases other kernels still read: only a tile that touches a keep zone looks at positions at all
        if (a.yseam > 0 && (m_new < a.ykeep || m_new + kTileIn > a.yseam - a.ykeep)) {
            char* yb = reinterpret_cast<char*>(y_out + tile0 + kCarry);       // wave-uniform base, 32-bit lane offsets
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                int m0 = m_new + 2 * (lp0 + 64 * i);
                if (m0 >= a.yseam) m0 -= a.yseam;
                const int m1 = m0 + 1 == a.yseam ? 0 : m0 + 1;
                float* yo = reinterpret_cast<float*>(yb + (off8 + 512u * i));
                if (keep_pos(m0, a.yseam, a.ykeep)) yo[0] = y[i].x;
                if (keep_pos(m1, a.yseam, a.ykeep)) yo[1] = y[i].y;
            }
        }
        const float2 ylast = y[kRounds - 1];
        lds_barrier();
        // the segment's last 64 positions are the next tile's first 64 (nobody reads the other segment before the next barrier,
        // and everybody has left it: they all passed this one): the tile's last 32 pairs, the upper half of the last wave's last round
        if (tid >= kNT - kCarry / 2) reinterpret_cast<float2*>(seg[s ^ 1])[tid - (kNT - kCarry / 2)] = ylast;
        resample_tile<E, true>(a, groups, out, T, seg[s], tid);
        advance();
        T++, s ^= 1;
    };
    float4 ra[kRounds], rb[kRounds];
    float2 pa, pb;
    // the run's first streaming tile on its own: the loop below is then always entered from the end of a tile, with the same
    // queue of outstanding loads and stores as on its back edge (the wait-count pass merges the two and keeps the shorter)
    if (fast(T)) {
        issue(T, true, ra, pa);
        tile(ra, pa, rb, pb);
    }
    while (fast(T)) {
        tile(rb, pb, ra, pa);
        if (!fast(T)) break;
        tile(ra, pa, rb, pb);
    }
    // ---- the run's last tiles, when their samples end before a whole tile does ----
    for (; T < t_end; T++, s ^= 1) {
        guarded_tile(a, in, y_out, a.origin + (int64_t)T * kTileIn, seg[s], tid);
        lds_barrier();
        resample_tile<E, false>(a, groups, out, T, seg[s], tid);
        lds_barrier();                 // (the next guarded tile rewrites this segment's neighbour only, but keep the cold path simple)
    }
    // the launch's two edges, y[0, nedge) and y[y_count - nedge, y_count), for the lead / tail launches of the generic kernel
    if (a.nedge > 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        for (int sg = 0; sg < 2; sg++) {
            if (blockIdx.x != (sg == 0 ? 0u : gridDim.x - 1)) continue;
            const int64_t p0 = sg == 0 ? 0 : a.y_count - a.nedge;
            for (int i = tid; i < a.nedge; i += kNT) {
                const int64_t p = p0 + i;
                const float2 v[2] = {(p > 0 || a.has_prev) ? z[p - 1] : make_float2(0.0f, 0.0f), z[p]};
                float y[1];
                fm_phase_voted<1>(v, y);
                y_out[p] = y[0];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same streaming kernel with the prefetch in LDS instead of registers (VERDICT r04 "next" 1 named it: global_load_lds_dwordx4).
// Every wave owns kRounds slots of 1 KiB (one round of 64 pairs each) and one 16-byte slot for the pair in front of its first one
// (the predecessor of its first sample).  A slot is refilled with the NEXT tile's round as soon as the current tile's round has
// been read out of it, so the loads are in flight for a whole tile and cost no VGPR: 128 registers suffice, four workgroups per CU
// (an even four waves per SIMD: tools/k4lab/issue_bench.hip).  The phases y have ONE segment here (LDS: 20.5 KB of slots + 10.5 KB
// + the table = 33 KB per workgroup), hence two barriers per tile: D | R | next D.
// The compiler knows nothing about LDS-DMA: the waits are counted by hand.  vmcnt returns in order, so "at most N outstanding" means
// everything but the newest N is done; N = the MINIMUM number of operations issued after the one waited for (conditional stores count
// as absent: if they were issued the wait is merely longer).  In a tile's round i the wave needs DMA(T, i), issued in round i of tile
// T - 1; after it came DMA(T, i+1 .. 4), the output store of T - 1, DMA(T+1, pred), DMA(T+1, 0 .. i-1): 6 operations (round 0: the
// predecessor slot too, issued just before DMA(T, 0): 5).  The run's first streaming tile has no output store before it: 5 / 4.
// The refills are issued unconditionally (on the run's last tile: to a harmless address) so that those counts hold.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int E>
__global__ void __launch_bounds__(kNT, 4) k_resample3_demod_stream_dma(const float* __restrict__ in, const float* __restrict__ groups,
                                                                       float* __restrict__ out, float* __restrict__ y_out, StreamArgs a)
{
    constexpr int kSlot = 64 * kRounds + 1;                                    // uint4 per wave: kRounds rounds + the predecessor pair
    __shared__ __attribute__((aligned(16))) float seg[kSeg];
    __shared__ __attribute__((aligned(16))) uint4 raw[kNT / 64][kSlot];
    __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int wave_t0 = wave * 64;
    atan_table_fill(atbl, tid);

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = t_begin + a.tiles_per_wg < a.ntiles ? t_begin + a.tiles_per_wg : a.ntiles;
    const float2* __restrict__ z = reinterpret_cast<const float2*>(in);
    auto fast = [&](int T) { return T < t_end && (T + 1) * kNT <= a.ncycles && a.origin + (int64_t)T * kTileIn + kSeg <= a.y_count; };
    int m_new = 0;
    if (a.yseam > 0) m_new = (int)((a.y_abs0 + a.origin + (int64_t)t_begin * kTileIn + kCarry) % a.yseam);
    auto advance = [&]() {
        if (a.yseam > 0) {
            m_new += kTileIn;
            if (m_new >= a.yseam) m_new -= a.yseam;
        }
    };
    float2* seg2 = reinterpret_cast<float2*>(seg);
    int T = t_begin;
    lds_barrier();                                                             // the table
    bool carried = false;
    {
        const int64_t tile0 = a.origin + (int64_t)T * kTileIn;
        if (T < t_end && !(tile0 >= 1 || (tile0 == 0 && a.has_prev))) {
            guarded_tile(a, in, y_out, tile0, seg, tid);
            lds_barrier();
            resample_tile<E, false>(a, groups, out, T, seg, tid);
            const float2 c = tid < kCarry / 2 ? seg2[kTileIn / 2 + tid] : make_float2(0.0f, 0.0f);
            lds_barrier();                                                     // everybody has read the segment
            if (tid < kCarry / 2) seg2[tid] = c;
            advance();
            T++, carried = true;
        }
    }
    const int lp0 = kRounds * wave_t0 + lane;                                  // this thread's pair in round 0, relative to the tile's first new pair
    const unsigned off8 = (unsigned)lp0 * 8u;
    // refill slot i (-1: the predecessor pair, lane 0 only) with tile Tn's data; `live` false: a harmless valid address
    auto refill = [&](int Tn, bool live, int i) {
        const int64_t n0 = a.origin + (int64_t)Tn * kTileIn + kCarry;         // even offset from `origin`: 16-byte aligned
        const uint4* src = reinterpret_cast<const uint4*>(in + 2 * n0);
        if (i >= 0) {
            const uint4* g = src + (live ? lp0 + 64 * i : 0);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)&raw[wave][64 * i], 16, 0, 0);
        } else {
            // the pair that ends right before the wave's first one: samples n0 + 2 * kRounds * wave_t0 - 2, - 1 (for tile 0 of a
            // stream that starts at sample 0 this run is never `live`: the guarded tile took it)
            const uint4* g = src + (live ? kRounds * wave_t0 - 1 : 0);
            if (lane == 0) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)&raw[wave][64 * kRounds], 16, 0, 0);
        }
    };
    // one streaming tile; FIRST: the run's first (no output store of a previous tile in the queue)
    auto tile = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int64_t tile0 = a.origin + (int64_t)T * kTileIn;
        const bool more = fast(T + 1);
        const int Tn = more ? T + 1 : T;
        if (!carried && tid < kCarry / 2) {
            const int64_t n = tile0 + 2 * tid;
            const float2 pv = z[n - 1], A = z[n], B = z[n + 1];
            const float2 y = make_float2(fm_phase_sel(A, pv), fm_phase_sel(B, A));
            seg2[tid] = y;
            if (a.yseam > 0) {
                const int m0 = (int)((a.y_abs0 + n) % a.yseam), m1 = m0 + 1 == a.yseam ? 0 : m0 + 1;
                if (keep_pos(m0, a.yseam, a.ykeep)) y_out[n] = y.x;
                if (keep_pos(m1, a.yseam, a.ykeep)) y_out[n + 1] = y.y;
            }
        }
        carried = true;
        float2 y[kRounds];
        float4 cur[kRounds];
        float2 pl = make_float2(0.0f, 0.0f);
        bool rare = false;
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            if (i == 0) wait_vm<FIRST ? 4 : 5>(); else wait_vm<FIRST ? 5 : 6>();
            const uint4 r = raw[wave][64 * i + lane];
            uint4 pr = make_uint4(0u, 0u, 0u, 0u);
            if (i == 0) pr = raw[wave][64 * kRounds];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // in registers before the slot is refilled
            if (i == 0) {
                refill(Tn, more, -1);
                pl = make_float2(__uint_as_float(pr.z), __uint_as_float(pr.w));
            }
            refill(Tn, more, i);
            cur[i] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
            float2 o = pl;
            if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].z), 63)),
                                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].w), 63)));
            const float2 pv = make_float2(dpp_shr1_or(o.x, cur[i].z), dpp_shr1_or(o.y, cur[i].w));
            bool q0, q1;
            y[i].x = fm_phase_common_tbl(A, pv, q0, atbl);
            __builtin_amdgcn_sched_barrier(0);
            y[i].y = fm_phase_common_tbl(B, A, q1, atbl);
            rare |= q0 | q1;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < kRounds; i++) seg2[kCarry / 2 + lp0 + 64 * i] = y[i];
        if (__any(rare)) {
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
                float2 o = pl;
                if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].z), 63)),
                                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].w), 63)));
                const float2 pv = make_float2(dpp_shr1_or(o.x, cur[i].z), dpp_shr1_or(o.y, cur[i].w));
                y[i] = make_float2(fm_phase_sel(A, pv), fm_phase_sel(B, A));
                seg2[kCarry / 2 + lp0 + 64 * i] = y[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.yseam > 0 && (m_new < a.ykeep || m_new + kTileIn > a.yseam - a.ykeep)) {
            char* yb = reinterpret_cast<char*>(y_out + tile0 + kCarry);
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                int m0 = m_new + 2 * (lp0 + 64 * i);
                if (m0 >= a.yseam) m0 -= a.yseam;
                const int m1 = m0 + 1 == a.yseam ? 0 : m0 + 1;
                float* yo = reinterpret_cast<float*>(yb + (off8 + 512u * i));
                if (keep_pos(m0, a.yseam, a.ykeep)) yo[0] = y[i].x;
                if (keep_pos(m1, a.yseam, a.ykeep)) yo[1] = y[i].y;
            }
        }
        const float2 ylast = y[kRounds - 1];
        lds_barrier();                                                         // D | R
        resample_tile<E, true>(a, groups, out, T, seg, tid);
        lds_barrier();                                                         // R | the next tile's D (and the carry below)
        if (tid >= kNT - kCarry / 2) seg2[tid - (kNT - kCarry / 2)] = ylast;  // the segment's last 64 positions are the next tile's first 64
        advance();
        T++;
    };
    if (fast(T)) {
        // prologue: the first tile's slots (the predecessor pair first: the counts above assume that order)
        refill(T, true, -1);
#pragma unroll
        for (int i = 0; i < kRounds; i++) refill(T, true, i);
        tile(std::true_type{});
        while (fast(T)) tile(std::false_type{});
        wait_vm<0>();                                                          // the last tile's harmless refills
    }
    for (; T < t_end; T++) {
        guarded_tile(a, in, y_out, a.origin + (int64_t)T * kTileIn, seg, tid);
        lds_barrier();
        resample_tile<E, false>(a, groups, out, T, seg, tid);
        lds_barrier();
    }
    if (a.nedge > 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        for (int sg = 0; sg < 2; sg++) {
            if (blockIdx.x != (sg == 0 ? 0u : gridDim.x - 1)) continue;
            const int64_t p0 = sg == 0 ? 0 : a.y_count - a.nedge;
            for (int i = tid; i < a.nedge; i += kNT) {
                const int64_t p = p0 + i;
                const float2 v[2] = {(p > 0 || a.has_prev) ? z[p - 1] : make_float2(0.0f, 0.0f), z[p]};
                float y[1];
                fm_phase_voted<1>(v, y);
                y_out[p] = y[0];
            }
        }
    }
}

// OFF by default: measured 0.236 ms per 2^26 inputs against 0.198 for the tile kernel with fmDemod in its loader (header comment)
std::atomic<int> g_stream_on{getenv("SDRHIP_RESAMP_STREAM") ? atoi(getenv("SDRHIP_RESAMP_STREAM")) : 0};
std::atomic<long long> g_stream_launches{0};

int device_cus()
{
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}

}  // namespace

void set_resample_demod_stream(int on) { g_stream_on.store(on, std::memory_order_relaxed); }
int resample_demod_stream_mode() { return g_stream_on.load(std::memory_order_relaxed); }
long long resample_demod_stream_launch_count() { return g_stream_launches.load(std::memory_order_relaxed); }

// The host arithmetic of the cut (CPU-testable): tiles, tiles per workgroup, workgroups
void resample_demod_stream_plan(int ncycles, int cus, int* ntiles, int* tiles_per_wg, int* grid)
{
    const int nt = (ncycles + kNT - 1) / kNT;
    const int slots = cus * SDRHIP_RSTREAM_WGS;
    const int per = nt > 0 ? (nt + slots - 1) / slots : 1;
    *ntiles = nt;
    *tiles_per_wg = per;
    *grid = nt > 0 ? (nt + per - 1) / per : 0;
}

bool launch_resample3_demod_stream(hipStream_t s, const float* d_iq, int64_t pos, int ncycles, bool iq_has_prev, int64_t y_count,
                                   const float* d_groups, int row_stride, float* d_out, float* d_y, int64_t y_abs0, int yseam, int ykeep,
                                   int nedge)
{
    int mode = g_stream_on.load(std::memory_order_relaxed);
    const bool dma = mode >= 1000;                               // 1000 + m: the LDS-DMA variant in mode m
    if (dma) mode -= 1000;
    if (mode == 0 || ncycles < 1 || pos < 0) return false;
    if (yseam > 0 && yseam < kSeg + 2 * ykeep) return false;     // a tile's new samples wrap the seam grid at most once
    int ntiles, per, grid;
    resample_demod_stream_plan(ncycles, device_cus(), &ntiles, &per, &grid);
    if (dma) {                                                   // four workgroups per CU
        const int slots = device_cus() * 4;
        per = ntiles > 0 ? (ntiles + slots - 1) / slots : 1;
        grid = ntiles > 0 ? (ntiles + per - 1) / per : 0;
    }
    // a run of fewer than a handful of tiles per workgroup has nothing to stream behind: the tile kernel serves it
    if (mode == 1 && per < 4) return false;
    if (mode > 2) {                                              // tests: every run, cut for `mode` workgroups (long runs of tiles at small sizes)
        per = (ntiles + mode - 1) / mode;
        grid = (ntiles + per - 1) / per;
    }
    const int e = (int)(((reinterpret_cast<uintptr_t>(d_iq) >> 3) + (uint64_t)pos) & 1);
    StreamArgs a = {};
    a.origin = pos - e;
    a.ntiles = ntiles; a.ncycles = ncycles; a.tiles_per_wg = per; a.has_prev = iq_has_prev ? 1 : 0;
    a.y_count = y_count;
    a.row_stride = row_stride;
    a.y_abs0 = y_abs0; a.yseam = yseam; a.ykeep = ykeep; a.nedge = nedge;
    if (dma) {
        if (e) hipLaunchKernelGGL(k_resample3_demod_stream_dma<1>, dim3(grid), dim3(kNT), 0, s, d_iq, d_groups, d_out, d_y, a);
        else hipLaunchKernelGGL(k_resample3_demod_stream_dma<0>, dim3(grid), dim3(kNT), 0, s, d_iq, d_groups, d_out, d_y, a);
    } else if (e) hipLaunchKernelGGL(k_resample3_demod_stream<1>, dim3(grid), dim3(kNT), 0, s, d_iq, d_groups, d_out, d_y, a);
    else hipLaunchKernelGGL(k_resample3_demod_stream<0>, dim3(grid), dim3(kNT), 0, s, d_iq, d_groups, d_out, d_y, a);
    g_stream_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

}  // namespace sdrhip
