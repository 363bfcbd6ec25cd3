// kernels_small.hip -- the WHOLE FM chain of examples/fm/fm.hs:34-41 as ONE kernel, for launch-bound batches:
//     u8 IQ --convert + firDecimator /8--> d[k] --fmDemod--> y[k] --firResampler 3/10--> z[m] --firFilter(sym) * gain--> audio[q]
//     convert.c:37-50, decimate.c:105-113   Demod.hs:21-46     resample.c:70-87            filter.c:60-68, fm.hs:40
//
// Why it exists.  BASELINE configs[4] shards the stream at 2^20 samples per GPU per pass.  At that size the five stage
// kernels are pure latency: each costs 4-9 us from dispatch to drain while its arithmetic needs well under a microsecond
// of the chip (the decimator launches exactly one workgroup per CU), so a pass takes 29 us = 36 Gsample/s where the same
// chip sustains 490 on a large batch.  Here a pass is ONE launch: nothing goes through HBM between the stages and there
// are no kernel boundaries to wait at.  The price is recomputation: a workgroup that produces A audio outputs needs
// A + 127 resampler outputs, (A + 129) * 10/3 + 64 fmDemod outputs and as many decimator outputs -- for A = 159 that is
// 1022 decimator outputs where a partition would give it 530, i.e. the first stage is computed 1.9 times over.  Worth it
// exactly while the launch is latency-bound (chain.cpp decides by the size of the run).
//
// One workgroup = 512 threads = a tile of A <= 159 audio outputs [qa, qa + A), qa a multiple of 3 (a polyphase cycle of
// the 3/10 resampler), computed back to front:
//   phase 0  the tile's 8312 input samples (u8 IQ, 16-byte loads, all in flight at once) -> LDS as float2 (the layout of
//            decimate_tile.hpp: the tiled decimator's loader and MAC loop are reused as they are)
//   phase 1  1024 decimator outputs d[y0 - 1 .. y0 + 1022], two per thread (mac_window, AVX "RC" order); outputs whose
//            window straddles a buffer boundary of the reference's Pipe recomputed sequentially (inline_cross_outputs)
//            -> LDS (the input tile is dead by then: its space is reused)
//   phase 2  y[j] = phase(d[j] * conj d[j-1]), two per thread                                       -> LDS
//   phase 3  z[m] for the tile's 3 * NC <= 288 resampler outputs, one per thread (8 lane partials + tree) -> LDS
//   phase 4  audio[q] = gain * sym_fir(z[q .. q+127]), one per thread (pair-add first, 8 lane partials + tree) -> HBM
// Seams (Cross outputs of every stage) are decided per output with the predicates of the fix-up kernels, exactly as
// kernels_tail.hip does.  Results are bit-identical to the stage kernels' (tests/test_gpu_chain.py).
#include "decimate_tile.hpp"
#include "demod.hpp"
#include "kernels.hpp"

namespace sdrhip {

namespace {

constexpr int SM_NT = 512;
using SmT = Tile<8, 128, 2, SM_NT>;                 // OUTS = 1024 decimator outputs, SPAN = 8312 samples, 74.8 KB of LDS
constexpr int SM_KD = SmT::OUTS;                    // decimator outputs per tile: local i <-> k = y0 - 1 + i
constexpr int SM_NL = 64;                           // resampler group length (padded)
constexpr int SM_LF = 128;                          // audio filter length (64 half-taps)
constexpr int SM_NC_MAX = 96;                       // polyphase cycles per tile: 10 * (NC - 1) + 7 + 64 <= SM_KD - 1
constexpr int SM_A_MAX = 3 * SM_NC_MAX - (SM_LF - 1) - 2;   // 159 audio outputs per tile
static_assert(10 * (SM_NC_MAX - 1) + 7 + SM_NL <= SM_KD - 1, "the tile's y values come from its own decimator outputs");
static_assert(SM_A_MAX % 3 == 0, "tiles start on a polyphase cycle");
constexpr int SM_NV = SmT::SPAN / 8;                // 16-byte vectors of u8 IQ per tile
static_assert(SmT::SPAN % 8 == 0, "whole vectors");
constexpr int SM_PER = (SM_NV + SM_NT - 1) / SM_NT;

struct SmallParams {
    int64_t s0, n_in;        // d_in holds samples [s0, s0 + n_in) of the stream
    int64_t q0, q1;          // audio outputs of this launch
    int A;                   // audio outputs per tile (multiple of 3, <= SM_A_MAX)
    int row_stride;          // floats between the resampler's group rows
    int ntaps;               // resampler taps (unpadded)
    int rLp;                 // numCoeffsR (192)
    float gain;
    int64_t seam;            // block size B of every Pipe (0 = contiguous stream)
};


// The reference's horizontal add of 8 lane partials, ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)) (common.h:18-29), over the 8 lanes
// of a group as DPP operands of three adds: xor-1, xor-2 partners (quad_perm), then lane l + 4 (row_shl:4).  IEEE addition
// is commutative, so lane 0 of the group -- the only one whose result is used -- holds exactly that tree.
template <int STEP>
__device__ __forceinline__ float sm_partner(float v)
{
    constexpr int ctrl = STEP == 1 ? 0xB1 /* quad_perm:[1,0,3,2] */ : STEP == 2 ? 0x4E /* quad_perm:[2,3,0,1] */ : 0x104 /* row_shl:4 */;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}
__device__ __forceinline__ float sm_tree8(float a)
{
    a = a + sm_partner<1>(a);
    a = a + sm_partner<2>(a);
    return a + sm_partner<4>(a);
}

// The two outputs of a thread in the reference's SEQUENTIAL order (decimateCrossHighLevel, FilterInternal.hs:397-402): one
// partial sum per output, taps in increasing order from +0, separate multiply and add.  Output 1's window is output 0's
// shifted by one block of 8 samples, so a block read from LDS serves both; blocks and tap chunks are fetched one step ahead
// of the arithmetic (the fence inside load_tap_chunk pins that order): the straddlers' wave is the workgroup's critical
// path, and a read-wait-use loop would cost four times as much.
template <class T>
__device__ __forceinline__ void seq_window2(const float2* __restrict__ win, const float* __restrict__ ltaps /* LDS */, float2 (&out)[2])
{
    static_assert(T::CHUNK == 16, "two outputs, 8 samples apart, per thread");
    constexpr int NB = 128 / 8;                         // tap blocks; the window of the pair has NB + 1 sample blocks
    auto rd_block = [&](int b, float4 (&dst)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sidx = 8 * b + 2 * i;
            dst[i] = *reinterpret_cast<const float4*>(&win[sidx + 2 * (sidx / T::CHUNK)]);
        }
    };
    float2 a0 = make_float2(0.0f, 0.0f), a1 = make_float2(0.0f, 0.0f);
    // The taps come from an LDS copy (every lane reads the same address: a broadcast), NOT by scalar loads: a scalar-load wait
    // is a full lgkmcnt(0) drain, which would expose the latency of the sample reads just issued in every step.
    float4 tcur[2], tnext[2];
    tcur[0] = *reinterpret_cast<const float4*>(ltaps);
    tcur[1] = *reinterpret_cast<const float4*>(ltaps + 4);
    // one step: fetch sample block b + 2 and tap chunk b + 1, then the 8 taps of chunk b on blocks b (output 0) and b + 1
    // (output 1).  Fully unrolled over a ring of three block buffers; a fence per step (an empty volatile asm that takes the
    // accumulators and clobbers memory) keeps every step's loads AND arithmetic in their own step -- left alone, the compiler
    // moves all loads to the top and spills -- so the waits are counted ones and no LDS latency is exposed after the first step.
    float4 blk[3][4];
    rd_block(0, blk[0]);
    rd_block(1, blk[1]);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (b + 1 < NB) {
            tnext[0] = *reinterpret_cast<const float4*>(ltaps + 8 * (b + 1));
            tnext[1] = *reinterpret_cast<const float4*>(ltaps + 8 * (b + 1) + 4);
        }
        if (b + 2 <= NB) rd_block(b + 2, blk[(b + 2) % 3]);
        const float hh[8] = {tcur[0].x, tcur[0].y, tcur[0].z, tcur[0].w, tcur[1].x, tcur[1].y, tcur[1].z, tcur[1].w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float h = hh[i];
            const float4 v0 = blk[b % 3][i / 2], v1 = blk[(b + 1) % 3][i / 2];
            const float2 x0 = (i & 1) ? make_float2(v0.z, v0.w) : make_float2(v0.x, v0.y);
            const float2 x1 = (i & 1) ? make_float2(v1.z, v1.w) : make_float2(v1.x, v1.y);
            a0.x = a0.x + h * x0.x;
            a0.y = a0.y + h * x0.y;
            a1.x = a1.x + h * x1.x;
            a1.y = a1.y + h * x1.y;
        }
        // the step's fence: volatile asms keep their order, this one consumes the step's arithmetic and clobbers memory
        asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y) : : "memory");
        tcur[0] = tnext[0];
        tcur[1] = tnext[1];
    }
    out[0] = a0;
    out[1] = a1;
}

// ONE output per lane in the sequential order, for the lane-per-straddler pass: output i of the tile (window = samples
// [8 i, 8 i + 128) of the tile, any i -- so the window may start in the middle of a 16-sample chunk of the padded layout:
// reads of the chunk halves that follow an odd start go through a second base pointer).  Packed multiply / add on
// (re, im); taps from the LDS copy; blocks of 8 taps, the next block's reads in flight during this block's arithmetic.
typedef float sm_v2f __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ float2 seq_one(const float2* __restrict__ tile, int i, const float* __restrict__ ltaps /* LDS */)
{
    static_assert(T::CHUNK == 16, "layout: 2 float2 of padding per 16 samples");
    constexpr int NB = 128 / 8;
    const int par = i & 1;
    const float2* wa = tile + 8 * i + 2 * (i >> 1);      // sample s of the window: wa[s + 2 * (s >> 4)] while (s & 15) < 8 ...
    const float2* wb = wa + 2 * par;                     // ... and wb[s + 2 * (s >> 4)] in the upper half of a 16-block
    auto rd_block = [&](int b, float4 (&dst)[4]) {
        const float2* w = (b & 1) ? wb : wa;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int sidx = 8 * b + 2 * t;
            dst[t] = *reinterpret_cast<const float4*>(&w[sidx + 2 * (sidx >> 4)]);
        }
    };
    sm_v2f acc = {0.0f, 0.0f};
    float4 blk[2][4], tp[2][2];
    rd_block(0, blk[0]);
    tp[0][0] = *reinterpret_cast<const float4*>(ltaps);
    tp[0][1] = *reinterpret_cast<const float4*>(ltaps + 4);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (b + 1 < NB) {
            rd_block(b + 1, blk[(b + 1) & 1]);
            tp[(b + 1) & 1][0] = *reinterpret_cast<const float4*>(ltaps + 8 * (b + 1));
            tp[(b + 1) & 1][1] = *reinterpret_cast<const float4*>(ltaps + 8 * (b + 1) + 4);
        }
        const float4 t0 = tp[b & 1][0], t1 = tp[b & 1][1];
        const float hh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float4 v = blk[b & 1][u / 2];
            const sm_v2f x = (u & 1) ? sm_v2f{v.z, v.w} : sm_v2f{v.x, v.y};
            const sm_v2f h2 = {hh[u], hh[u]};
            acc = acc + h2 * x;
        }
        // the step's fence (see seq_window2): loads and arithmetic stay in their step
        asm volatile("" : "+v"(acc) : : "memory");
    }
    return make_float2(acc.x, acc.y);
}

// dtaps: the decimator's 128 plain taps pre-scaled by 1/128 (FirDesc::d_scaled); groups: 3 rows of row_stride floats;
// rplain: the resampler's plain taps; fplain: the audio filter's 128 plain taps (coeffs ++ reverse coeffs: its first 64 are the half-taps)
template <int PSKIP>
__global__ void __launch_bounds__(SM_NT, 2) k_fm_chain_small(const uint8_t* __restrict__ in, float* __restrict__ audio,
                                                             const float* __restrict__ dtaps, const float* __restrict__ groups,
                                                             const float* __restrict__ rplain, const float* __restrict__ fplain,
                                                             SmallParams p
)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    __shared__ __attribute__((aligned(16))) float gtab[3 * SM_NL];   // the three polyphase groups
    __shared__ float rpl[3 * SM_NL];                                  // un-grouped taps of the two sequential (Cross) kernels
    __shared__ float fpl[SM_LF];
    __shared__ __attribute__((aligned(16))) float dtl[128];           // the decimator's taps (scaled) for the sequential outputs
    const int tid = threadIdx.x;

    const int A = p.A;
    const int64_t qa = (p.q0 / 3) * 3 + (int64_t)blockIdx.x * A;     // first audio output of the tile (cycle aligned)
    const int64_t c0 = qa / 3;                                        // first resampler cycle
    const int64_t y0 = 10 * c0;                                       // first y of the tile = in_offset(3 c0)
    const int64_t kb = y0 - 1;                                        // decimator output of local index 0 (-1 at stream start)
    const int NC = A / 3 + (SM_LF - 1 + 2) / 3;                       // cycles of the tile: (A + 129) / 3
    const int NY = 10 * (NC - 1) + 7 + SM_NL;                         // y values the cycles touch
    // what this launch needs from the tile (a short launch fills only part of its last tile)
    const int64_t m_need_lo = p.q0 > qa ? p.q0 : qa;
    const int64_t q_hi = p.q1 < qa + A ? p.q1 : qa + A;
    const int64_t m_need_hi = q_hi + SM_LF - 1;

    // ---- phase 0: the tile's input, [8 kb, 8 kb + SPAN), into LDS.  Samples outside [s0, s0 + n_in) feed no output this
    // launch owns (sdrhip_fm_chain_run checks the receptive field of [q0, q1)): they read as u8 128 = 0.0f.
    {
        Stage<SmT, true, SM_NT> st;
        const int64_t rel0 = kb * 8 - p.s0;                           // tile start relative to in[0], a multiple of 8 samples
        const bool interior = rel0 >= 0 && rel0 + SmT::SPAN <= p.n_in;
        if (interior) {
            const uint8_t* src = in + 2 * rel0;
#pragma unroll
            for (int i = 0; i < SM_PER; i++) {
                const int v = tid + i * SM_NT;
                if (i + 1 < SM_PER || v < SM_NV) st.r[i] = *reinterpret_cast<const uint4*>(src + 16 * (int64_t)v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < SM_PER; i++) {
                const int v = tid + i * SM_NT;
                uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                if (v < SM_NV) {
                    const int64_t r8 = rel0 + 8 * (int64_t)v;
                    if (r8 >= 0 && r8 + 8 <= p.n_in) {
                        const uint4 q = *reinterpret_cast<const uint4*>(in + 2 * r8);
                        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
                    } else {
                        for (int e = 0; e < 16; e++) {
                            const int64_t sidx = r8 + e / 2;
                            if (sidx >= 0 && sidx < p.n_in)
                                w[e >> 2] = (w[e >> 2] & ~(0xffu << (8 * (e & 3)))) | ((uint32_t)in[2 * r8 + e] << (8 * (e & 3)));
                        }
                    }
                }
                st.r[i] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        // the small tables ride behind the tile's loads (same HBM round trip)
        float t_g = 0.0f, t_r = 0.0f, t_f = 0.0f;
        if (tid < 3 * SM_NL) {
            t_g = groups[(tid / SM_NL) * p.row_stride + (tid % SM_NL)];
            t_r = tid < p.ntaps ? rplain[tid] : 0.0f;
        }
        if (tid < SM_LF) t_f = fplain[tid];
        else if (tid < SM_LF + 128) t_f = dtaps[tid - SM_LF];
        st.store(lds);
        if (tid < 3 * SM_NL) { gtab[tid] = t_g; rpl[tid] = t_r; }
        if (tid < SM_LF) fpl[tid] = t_f;
        else if (tid < SM_LF + 128) dtl[tid - SM_LF] = t_f;
    }
    __syncthreads();

    // ---- phase 1: decimator outputs k = kb + 2 tid + {0, 1}.  Waves whose outputs no y of the tile reads skip the MACs.
    float2 res[2] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    const int wave_first = 2 * (tid & ~63);                           // local index of the wave's first output
    const float2* win = lds + SmT::lds_idx(tid * SmT::CHUNK);
    if (wave_first <= NY) {
        float2 acc[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[r][k] = make_float2(0.0f, 0.0f);
        mac_window<8, 128, 2, SmT, 8, false, 4, PSKIP>(win, dtaps, acc, 0);
#pragma unroll
        for (int r = 0; r < 2; r++) res[r] = fold_partials<4, 0>(acc[r]);
    }
    // Cross outputs (decimateCrossHighLevel, FilterInternal.hs:397-402): window [8k, 8k + 128) straddles a multiple of seam.
    // They sit in runs of 15 or 16 consecutive outputs below each buffer boundary, i.e. in a handful of lanes of one or two
    // waves, and whatever wave computes them is the workgroup's critical path (a lone wave issues one VALU instruction
    // every ~5 cycles): so they are re-dealt ONE PER LANE -- lane group n of 16 lanes takes the candidates of boundary n --
    // which halves the instructions of that path (seq_one).  More than 32 boundaries per tile (buffers shorter than 260
    // samples): the owners walk their own pair instead (seq_window2).
    bool cross[2] = {false, false};
    float2 sv = make_float2(0.0f, 0.0f);
    int si = -1;                                                      // the output this lane recomputed (fast path)
    bool owners_walk = false;
    if (p.seam > 0) {
        const int seam = (int)p.seam;
        int rt = (int)((kb * 8) % p.seam);                            // position of the tile's first sample inside its buffer
        if (rt < 0) rt += seam;
#pragma unroll
        for (int r = 0; r < 2; r++) cross[r] = (rt + (2 * tid + r) * 8) % seam + 128 > seam;
        const int e0 = seam - rt;                                     // first boundary, in samples from the tile start
        const int nbnd = e0 < SmT::SPAN ? (SmT::SPAN - 1 - e0) / seam + 1 : 0;
        owners_walk = nbnd > 32;
        if (!owners_walk) {
            const int n = 4 * (tid >> 6) + ((tid & 63) >> 4);         // this lane group's boundary
            bool valid = false;
            int i = 0;
            if (n < nbnd) {
                const int e = e0 + n * seam;
                i = ((e - 128) >> 3) + 1 + (tid & 15);                // candidates: 8 i < e < 8 i + 128
                valid = i <= ((e + 7) >> 3) - 1 && i >= 0 && i < SM_KD;
            }
            if (__any(valid)) {
                sv = seq_one<SmT>(lds, valid ? i : 0, dtl);
                if (valid) si = i;
            }
        } else if (wave_first <= NY && __any(cross[0] || cross[1])) {
            float2 seq[2];
            seq_window2<SmT>(win, dtl, seq);
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (cross[r]) res[r] = seq[r];
        }
    }
    __syncthreads();                                                  // every read of the input tile is done: reuse its space
    float2* dl = lds;                                                 // d of the tile, SM_KD float2
    float* ys = reinterpret_cast<float*>(dl + SM_KD);                 // y[j] <-> k = y0 + j, SM_KD floats
    float* zs = ys + SM_KD + 16;                                      // z[i] <-> m = 3 c0 + i, 3 NC floats
    {
        // d[-1] = 0: fmDemod's carried sample at stream start (Demod.hs:41).  Cross outputs come from the lanes that
        // recomputed them (fast path), everything else from its owner.
        const bool skip0 = cross[0] && !owners_walk, skip1 = cross[1] && !owners_walk;
        if (!skip0) dl[2 * tid] = kb + 2 * tid >= 0 ? res[0] : make_float2(0.0f, 0.0f);
        if (!skip1) dl[2 * tid + 1] = res[1];
        if (si >= 0) dl[si] = kb + si >= 0 ? sv : make_float2(0.0f, 0.0f);
    }
    __syncthreads();

    // ---- phase 2: fmDemod, y[j] = phase(d[j + 1] * conj d[j]) in local indices, two per thread
    {
        const int j = 2 * tid;
        if (j < NY) {
            const float2 a = dl[j], b = dl[j + 1];
            const float2 c = j + 2 < SM_KD ? dl[j + 2] : make_float2(0.0f, 0.0f);
            const float2 v[3] = {a, b, c};
            float yv[2];
            fm_phase_voted<2>(v, yv);                             // common case + vote among the lanes in here (demod.hpp)
            *reinterpret_cast<float2*>(&ys[j]) = make_float2(yv[0], yv[1]);
        }
    }
    __syncthreads();

    // ---- phase 3: polyphase resampler.  z[3 cl + g] = sum_j groups[g][j] * y[10 cl + pre[g] + j] with lane partial j & 7
    // accumulated from +0 in increasing j and the tree ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)) (resample.c:70-87, common.h:18-29).
    // The 8 lane partials of an output ARE 8 lanes here (8 MACs each, then the tree as three DPP adds): 64 outputs per
    // round of the workgroup instead of one 64-MAC chain per thread -- the phase is latency, not throughput.
    // seam positions in 32-bit arithmetic relative to the tile: one 64-bit modulo per workgroup and stage (rz0 / rq0 = position
    // of the tile's first z window / first audio window inside its buffer, in upsampled units / z samples)
    const int sBI = (int)(p.seam * 3);
    const int rz0 = sBI > 0 ? (int)((30 * c0) % sBI) : 0;
    const int rq0 = p.seam > 0 ? (int)(qa % p.seam) : 0;
    const int zi_lo = (int)(m_need_lo - qa);
    const int zi_hi = (int)(m_need_hi - qa) < 3 * NC ? (int)(m_need_hi - qa) : 3 * NC;
    {
        // 63 outputs per round (group t = tid / 8 < 63 takes outputs zi_lo + t + 63 rd): 63 is a multiple of 3, so a lane keeps
        // its polyphase group -- and with it its 8 taps, in registers -- across the rounds (taps read per round from the LDS
        // table are a three-way bank conflict: the three groups' rows are 64 floats apart)
        const int l = tid & 7, t = tid >> 3;
        constexpr int ZG = 63;
        constexpr int ZR = (3 * SM_NC_MAX + ZG - 1) / ZG;             // rounds at most: 5
        const int g = (zi_lo + t) % 3;
        float cv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = gtab[SM_NL * g + l + 8 * i];
        const int pre = g == 0 ? 0 : (g == 1 ? 4 : 7);
        float yv[ZR][8];
#pragma unroll
        for (int rd = 0; rd < ZR; rd++) {
            int zi = zi_lo + rd * ZG + t;
            if (zi >= zi_hi || t >= ZG) zi = zi_lo + (t % 3);         // idle groups: an output of their own group (never stored)
            const float* w = ys + 10 * (zi / 3) + pre + l;
#pragma unroll
            for (int i = 0; i < 8; i++) yv[rd][i] = w[8 * i];
        }
#pragma unroll
        for (int rd = 0; rd < ZR; rd++) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) acc = acc + cv[i] * yv[rd][i];
            const float r = sm_tree8(acc);
            const int zi = zi_lo + rd * ZG + t;
            if (l == 0 && t < ZG && zi < zi_hi) zs[zi] = r;
        }
    }
    if (sBI > 0) {
        // Cross outputs (resampleCrossHighLevel, FilterInternal.hs:410-423: stride 3 through the unpadded taps, sequential),
        // one per thread, written over the One value another lane stored above (hence the barrier)
        __syncthreads();
        const int zi = zi_lo + tid;
        if (zi < zi_hi) {
            const int rv = (rz0 + 10 * zi) % sBI;                     // window start inside its buffer (upsampled units)
            if (rv + p.rLp > sBI) {                                   // the window straddles the buffer boundary `edge`
                const int64_t m = 3 * c0 + zi;
                const int64_t v = m * 10;
                const int64_t edge = v + (sBI - rv);
                if (seam_has_crossover(edge, 3, 10, p.rLp) && !late_output_is_one(m, edge, 3, 10, p.seam)) {
                    const int64_t pos = (v + 2) / 3;
                    const int fo = (int)(pos * 3 - v);
                    const float* x = ys + (pos - y0);
                    const int nterms = (p.ntaps - fo + 2) / 3;        // taps fo, fo+3, .. < ntaps
                    const float* tp = rpl + fo;
                    float sacc = 0.0f;
#pragma unroll 16
                    for (int t = 0; t < nterms; t++) sacc = sacc + x[t] * tp[3 * t];
                    zs[zi] = sacc;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 4: symmetric audio filter + gain: pair-add first, lane partial k & 7 from +0 in increasing k, tree
    // (filter.c:60-68, common.h:181-201); fm.hs:40's `* gain` as a separate multiply of the rounded output.  8 lanes per
    // output as in phase 3; lane l keeps its 8 half-taps c[l], c[l + 8], .. in registers.
    {
        const int l = tid & 7;
        const int o_lo = (int)(m_need_lo - qa), o_hi = (int)(q_hi - qa);
        float cf[8];
#pragma unroll
        for (int i = 0; i < 8; i++) cf[i] = fpl[l + 8 * i];            // coeffs ++ reverse coeffs: the first half are the half-taps
        constexpr int FR = (SM_A_MAX * 8 + SM_NT - 1) / SM_NT;         // rounds at most: 3
        float fa[FR][8], fb[FR][8];
#pragma unroll
        for (int rd = 0; rd < FR; rd++) {
            int o = o_lo + rd * (SM_NT / 8) + (tid >> 3);
            if (o >= o_hi) o = o_hi - 1;
            const float* win = zs + o;
#pragma unroll
            for (int i = 0; i < 8; i++) { fa[rd][i] = win[l + 8 * i]; fb[rd][i] = win[SM_LF - 1 - l - 8 * i]; }
        }
        float seq_c = 0.0f;
        bool is_cross = false;
        if (p.seam > 0) {
            // filterCrossHighLevel (FilterInternal.hs:404-408) on coeffs ++ reverse coeffs: sequential, no pair-add; one output per
            // thread (a tile either has none of them or up to 127)
            const int o = o_lo + tid;
            const int rq = (rq0 + o) % (int)p.seam;                   // window start inside its buffer of z samples
            is_cross = o < o_hi && rq + SM_LF > (int)p.seam;
            if (is_cross) {
                const float* win = zs + o;
#pragma unroll 32
                for (int j = 0; j < SM_LF; j++) seq_c = seq_c + win[j] * fpl[j];
            }
        }
#pragma unroll
        for (int rd = 0; rd < FR; rd++) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) acc = acc + cf[i] * (fa[rd][i] + fb[rd][i]);
            const float r = sm_tree8(acc);
            const int o = o_lo + rd * (SM_NT / 8) + (tid >> 3);
            if (l == 0 && o < o_hi) {
                bool cr = false;
                if (p.seam > 0) cr = (rq0 + o) % (int)p.seam + SM_LF > (int)p.seam;
                if (!cr) audio[qa + o - p.q0] = r * p.gain;
            }
        }
        if (is_cross) audio[qa + o_lo + tid - p.q0] = seq_c * p.gain;
    }
}


std::atomic<long long> g_small_launches{0};

}  // namespace

long long fm_chain_small_launch_count() { return g_small_launches.load(); }

int fm_chain_small_tile_outputs(int64_t n_out)
{
    // about one tile per CU (256 of them) while the run is small enough; never below 96 outputs per tile (the tile's
    // overlap -- 127 resampler outputs, ~490 decimator outputs -- is paid, and read, once per tile)
    int64_t a = (n_out + 255) / 256;
    a = (a + 2) / 3 * 3;
    if (a < 96) a = 96;
    if (a > SM_A_MAX) a = SM_A_MAX;
    return (int)a;
}

bool launch_fm_chain_small(hipStream_t s, const uint8_t* d_in, int64_t s0, int64_t n_in, float* d_audio, int64_t q0, int64_t q1,
                           int dD, int dP, const float* d_dscaled, bool last_tap_zero, const float* d_groups, int row_stride, int nloop,
                           const int* increments, int ngroups, int I, int D, int rLp, const float* d_rplain, int ntaps,
                           const float* d_fhalf, int nhalf, const float* d_fplain, float gain, int64_t seam, int tile_outputs)
{
    // specialised for the FM chain of BASELINE configs[2] / [4]: /8 decimator with 128 (padded) taps in the AVX order, 3/10
    // resampler with 64-float groups, 64 half-tap symmetric filter
    if (!(dD == 8 && dP == 128 && d_dscaled != nullptr)) return false;
    if (!(ngroups == 3 && nloop == SM_NL && I == 3 && D == 10 && increments[0] == 4 && increments[1] == 3 && increments[2] == 3)) return false;
    if (!(nhalf == SM_LF / 2 && d_fhalf != nullptr && rLp <= 3 * SM_NL && ntaps <= rLp)) return false;
    if (seam != 0 && (seam < 192 || seam > (1 << 26))) return false;       // 32-bit seam arithmetic; a stage's window meets one boundary at most
    if ((reinterpret_cast<uintptr_t>(d_in) & 15) != 0 || (s0 & 7) != 0) return false;   // 16-byte aligned tile loads
    if (q1 <= q0) return true;
    // tile_outputs < 0: the largest tile -- for input that is read over PCIe (the host-block operators' in-place pushes): a tile's
    // ~4000-sample overlap is READ once per tile, and on the link that traffic, not the number of workgroups, is what a push costs
    // (round 6, tools/stream_tile_probe.py: 16 blocks per push 22.1 -> 14.6 us with 159 instead of 96 outputs per tile)
    int A = tile_outputs > 0 ? tile_outputs : tile_outputs < 0 ? SM_A_MAX : fm_chain_small_tile_outputs(q1 - (q0 / 3) * 3);
    A = A / 3 * 3;
    if (A < 3) A = 3;
    if (A > SM_A_MAX) A = SM_A_MAX;
    SmallParams p;
    p.s0 = s0; p.n_in = n_in; p.q0 = q0; p.q1 = q1; p.A = A;
    p.row_stride = row_stride; p.ntaps = ntaps; p.rLp = rLp; p.gain = gain; p.seam = seam;
    const int64_t qa0 = (q0 / 3) * 3;
    const int64_t tiles = (q1 - qa0 + A - 1) / A;
    static std::atomic<bool> attr_set[2][64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto launch = [&](auto kern, int which) {
        if (dev < 0 || dev >= 64 || !attr_set[which][dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SmT::LDS_BYTES);
            if (dev >= 0 && dev < 64) attr_set[which][dev] = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(SM_NT), SmT::LDS_BYTES, s, d_in, d_audio, d_dscaled, d_groups, d_rplain, d_fplain, p);
    };
    if (last_tap_zero) launch(k_fm_chain_small<1>, 1);
    else launch(k_fm_chain_small<0>, 0);
    g_small_launches++;
    return true;
}

}  // namespace sdrhip
