// decimate_tile.hpp -- the LDS-tiled complex FIR decimator (templates), shared by kernels_fast.hip (the AVX "RC" order of
// the FM chain) and kernels_fast_orders.hip (the SSE and "RC2" orders).  See kernels_fast.hip for the design notes.
#pragma once
#include <atomic>
#include <stdlib.h>

#include "crossfix.hpp"
#include "kernels.hpp"

namespace sdrhip {
namespace {

template <int D, int P, int R, int NT>
struct Tile {
    static constexpr int OUTS = NT * R;                   // outputs per workgroup
    static constexpr int SPAN = (OUTS - 1) * D + P;       // input samples per workgroup
    static constexpr int CHUNK = D * R;                   // samples between adjacent threads' windows
    static constexpr int WIN = (R - 1) * D + P;           // samples one thread reads
    static_assert(CHUNK % 2 == 0, "chunk must hold whole float4s");
    // padded LDS layout: 2 float2 of padding after every CHUNK samples
    __host__ __device__ static constexpr int lds_idx(int s) { return s + 2 * (s / CHUNK); }
    // + 8 float2 of slack: the last thread's final sample-block prefetch of mac_window (blocks of up to 8 samples, NB = ceil(WIN / TC))
    // may reach past SPAN when D is not a multiple of the block (the D = 1 filter tiles); the values are never used (taps >= P are
    // skipped) but the reads must stay inside the allocation
    static constexpr int LDS_F2 = SPAN + 2 * (SPAN / CHUNK) + 2 + 8;
    static constexpr size_t LDS_BYTES = (size_t)LDS_F2 * 8;
};

// One scalar load of TC taps, pinned in program order: the empty volatile asm makes
// the address opaque (the load cannot be hoisted above it, nor out of a loop) and
// fences memory operations, so neither the tap loads nor the LDS reads of later
// sample blocks can pile up at the top of the unrolled code.  Constant address
// space + an SGPR-resident address => s_load_dwordx8.
template <int TC> struct TapVec;
template <> struct TapVec<8> { typedef float type __attribute__((ext_vector_type(8))); };
template <> struct TapVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int TC>
__device__ __forceinline__ typename TapVec<TC>::type load_tap_chunk(const float* taps, int c)
{
    typedef const __attribute__((address_space(4))) typename TapVec<TC>::type* ctapp;
    uint64_t a = reinterpret_cast<uint64_t>(taps) + (4u * TC) * (uint32_t)c;
    asm volatile("" : "+s"(a));
    return *reinterpret_cast<ctapp>(a);
}

// The MAC loop of one thread: WIN samples from LDS (two per ds_read_b128), each
// feeding up to R outputs; fully unrolled.  Taps are wave-uniform and live in SGPRs.
// They arrive 8 at a time, one chunk ahead of first use: a scalar-load wait is a
// full lgkmcnt(0) drain (SMEM returns out of order) that also stalls the LDS
// pipeline, so there must be few of them -- P/8 per tile instead of one per tap pair.
// GUARD: the filter actually has nch_eff * TC <= P taps (run-time, wave-uniform); blocks and (block, output) pairs that
// only meet taps beyond them are skipped, so one instantiation serves every shorter filter with exactly the arithmetic of
// an exact-length kernel (nothing is multiplied by padding).
// PSKIP: the last PSKIP taps are known to be zero (the padding `Filter.hs:146-148` appends) and every sample is finite (u8
// input): their MACs are skipped.  Exact: a partial sum that started at +0 is never -0 (x + (-x) and (+0) + (-0) are +0 in
// round-to-nearest, and an addition never underflows to zero), so adding the product (+-0) * (finite) = +-0 leaves it as it is.
template <int D, int P, int R, class T, int TC, bool GUARD, int NP = 4, int PSKIP = 0>
__device__ __forceinline__ void mac_window(const float2* __restrict__ win, const float* __restrict__ taps, float2 (&acc)[R][NP],
                                           int nch_eff)
{
    static_assert(P % TC == 0, "taps are walked in blocks of TC");
    // the guarded walk pairs whole tap chunks with whole sample blocks: decimation a multiple of TC.  The exact-length walk
    // takes any decimation, D = 1 (a FILTER: R consecutive outputs per thread, a sample meets up to R taps) included; its
    // last sample block may then reach past the window (taps >= P are skipped, the reads stay inside the tile's slack: Tile::LDS_F2)
    static_assert(!GUARD || (T::WIN % TC == 0 && D % TC == 0), "the guarded walk pairs tap chunks with sample blocks");
    static_assert(TC % NP == 0 || NP % TC == 0, "partial of a tap = its index mod NP");
    static_assert(!GUARD || TC % NP == 0, "the guarded walk derives the partial from the index inside the chunk");
    constexpr int NCH = P / TC;
    const int nb_eff = GUARD ? nch_eff + (R - 1) * (D / TC) : 0;   // sample blocks that still meet a live tap
    constexpr int NB = (T::WIN + TC - 1) / TC;  // sample blocks per thread
    typename TapVec<TC>::type tc[NCH];
    float4 buf[2][TC / 2];                      // LDS reads are double-buffered one block ahead
    tc[0] = load_tap_chunk<TC>(taps, 0);
#pragma unroll
    for (int i = 0; i < TC / 2; i++) buf[0][i] = *reinterpret_cast<const float4*>(&win[2 * i + 2 * ((2 * i) / T::CHUNK)]);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (GUARD && b >= nb_eff) continue;         // wave-uniform: nothing left for this block
        // the fence inside load_tap_chunk keeps these reads (block b+1) here, ahead of block b's MACs
        if (b + 1 < NCH && (!GUARD || b + 1 < nch_eff)) tc[b + 1] = load_tap_chunk<TC>(taps, b + 1);   // never past the filter's own taps
        else asm volatile("" ::: "memory");
        if (b + 1 < NB) {
#pragma unroll
            for (int i = 0; i < TC / 2; i++) {
                const int s = TC * (b + 1) + 2 * i;
                buf[(b + 1) & 1][i] = *reinterpret_cast<const float4*>(&win[s + 2 * (s / T::CHUNK)]);
            }
        }
        if constexpr (!GUARD) {
#pragma unroll
            for (int i = 0; i < TC / 2; i++) {
                const float4 v2 = buf[b & 1][i];
                const float2 v[2] = {make_float2(v2.x, v2.y), make_float2(v2.z, v2.w)};
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int ss = TC * b + 2 * i + e;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int j = ss - r * D;
                        if (j >= 0 && j < P - PSKIP) {
                            const float h = tc[j / TC][j % TC];
                            acc[r][j % NP].x = acc[r][j % NP].x + h * v[e].x;
                            acc[r][j % NP].y = acc[r][j % NP].y + h * v[e].y;
                        }
                    }
                }
            }
        } else {
            // output r meets tap chunk cb = b - r*D/TC in this block: live iff 0 <= cb < nch_eff (per output the taps are
            // still walked in increasing order, which is all the summation order asks for)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int cb = b - r * (D / TC);
                if (cb < 0 || cb >= NCH) continue;
                if (cb >= nch_eff) continue;        // wave-uniform
#pragma unroll
                for (int i = 0; i < TC / 2; i++) {
                    const float4 v2 = buf[b & 1][i];
                    const float2 v[2] = {make_float2(v2.x, v2.y), make_float2(v2.z, v2.w)};
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int jj = 2 * i + e;   // tap cb*TC + jj
                        const float h = tc[cb][jj];
                        acc[r][jj % NP].x = acc[r][jj % NP].x + h * v[e].x;      // TC is a multiple of NP: tap cb*TC + jj sits in partial jj % NP
                        acc[r][jj % NP].y = acc[r][jj % NP].y + h * v[e].y;
                    }
                }
            }
        }
    }
}

// Staging of one tile: all of a thread's 16-byte global loads are issued before the first wait, then
// converted (u8) and written to the padded LDS layout.
template <class T, bool U8, int NT>
struct Stage {
    static constexpr int SPV = U8 ? 8 : 2;                       // samples per 16-byte vector
    static constexpr int NV = (T::SPAN + SPV - 1) / SPV;         // vectors per tile
    static constexpr int PER = (NV + NT - 1) / NT;               // vectors per thread
    uint4 r[PER];

    // a whole tile: unconditional 16-byte loads into a plain register array, all in flight together
    static __device__ __forceinline__ void load_whole(const void* __restrict__ src_v, int64_t sample0, uint4 (&regs)[PER])
    {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src_v) + (U8 ? 2 : 8) * sample0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int v = threadIdx.x + i * NT;
            regs[i] = (i + 1 < PER || v < NV) ? src[v] : make_uint4(0u, 0u, 0u, 0u);
        }
    }

    __device__ __forceinline__ void load(const void* __restrict__ src_v, int64_t sample0, int avail)
    {
        const char* src = reinterpret_cast<const char*>(src_v) + (U8 ? 2 : 8) * sample0;
        if (avail >= T::SPAN) {
            // every tile but the last: unconditional 16-byte loads, all in flight together
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int v = threadIdx.x + i * NT;
                if (i + 1 < PER || v < NV) r[i] = *reinterpret_cast<const uint4*>(src + 16 * (int64_t)v);
            }
            return;
        }
        // ragged end of the stream (one workgroup per launch): element-wise, zero-filled (u8 128 == 0.0f)
#pragma unroll 1
        for (int i = 0; i < PER; i++) {
            const int v = threadIdx.x + i * NT;
            const int s = v * SPV;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = U8 ? 0x80808080u : 0u;
            if (v < NV) {
                if constexpr (U8) {
                    const uint8_t* b = reinterpret_cast<const uint8_t*>(src) + 16 * (int64_t)v;
                    for (int e = 0; e < 16; e++)
                        if (s + e / 2 < avail) w[e >> 2] = (w[e >> 2] & ~(0xffu << (8 * (e & 3)))) | ((uint32_t)b[e] << (8 * (e & 3)));
                } else {
                    const uint32_t* f = reinterpret_cast<const uint32_t*>(src) + 4 * (int64_t)v;
                    for (int e = 0; e < 4; e++)
                        if (s + e / 2 < avail) w[e] = f[e];
                }
            }
            // PER is small and compile-time: a select chain keeps r[] in registers
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (q == i) r[q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }

    __device__ __forceinline__ void store(float2* __restrict__ lds) const { store_regs(r, lds); }

    static __device__ __forceinline__ void store_regs(const uint4 (&r)[PER], float2* __restrict__ lds)
    {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int v = threadIdx.x + i * NT;
            const int s = v * SPV;
            if (v < NV) {
                if constexpr (!U8) {
                    *reinterpret_cast<uint4*>(&lds[T::lds_idx(s)]) = r[i];
                } else {
                    // u8 IQ: LDS receives m = u - 128 as a float (xor 0x80 turns the biased byte into a signed one: one
                    // v_cvt_f32_i32 with a sign-extended byte operand per component), and the kernel's taps arrive pre-scaled by
                    // 1/128 -- round((h/128) * m) and the reference's round(h * (m/128)) are the correctly rounded value of the same
                    // real number (h/128 is exact for every tap the launcher lets through), so no bit changes and the
                    // (u - 128) * (1/128) multiply disappears from the loader
                    const uint32_t w[4] = {r[i].x ^ 0x80808080u, r[i].y ^ 0x80808080u, r[i].z ^ 0x80808080u, r[i].w ^ 0x80808080u};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float4 f;
                        f.x = (float)(signed char)(w[k] & 0xff);
                        f.y = (float)(signed char)((w[k] >> 8) & 0xff);
                        f.z = (float)(signed char)((w[k] >> 16) & 0xff);
                        f.w = (float)(signed char)(w[k] >> 24);
                        const int ss = s + 2 * k;
                        if (ss < T::SPAN + 1) *reinterpret_cast<float4*>(&lds[T::lds_idx(ss)]) = f;
                    }
                }
            }
        }
    }
};

// Cross outputs computed inside a tile kernel (small launches: no fix-up launch).  An output whose window straddles a buffer
// boundary of the reference's Pipes is recomputed in the reference's sequential order (decimateCrossHighLevel,
// FilterInternal.hs:397-402) from the same LDS tile.  `taps` IS the plain tap array in tap order (for u8 input pre-scaled
// by 1/128 like the samples in LDS are un-scaled: the products are the reference's, see Stage::store); the window of
// output r starts r*D samples into the thread's.  All R chains advance together (a thread's outputs usually straddle
// together), IS taps per step so that the LDS reads and the tap load of a step are in flight at once.
template <int D, int R, class T, int TC, bool GUARD>
__device__ __forceinline__ void inline_cross_outputs(const float2* __restrict__ win, const float* __restrict__ taps, int plen,
                                                     const bool (&cross)[R], float2 (&res)[R])
{
    float re[R], im[R];
#pragma unroll
    for (int r = 0; r < R; r++) re[r] = im[r] = 0.0f;
    constexpr int IS = 4;
#pragma unroll 1
    for (int j0 = 0; j0 < plen; j0 += IS) {           // plen is a multiple of TC (4 or 8)
        float2 x[R][IS];
        float h[IS];
#pragma unroll
        for (int u = 0; u < IS; u++) {
            h[u] = taps[j0 + u];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int sidx = r * D + j0 + u;
                x[r][u] = win[sidx + 2 * (sidx / T::CHUNK)];
            }
        }
#pragma unroll
        for (int u = 0; u < IS; u++)
#pragma unroll
            for (int r = 0; r < R; r++) {
                re[r] = re[r] + x[r][u].x * h[u];
                im[r] = im[r] + x[r][u].y * h[u];
            }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        if (cross[r]) res[r] = make_float2(re[r], im[r]);
}

// The partial sums of one output and their fold (ComplexOrder of kernels.hpp):
//   NP = 4, ORD 0  "RC"  AVX   (L0 + L1) + (L2 + L3)                               decimate.c:105-113, common.h:82-90
//   NP = 2, ORD 0  "RC"  SSE   L0 + L1                                             decimate.c:84-93,  common.h:77-80
//   NP = 8, ORD 1  "RC2" AVX   q_k = p_k + p_{k+4};  (q0 + q1) + (q2 + q3)         common.h:129-155
//   NP = 4, ORD 1  "RC2" SSE   q_k = p_k + p_{k+2};  q0 + q1                       common.h:108-127
template <int NP, int ORD>
__device__ __forceinline__ float2 fold_partials(const float2 (&a)[NP])
{
    if constexpr (ORD == 0 && NP == 4) return make_float2((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y));
    else if constexpr (ORD == 0 && NP == 2) return make_float2(a[0].x + a[1].x, a[0].y + a[1].y);
    else if constexpr (ORD == 1 && NP == 8) {
        float2 q[4];
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = make_float2(a[l].x + a[l + 4].x, a[l].y + a[l + 4].y);
        return make_float2((q[0].x + q[1].x) + (q[2].x + q[3].x), (q[0].y + q[1].y) + (q[2].y + q[3].y));
    } else {
        static_assert(ORD == 1 && NP == 4, "unknown summation order");
        const float2 q0 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), q1 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y);
        return make_float2(q0.x + q1.x, q0.y + q1.y);
    }
}

// FULL: the tile is a whole one (all SPAN input samples exist, all OUTS outputs are wanted) and no Cross outputs are computed
// in place: no ragged-end loader, no seam arithmetic, unconditional stores.  The same instructions for the common case, but
// the slow paths no longer shape the register allocation and the epilogue of the fast one (alternating A/B inside one process,
// tools/full_tiles_ab.py: decimate stage -1.8 %).
template <int D, int P, int R, int NT, bool U8, int TC, bool GUARD, int NP, int ORD, int PSKIP, bool FULL>
__device__ __forceinline__ void decimate_c4_tile(int tile, const void* __restrict__ in, int64_t x0, int count, const float* __restrict__ taps,
                                                 float* __restrict__ out, int p_eff, int inl_seam, int inl_r0)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);

    const int out0 = tile * T::OUTS;
    const int64_t s0 = (int64_t)out0 * D;                     // first sample of the tile, relative to x0
    {
        // all of the tile's global loads in flight at once, then one wait
        if constexpr (FULL) {
            uint4 regs[Stage<T, U8, NT>::PER];
            Stage<T, U8, NT>::load_whole(in, x0 + s0, regs);
            Stage<T, U8, NT>::store_regs(regs, lds);
        } else {
            Stage<T, U8, NT> st;
            const int64_t total_avail = (int64_t)(count - 1) * D + (GUARD ? p_eff : P); // samples that exist from x0 on
            const int64_t av = total_avail - s0;
            st.load(in, x0 + s0, av > T::SPAN ? T::SPAN : (int)av);
            st.store(lds);
        }
    }
    __syncthreads();

    // per-thread window starts at sample tid*CHUNK of the tile
    const float2* win = lds + T::lds_idx(threadIdx.x * T::CHUNK);
    float2 acc[R][NP];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < NP; k++) acc[r][k] = make_float2(0.0f, 0.0f);

    static_assert(PSKIP == 0 || (U8 && !GUARD), "skipping zero taps needs finite samples");
    mac_window<D, P, R, T, TC, GUARD, NP, PSKIP>(win, taps, acc, GUARD ? p_eff / TC : 0);

    const int o = out0 + threadIdx.x * R;
    float2 res[R];
#pragma unroll
    for (int r = 0; r < R; r++) res[r] = fold_partials<NP, ORD>(acc[r]);
    if (!FULL && inl_seam > 0) {
        // Small launches (one host block per push): the seam fix-up is not worth a launch of its own.  An output whose window
        // straddles a multiple of inl_seam samples is recomputed right here in the reference's sequential order
        // (decimateCrossHighLevel, FilterInternal.hs:397-402) from the same LDS tile.  rt = the tile's first window start
        // inside its buffer (one 64-bit modulo per workgroup); a tile spans less than one buffer (the launcher checks).
        const int plen = GUARD ? p_eff : P;
        const int rt = (int)(((int64_t)inl_r0 + s0) % inl_seam);
        bool cross[R];
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            int rr = rt + (threadIdx.x * R + r) * D;
            if (rr >= inl_seam) rr -= inl_seam;
            cross[r] = rr + plen > inl_seam;
            any |= cross[r];
        }
        if (any) inline_cross_outputs<D, R, T, TC, GUARD>(win, taps, plen, cross, res);
    }
    if (R % 2 == 0 && (FULL || o + R <= count)) {
        float4* dst = reinterpret_cast<float4*>(out + 2 * (int64_t)o);
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) dst[r / 2] = make_float4(res[r].x, res[r].y, res[r + 1].x, res[r + 1].y);
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (o + r < count) *reinterpret_cast<float2*>(out + 2 * (int64_t)(o + r)) = res[r];
    }
}

// MIXED = false: every tile through the general body.  MIXED = true: tiles [0, nfull) -- whole ones, in a launch without in-place
// Cross outputs -- take the FULL body, the remainder (the ragged end) the general one, in the SAME launch (a second launch for the
// last tile or two costs 9-10 us of pure latency behind a 0.2-0.7 ms kernel).
template <int D, int P, int R, int NT, bool U8, int TC = ((P % 8 == 0) ? 8 : 4), bool GUARD = false, int NP = 4, int ORD = 0, int PSKIP = 0,
          bool MIXED = false>
__global__ void __launch_bounds__(NT) k_decimate_c4(const void* __restrict__ in, int64_t x0 /* sample index of output 0's window in `in` */,
                                                    int count, const float* __restrict__ taps, float* __restrict__ out,
                                                    int p_eff /* GUARD: taps of the filter (multiple of TC, <= P); else unused */,
                                                    int inl_seam /* > 0: compute the Cross outputs of buffers this long HERE */,
                                                    int inl_r0 /* window start of output 0 inside its buffer */,
                                                    int nfull /* MIXED: tiles [0, nfull) are whole */)
{
    using T = Tile<D, P, R, NT>;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Within every group
    // of 64 consecutive tiles XCD x takes the 8 consecutive tiles [8x, 8x+8), so 7 of 8 tile-to-tile
    // overlaps (120 samples each) hit in the SAME L2, while all XCDs still stream through the same
    // ~2 MB neighbourhood of HBM (giving every XCD its own far-apart eighth of the buffer measured
    // 20 % slower: DRAM locality matters more than the 3 % of re-reads).
    const int ntiles = (count + T::OUTS - 1) / T::OUTS;
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    if constexpr (MIXED) {
        if (tile < nfull) {
            decimate_c4_tile<D, P, R, NT, U8, TC, GUARD, NP, ORD, PSKIP, true>(tile, in, x0, count, taps, out, p_eff, 0, 0);
            return;
        }
    }
    decimate_c4_tile<D, P, R, NT, U8, TC, GUARD, NP, ORD, PSKIP, false>(tile, in, x0, count, taps, out, p_eff, inl_seam, inl_r0);
}

// Cross outputs: sequential order over the Lp plain taps (decimateCrossHighLevel,
// FilterInternal.hs:397-402).  The <= ceil((Lp-1)/D) straddlers of one seam have
// windows that overlap almost entirely, so a group of PER threads stages their union
// (Lp + (PER-1)*D samples, converted once) in LDS with coalesced loads and each
// thread then walks its own window.  SPW seams per workgroup.
// layout of the staged union: one float2 of padding after every 8 samples (candidate c then starts at 9c float2 = 18c
// dwords: 16 distinct bank pairs), rows 16 (mod 32) float2 apart (the two seams of a 32-lane group
// land on complementary banks): ds_read_b64 conflict-free.
template <int D, int LP, int PER>
constexpr int crossfix_row_float2() { return ((LP + (PER - 1) * D + (LP + (PER - 1) * D) / 8 + 31) / 32) * 32 + 16; }

// The work of ONE workgroup of PER * SPW threads: seams [wg * SPW, wg * SPW + SPW) of the launch, `lds` = SPW rows of
// crossfix_row_float2 float2.  A kernel of its own below (k_decimate_c_crossfix); round 6: also the body of the fix-up
// workgroups interleaved into the systolic decimator's launch (kernels_systolic.hip).
template <bool U8, int D, int LP, int PER, int SPW, bool RT = false>
__device__ __forceinline__ void decimate_c_crossfix_wg(float2* __restrict__ lds, int wg, const Geom& g, const float* __restrict__ xtaps,
                                                       const void* __restrict__ in, float* __restrict__ out, int64_t first_seam, int nseams)
{
    // RT: the filter has g.Lp <= LP taps (run-time); the staging layout is still the one of LP taps
    const int lp = RT ? g.Lp : LP;
    static_assert(D == 8 && PER == 16 && LP <= 128, "LDS layout below is worked out for 16 candidate slots 8 samples apart");
    constexpr int UNI = LP + (PER - 1) * D;          // samples in the union of one seam's windows
    constexpr int ROW = crossfix_row_float2<D, LP, PER>();
    static_assert(UNI <= PER * SPW, "one staging pass per seam");
    const int tid = threadIdx.x;
    const int seam0 = wg * SPW;
    const int64_t lo = g.k_begin * D - g.in_base, hi = (g.k_begin + g.count - 1) * (int64_t)D + lp - g.in_base;
    {
        // staging: the PER lanes of a seam load its union with 16-byte vectors, all seams of the
        // workgroup at once (one HBM round trip).  The union starts at a multiple of 8 samples, i.e.
        // 16-byte aligned like the tiles of the main kernel.
        const int sl = tid / PER, lane = tid - sl * PER;
        const int si = seam0 + sl;
        if (si < nseams) {
            const int64_t edge = (first_seam + si) * g.seamBI;
            const int64_t m_hi = (edge + D - 1) / D - 1;                 // last output starting before the edge
            const int64_t u0 = (m_hi - (PER - 1)) * D - g.in_base;       // first sample of the union, relative to `in`
            constexpr int SPV = U8 ? 8 : 2;                              // samples per 16-byte vector
            constexpr int NV = (UNI + SPV - 1) / SPV;
            float2* row = lds + sl * ROW;
            // every 16-byte load of the lane in flight before the first conversion / LDS store: the kernel is a few
            // microseconds of latency, a load-store loop would pay one HBM round trip per iteration
            constexpr int NIT = (NV + PER - 1) / PER;
            uint4 raw[NIT];
            bool whole[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int v = lane + it * PER;
                const int64_t idx = u0 + (int64_t)v * SPV;
                // vectors that poke outside the launch's own windows only feed discarded candidates
                whole[it] = v < NV && idx >= lo && idx + SPV <= hi;
                raw[it] = make_uint4(0u, 0u, 0u, 0u);
                if (whole[it]) {
                    if constexpr (U8) raw[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + 2 * idx);
                    else raw[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(in) + 2 * idx);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int v = lane + it * PER;
                if (v >= NV) continue;
                const int64_t idx = u0 + (int64_t)v * SPV;
                float2 smp[SPV];
                if (whole[it]) {
                    const uint4 q = raw[it];
                    if constexpr (U8) {
                        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            smp[2 * k] = make_float2(((float)(w[k] & 0xff) - 128.0f) * (1.0f / 128.0f),
                                                     ((float)((w[k] >> 8) & 0xff) - 128.0f) * (1.0f / 128.0f));
                            smp[2 * k + 1] = make_float2(((float)((w[k] >> 16) & 0xff) - 128.0f) * (1.0f / 128.0f),
                                                         ((float)(w[k] >> 24) - 128.0f) * (1.0f / 128.0f));
                        }
                    } else {
                        smp[0] = make_float2(__uint_as_float(q.x), __uint_as_float(q.y));
                        smp[1] = make_float2(__uint_as_float(q.z), __uint_as_float(q.w));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPV; k++) {
                        const int64_t ik = idx + k;
                        float2 t = make_float2(0.0f, 0.0f);
                        if (ik >= lo && ik < hi) {
                            if constexpr (U8) {
                                const uchar2 u = *reinterpret_cast<const uchar2*>(reinterpret_cast<const uint8_t*>(in) + 2 * ik);
                                t = make_float2(((float)u.x - 128.0f) * (1.0f / 128.0f), ((float)u.y - 128.0f) * (1.0f / 128.0f));
                            } else {
                                t = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(in) + 2 * ik);
                            }
                        }
                        smp[k] = t;
                    }
                }
#pragma unroll
                for (int k = 0; k < SPV; k++) {
                    const int so = v * SPV + k;
                    if (so < UNI) row[so + so / 8] = smp[k];
                }
            }
        }
    }
    __syncthreads();
    const int sl = tid / PER, ci = tid - sl * PER;
    const int si = seam0 + sl;
    if (si >= nseams) return;
    const int64_t edge = (first_seam + si) * g.seamBI;
    const int64_t m_hi = (edge + D - 1) / D - 1;
    const int64_t m = m_hi - (PER - 1) + ci;
    if (m < g.k_begin || m >= g.k_begin + g.count) return;
    const int64_t vm = m * D;
    if (!(vm < edge && vm + lp > edge)) return;
    const float2* w = lds + sl * ROW + ci * (D + 1);
    float re = 0.0f, im = 0.0f;
    if constexpr (RT) {
        for (int j = 0; j < lp; j++) {
            const float2 x = w[j + j / 8];
            const float h = xtaps[j];
            re = re + x.x * h;
            im = im + x.y * h;
        }
    } else if constexpr (LP % 16 == 0) {
        // blocks of 16 taps, the next block's LDS reads and tap chunk issued BEFORE this block's MACs (the fence pins that
        // order): left to itself the compiler reads four samples, waits, uses them -- 32 exposed LDS latencies and a full
        // lgkmcnt drain per tap chunk in a kernel that is nothing but latency
        typedef float cf16 __attribute__((ext_vector_type(16)));
        auto chunk = [&](int b) -> cf16 {
            typedef const __attribute__((address_space(4))) cf16* cp;
            uint64_t a = reinterpret_cast<uint64_t>(xtaps) + 64u * (uint32_t)b;
            asm volatile("" : "+s"(a));
            return *reinterpret_cast<cp>(a);
        };
        float2 xs[2][16];
        cf16 tc[2];
        tc[0] = chunk(0);
#pragma unroll
        for (int i = 0; i < 16; i++) xs[0][i] = w[i + i / 8];
#pragma unroll
        for (int b = 0; b < LP / 16; b++) {
            if (b + 1 < LP / 16) {
                tc[(b + 1) & 1] = chunk(b + 1);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int j = 16 * (b + 1) + i;
                    xs[(b + 1) & 1][i] = w[j + j / 8];
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float h = tc[b & 1][i];
                re = re + xs[b & 1][i].x * h;
                im = im + xs[b & 1][i].y * h;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < LP; j++) {
            const float2 x = w[j + j / 8];
            const float h = xtaps[j];
            re = re + x.x * h;
            im = im + x.y * h;
        }
    }
    *reinterpret_cast<float2*>(out + 2 * (m - g.k_begin)) = make_float2(re, im);
}

template <bool U8, int D, int LP, int PER, int SPW, bool RT = false>
__global__ void __launch_bounds__(PER * SPW) k_decimate_c_crossfix(Geom g, const float* __restrict__ xtaps,
                                                                   const void* __restrict__ in, float* __restrict__ out,
                                                                   int64_t first_seam, int nseams)
{
    __shared__ float2 lds[SPW * crossfix_row_float2<D, LP, PER>()];
    decimate_c_crossfix_wg<U8, D, LP, PER, SPW, RT>(lds, (int)blockIdx.x, g, xtaps, in, out, first_seam, nseams);
}

// inline_cross: the kernel computes the Cross outputs itself (no fix-up launch); *inlined tells whether the geometry allowed
// it (a tile must span less than one buffer)
template <int D, int P, int R, int NT, bool U8, int TC = ((P % 8 == 0) ? 8 : 4), bool GUARD = false, int NP = 4, int ORD = 0, int PSKIP = 0>
void launch_c4(hipStream_t s, const Geom& g, const float* taps, const void* in, float* out, bool inline_cross = false,
               bool* inlined = nullptr)
{
    using T = Tile<D, P, R, NT>;
    // the dynamic-LDS attribute is per device: one flag per (instantiation, device); idempotent, a race only repeats the call
    static std::atomic<bool> attr_set[64];
    auto kern = k_decimate_c4<D, P, R, NT, U8, TC, GUARD, NP, ORD, PSKIP>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)T::LDS_BYTES);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    int tiles = (g.count + T::OUTS - 1) / T::OUTS;
    int grid = ((tiles + 63) / 64) * 64;       // whole groups of 64: the kernel permutes blockIdx -> tile within a group
    int64_t x0 = g.k_begin * D - g.in_base;
    int inl_seam = 0, inl_r0 = 0;
    if (inline_cross && g.seamBI >= (int64_t)T::OUTS * D + g.Lp && g.seamBI < (1 << 30)) {
        inl_seam = (int)g.seamBI;
        inl_r0 = (int)((g.k_begin * D) % g.seamBI);
    }
    if (inlined) *inlined = inl_seam > 0;
    // Large launches without in-kernel seams: whole tiles take the FULL body inside the same launch (MIXED kernel).  A tile is
    // whole when its OUTS outputs are wanted and its SPAN samples exist: the launch's samples end at (count - 1) * D + Lp, and
    // SPAN = (OUTS - 1) * D + P reaches past that of a shorter (guarded) filter's own length, hence one tile less there.
    if (inl_seam == 0 && g.count >= 128 * T::OUTS) {
        const int nfull = g.count / T::OUTS - ((GUARD && g.Lp < P) ? 1 : 0);
        static std::atomic<bool> attr_full[64];
        auto kmixed = k_decimate_c4<D, P, R, NT, U8, TC, GUARD, NP, ORD, PSKIP, true>;
        if (dev < 0 || dev >= 64 || !attr_full[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kmixed), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
            if (dev >= 0 && dev < 64) attr_full[dev] = true;
        }
        hipLaunchKernelGGL(kmixed, dim3(grid), dim3(NT), T::LDS_BYTES, s, in, x0, g.count, taps, out, g.Lp, 0, 0, nfull);
        return;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), T::LDS_BYTES, s, in, x0, g.count, taps, out, g.Lp, inl_seam, inl_r0, 0);
}


}  // namespace
}  // namespace sdrhip
