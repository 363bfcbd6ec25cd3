// kernels_chain.hip -- fast paths for the low-rate stages of the FM chain (they run
// at 1/8 .. 3/80 of the input rate, so they are latency/HBM-bound little kernels: the
// job here is to keep them off the critical path, not to chase a roofline).
//
//   K3  fmDemod             branch-free atan/atan2, 4 samples per thread
//   K4  3/10 resampler      LDS tile, one full polyphase cycle (3 outputs) per thread,
//                            taps wave-uniform (scalar loads)
//   K5  symmetric real FIR  LDS tile, 4 consecutive outputs per thread, pair-add first,
//                            optional fused gain (fm.hs:40)
//   + seam fix-up kernels (Cross outputs, sequential order) for real FIR and resampler
//
// Arithmetic contract as in kernels_generic.hip (-ffp-contract=off, lane orders of the
// AVX variants: resampleAVXRR resample.c:70-87, filterAVXSymmetricRR filter.c:60-68).
#include "kernels.hpp"
#include "crossfix.hpp"
#include "demod.hpp"

namespace sdrhip {

namespace {

// K3 stand-alone: the common case of fmDemod (demod.hpp: fm_phase_common_tbl, atanf's argument reduction looked up in an LDS table)
// for every lane, a wave vote, and the full select form behind it for a wave that holds anything else -- the fused loader's form.
// (Rounds 1-5 measured four other restatements of the same arithmetic -- ternaries, selects, the vote without the table, packed
// pairs -- all bit-equal and slower: tools/lab_variants/, LABNOTES.)
__device__ __forceinline__ float4 fm_phase_quad(float2 prev, float2 s0, float2 s1, float2 s2, float2 s3, const float* atbl)
{
    const float2 v[5] = {prev, s0, s1, s2, s3};
    float y[4];
    bool rare = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        bool q;
        y[e] = fm_phase_common_tbl(v[e + 1], v[e], q, atbl);
        rare |= q;
    }
    if (__any(rare)) {
#pragma unroll
        for (int e = 0; e < 4; e++) y[e] = fm_phase_sel(v[e + 1], v[e]);
    }
    return make_float4(y[0], y[1], y[2], y[3]);
}

__global__ void __launch_bounds__(256) k_fm_demod_fast(const float* __restrict__ in, float* __restrict__ out, int64_t count,
                                                        int has_prev, float last_re, float last_im, int out_vec)
{
    // 4 samples per thread: five 8-byte loads (the IQ stream is only 8-byte aligned in
    // general: it usually starts one sample into a buffer), one 16-byte store
    const float2* in2 = reinterpret_cast<const float2*>(in);
    const int64_t nquad = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
    atan_table_fill(atbl, threadIdx.x);
    __syncthreads();
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        // (round 4, measured and not kept: 16-byte loads for interior quads 0.164 ms against 0.159 for these five 8-byte loads per
        // 2^26 samples, non-temporal 0.185 -- the decimator's output is still partly in the last-level cache when this kernel reads it)
        const float2 s0 = in2[4 * q], s1 = in2[4 * q + 1], s2 = in2[4 * q + 2], s3 = in2[4 * q + 3];
        float2 prev;
        if (q > 0 || has_prev) prev = in2[4 * q - 1];
        else prev = make_float2(last_re, last_im);
        const float4 r = fm_phase_quad(prev, s0, s1, s2, s3, atbl);
        if (out_vec) {
            reinterpret_cast<float4*>(out)[q] = r;
        } else {
            out[4 * q] = r.x; out[4 * q + 1] = r.y; out[4 * q + 2] = r.z; out[4 * q + 3] = r.w;
        }
    }
    // tail (< 4 samples)
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t i = (nquad << 2) + threadIdx.x;
        const float2 cur = in2[i];
        float2 prev;
        if (i > 0 || has_prev) prev = in2[i - 1];
        else prev = make_float2(last_re, last_im);
        out[i] = fm_phase_sel(cur, prev);
    }
}

// ---------------------------------------------------------------------------
// K5  real FIR filters (D = 1), 8 lanes:
//   SYM  (filterAVXSymmetricRR, filter.c:60-68 -> avx_sym_dotprod_R common.h:181-201):
//        out[o] = tree8(a_l),  a_l = sum_{k = l, l+8, ..} c[k] * (x[o+k] + x[o+2n-1-k])   (pair-add FIRST)
//   !SYM (filterAVXRR, filter.c:36-46 -> avx_dotprod_R common.h:58-72):
//        a_l = sum_{k = l, l+8, ..} c[k] * x[o+k]
// nk = number of taps the kernel walks (n half-taps / all taps), a multiple of 8, run-time.
// One workgroup = NT*R consecutive outputs staged in (dynamic) LDS; one thread = R = 4
// consecutive outputs.  The taps are walked eight at a time in a ROLLED loop (unrolling it lets
// the scheduler hoist all LDS reads to the top: 130+ VGPRs, spills, half the occupancy);
// iteration j (k = 8j..8j+7) needs the front samples w[8j .. 8j+R+6] and, for SYM, the back
// samples w[2n-8-8j .. 2n-1-8j+R-1]: three 16-byte LDS reads each.
// ---------------------------------------------------------------------------
// L = 8: the AVX order; L = 4: the SSE order (filterSSERR / filterSSESymmetricRR, filter.c:24-34,48-58 -> sse_dotprod_R /
// sse_sym_dotprod_R, common.h:34-45,157-179: lane partial k & 3, tree (a0 + a1) + (a2 + a3)) -- the same walk, eight taps per
// step, four accumulators per output instead of eight.
template <bool SYM, int R, int NT, int L = 8>
__global__ void __launch_bounds__(NT) k_fir_real8_fast(const float* __restrict__ in, int64_t x0, int count,
                                                        const float* __restrict__ taps, int nk, float* __restrict__ out,
                                                        float gain, int apply_gain, int aligned)
{
    static_assert(R == 4, "windows assume 4 outputs per thread (16-byte aligned thread windows)");
    constexpr int OUTS = NT * R;
    const int full = SYM ? 2 * nk : nk;                 // filter length
    const int span = OUTS + full - 1;
    const int span4 = (span + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float* lds = lds_dyn;

    const int out0 = blockIdx.x * OUTS;
    const int64_t total_avail = (int64_t)count + full - 1;
    const int64_t av64 = total_avail - out0;
    const int avail = av64 > span ? span : (int)av64;
    const float* src = in + x0 + out0;
    const int nvec = span4 + 3;                                // + 3: the last window's reads run 12 floats past it
    if (aligned && av64 >= 4 * (int64_t)nvec && nvec <= 2 * NT) {      // the floats past the tile's span exist (they feed no output)
        // interior tile of a filter of up to ~250 taps: both 16-byte loads of a thread in flight at once (a load-store loop
        // pays one HBM round trip per iteration)
        const int v0 = threadIdx.x, v1 = threadIdx.x + NT;
        const float4 a = *reinterpret_cast<const float4*>(src + 4 * v0);
        float4 b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (v1 < nvec) b = *reinterpret_cast<const float4*>(src + 4 * v1);
        *reinterpret_cast<float4*>(&lds[4 * v0]) = a;
        if (v1 < nvec) *reinterpret_cast<float4*>(&lds[4 * v1]) = b;
    } else {
        for (int v = threadIdx.x; v < nvec; v += NT) {
            const int s = 4 * v;
            float4 val;
            if (aligned && s + 3 < avail) {
                val = *reinterpret_cast<const float4*>(src + s);
            } else {
                val.x = s + 0 < avail ? src[s + 0] : 0.0f;
                val.y = s + 1 < avail ? src[s + 1] : 0.0f;
                val.z = s + 2 < avail ? src[s + 2] : 0.0f;
                val.w = s + 3 < avail ? src[s + 3] : 0.0f;
            }
            *reinterpret_cast<float4*>(&lds[s]) = val;
        }
    }
    __syncthreads();

    typedef float f8v __attribute__((ext_vector_type(8)));
    const float* win = lds + threadIdx.x * R;
    static_assert(L == 8 || L == 4, "AVX or SSE lane count");
    float acc[R][L];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int l = 0; l < L; l++) acc[r][l] = 0.0f;
    // The taps are fetched ONE STEP AHEAD (a scalar-load wait is a full lgkmcnt(0) drain: with the load issued a whole step
    // of arithmetic earlier it finds the data there); the loop stays rolled.
    f8v cnext = *reinterpret_cast<const f8v*>(taps);
#pragma unroll 1
    for (int j = 0; j < nk / 8; j++) {
        const f8v c8 = cnext;
        const float* fp = win + 8 * j;
        float fw[12], bw[12];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float4 a = *reinterpret_cast<const float4*>(fp + 4 * q);
            fw[4 * q] = a.x; fw[4 * q + 1] = a.y; fw[4 * q + 2] = a.z; fw[4 * q + 3] = a.w;
        }
        if constexpr (SYM) {
            const float* bp = win + 2 * nk - 8 - 8 * j;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * q);
                bw[4 * q] = b.x; bw[4 * q + 1] = b.y; bw[4 * q + 2] = b.z; bw[4 * q + 3] = b.w;
            }
        }
        {
            // Next step's taps, issued AFTER this step's LDS reads have been waited for (the empty asm consumes the last values
            // read: while a scalar load is outstanding every LDS wait is a full drain, so it must not be in flight beside them)
            // and a whole step of arithmetic before they are used (alternating A/B on 2^24 outputs: 89.5 against 90.7 us).
            if constexpr (SYM) asm volatile("" : : "v"(fw[11]), "v"(bw[11]));
            else asm volatile("" : : "v"(fw[11]));
            uint64_t a = reinterpret_cast<uint64_t>(taps) + 32u * (uint32_t)(j + 1 < nk / 8 ? j + 1 : j);
            asm volatile("" : "+s"(a));
            cnext = *reinterpret_cast<const __attribute__((address_space(4))) f8v*>(a);
        }
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                // k = 8j+kk: front w[r+k] = fw[r+kk]; back w[r+2n-1-k] = bw[r + 7 - kk]; lane k & (L-1) = kk & (L-1)
                if constexpr (SYM) acc[r][kk % L] = acc[r][kk % L] + c8[kk] * (fw[r + kk] + bw[r + 7 - kk]);
                else acc[r][kk % L] = acc[r][kk % L] + c8[kk] * fw[r + kk];
            }
        }
    }
    const int o = out0 + threadIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float res;
        if constexpr (L == 8) res = ((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3])) + ((acc[r][4] + acc[r][5]) + (acc[r][6] + acc[r][7]));
        else res = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        if (apply_gain) res = res * gain;
        if (o + r < count) out[o + r] = res;
    }
}

// ---------------------------------------------------------------------------
// Complex FIR filter (D = 1), real taps, AVX "RC" order (filterAVXRC, filter.c:106-114): 4 complex
// lanes over DUPLICATED taps (re uses taps[2k], im uses taps[2k+1], honoured as passed),
// out = (L0+L1)+(L2+L3).  Same tiling as above with float2 elements: R = 4 outputs per thread,
// 4 taps per rolled step (w[4j .. 4j+6]: four 16-byte LDS reads, one s_load_dwordx8 of taps).
// ---------------------------------------------------------------------------
template <int R, int NT>
__global__ void __launch_bounds__(NT) k_filter_cplx4_fast(const float* __restrict__ in, int64_t x0, int count,
                                                           const float* __restrict__ taps2, int P, float* __restrict__ out)
{
    static_assert(R == 4, "4 outputs per thread");
    constexpr int OUTS = NT * R;
    const int span = OUTS + P - 1;                        // complex samples
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float2* lds = reinterpret_cast<float2*>(lds_dyn);

    const int out0 = blockIdx.x * OUTS;
    const int64_t total_avail = (int64_t)count + P - 1;
    const int64_t av64 = total_avail - out0;
    const int avail = av64 > span ? span : (int)av64;
    const float2* src = reinterpret_cast<const float2*>(in) + x0 + out0;
    const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    const int nvec = (span + 1) / 2 + 4;                       // + 4: the last window's reads run 8 samples past it
    if (al && av64 >= 2 * (int64_t)nvec && nvec <= 3 * NT) {
        // interior tile of a filter of up to ~500 taps: every 16-byte load of a thread in flight before the first LDS store
        float4 val[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int v = threadIdx.x + i * NT;
            v = v < nvec ? v : nvec - 1;
            val[i] = *reinterpret_cast<const float4*>(src + 2 * v);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int v = threadIdx.x + i * NT;
            if (v < nvec) *reinterpret_cast<float4*>(&lds[2 * v]) = val[i];
        }
    } else {
        for (int v = threadIdx.x; v < nvec; v += NT) {
            const int s = 2 * v;
            float4 val;
            if (al && s + 1 < avail) {
                val = *reinterpret_cast<const float4*>(src + s);
            } else {
                const float2 a = s < avail ? src[s] : make_float2(0.0f, 0.0f);
                const float2 b = s + 1 < avail ? src[s + 1] : make_float2(0.0f, 0.0f);
                val = make_float4(a.x, a.y, b.x, b.y);
            }
            *reinterpret_cast<float4*>(&lds[s]) = val;
        }
    }
    __syncthreads();

    typedef float f8v __attribute__((ext_vector_type(8)));
    const float2* win = lds + threadIdx.x * R;
    float2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int l = 0; l < 4; l++) acc[r][l] = make_float2(0.0f, 0.0f);
    // sliding 8-sample window: w[0..3] carried from the previous step, w[4..7] read fresh (two 16-byte reads)
    float2 w[8];
    {
        const float4 a = *reinterpret_cast<const float4*>(win);
        const float4 b = *reinterpret_cast<const float4*>(win + 2);
        w[4] = make_float2(a.x, a.y); w[5] = make_float2(a.z, a.w);
        w[6] = make_float2(b.x, b.y); w[7] = make_float2(b.z, b.w);
    }
#pragma unroll 1
    for (int j = 0; j < P / 4; j++) {
        const f8v c8 = *reinterpret_cast<const f8v*>(taps2 + 8 * j);     // (re,im) tap pairs of taps 4j..4j+3
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = w[q + 4];
        {
            const float4 a = *reinterpret_cast<const float4*>(win + 4 * j + 4);
            const float4 b = *reinterpret_cast<const float4*>(win + 4 * j + 6);
            w[4] = make_float2(a.x, a.y); w[5] = make_float2(a.z, a.w);
            w[6] = make_float2(b.x, b.y); w[7] = make_float2(b.z, b.w);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                acc[r][kk].x = acc[r][kk].x + c8[2 * kk] * w[r + kk].x;
                acc[r][kk].y = acc[r][kk].y + c8[2 * kk + 1] * w[r + kk].y;
            }
        }
    }
    const int o = out0 + threadIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float2 res = make_float2((acc[r][0].x + acc[r][1].x) + (acc[r][2].x + acc[r][3].x),
                                       (acc[r][0].y + acc[r][1].y) + (acc[r][2].y + acc[r][3].y));
        if (o + r < count) *reinterpret_cast<float2*>(out + 2 * (int64_t)(o + r)) = res;
    }
}

// The filter case (D == 1, LP taps): the LP-1 straddlers of a seam have windows one sample apart;
// one workgroup stages their union (2*LP - 2 floats, coalesced) in LDS and thread c walks window c.
template <int LP>
__global__ void __launch_bounds__(LP) k_filter_real_crossfix_lds(Geom g, const float* __restrict__ xtaps,
                                                                  const float* __restrict__ in, float* __restrict__ out,
                                                                  int64_t first_seam, float gain, int apply_gain)
{
    constexpr int UNI = 2 * LP - 2;
    __shared__ float lds[2 * LP];
    const int tid = threadIdx.x;
    const int64_t edge = (first_seam + blockIdx.x) * g.seamBI;
    const int64_t v0 = edge - (LP - 1);                               // first straddler starts here
    const int64_t lo = g.k_begin - g.in_base, hi = g.k_begin + g.count - 1 + LP - g.in_base;
    for (int e = tid; e < UNI; e += LP) {
        const int64_t idx = v0 + e - g.in_base;
        lds[e] = (idx >= lo && idx < hi) ? in[idx] : 0.0f;
    }
    __syncthreads();
    const int64_t m = v0 + tid;                                        // candidates v0 .. v0 + LP - 2
    if (tid >= LP - 1 || m < g.k_begin || m >= g.k_begin + g.count) return;
    float r = 0.0f;
#pragma unroll 16
    for (int j = 0; j < LP; j++) r = r + lds[tid + j] * xtaps[j];
    if (apply_gain) r = r * gain;
    out[m - g.k_begin] = r;
}

// ---------------------------------------------------------------------------
// K4  polyphase resampler, 8 lanes (resampleAVXRR, resample.c:70-87), specialised
// for NG polyphase groups of NLOOP (padded) taps.  The launch starts at an output
// whose group is 0, so thread t owns outputs 3t..3t+NG-1 = groups 0..NG-1 and every
// tap is wave-uniform.  inc[] (the per-group input increments) are compile-time.
// ---------------------------------------------------------------------------
// DEMOD: `in` is the decimator's complex output and the tile's inputs are demodulated on the way into LDS
// (y[p] = phase(d[p] * conj d[p-1]), fmDemod, Demod.hs:21-46) -- y never makes the round trip through HBM that the
// stand-alone fmDemod kernel pays 12 B per sample for (8 read + 4 written) and this kernel 4 more.  Only the y values
// other kernels still read are written out (y_out): those within `ykeep` of a multiple of `yseam` (the seam fix-up's
// windows).  in[2p] is y-input p of the launch; has_prev: in[-2..-1] exists (else the carried sample is 0, Demod.hs:41).
struct DemodSide {
    int has_prev;
    float* y_out;          // y_out[p] for p relative to the launch's first input
    int64_t y_abs0;        // absolute stream index of p = 0 (seams sit at absolute multiples of yseam)
    int yseam, ykeep;      // yseam = 0: nothing to keep
    // the launch's two edges: y[0, nedge) and y[y_count - nedge, y_count), which the lead / tail launches of the generic kernel and the
    // seam fix-up read and the tiles do not cover; the first and the last workgroup write them on their way out (round 4: they were
    // two launches, then one, of 4.8 us of pure launch latency in front of the tile kernel)
    int nedge;
    int64_t y_count;
};

// first workgroup: y[0, nedge); last workgroup: y[y_count - nedge, y_count) -- one sample per thread and round
__device__ __forceinline__ void demod_edges(const float* __restrict__ in, const DemodSide& dm)
{
    if (dm.nedge <= 0 || (blockIdx.x != 0 && blockIdx.x != gridDim.x - 1)) return;
    const float2* z = reinterpret_cast<const float2*>(in);
    for (int seg = 0; seg < 2; seg++) {
        if (blockIdx.x != (seg == 0 ? 0u : gridDim.x - 1)) continue;
        const int64_t p0 = seg == 0 ? 0 : dm.y_count - dm.nedge;
        for (int i = threadIdx.x; i < dm.nedge; i += blockDim.x) {
            const int64_t p = p0 + i;
            const float2 v[2] = {(p > 0 || dm.has_prev) ? z[p - 1] : make_float2(0.0f, 0.0f), z[p]};
            float y[1];
            fm_phase_voted<1>(v, y);
            dm.y_out[p] = y[0];
        }
    }
}

// PK (round 4): the 8 lane partials as four packed pairs -- v_pk_mul_f32 by an SGPR pair of taps + v_pk_add_f32, half the VALU
// instructions of the scalar walk.  A group whose window starts at an odd float (PRE = 7) pairs the partials (1,2) (3,4) (5,6) (7,0)
// instead, so that its sample pairs are the same aligned register pairs; taps 0 and 63 are then single operations.  Every partial
// still adds its products in increasing tap order from +0: same bits.
// __launch_bounds__(NT, 6) (HIP: minimum WAVES per SIMD): 78 VGPRs, six waves per SIMD -- fused kernel 0.232 -> 0.227 ms per pass (same-box A/B, 4 / 5 / 6)
// wave_shr:1 -- lane l takes lane l - 1's `src`; lane 0, which has no source, keeps `old`
__device__ __forceinline__ float dpp_shr1_or(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}

constexpr int kDemodCycles = 249;   // DEMOD: cycles per workgroup -- 2551 inputs + the predecessor + the 16-byte phase fit five rounds of 256 pairs exactly

// DEMOD loader (round 6; rounds 4-5 loaded every sample and its predecessor by two 8-byte loads, 22 per thread: tools/lab_variants/
// resample_loader_variants.hip): a wave owns 320 consecutive PAIRS of samples, round i = pairs [64 i, 64 i + 64) of them: one 16-byte
// load per pair (five per thread), the predecessor of a pair's first sample from the lane below (DPP wave_shr:1), for lane 0 from lane
// 63 of the round before (v_readlane) -- only round 0 loads it (a wave-uniform address).  The tile is 249 cycles: ten samples per
// thread, no ragged round.  Same bits; the fused stage 0.1927-0.1941 ms against 0.1965-0.2019 per 2^29-sample pass, alternating in one
// process (profiles/r06/k4_pair_loader_ab.txt) -- 1.7 %, and with that the kernel is declared final: it sits at 1.4 x its issue floor
// with its loads hidden behind other workgroups (LABNOTES), and fewer load instructions were the last thing left to take out.
template <int NG, int NLOOP, int INC0, int INC1, int INC2, int NT, bool DEMOD = false, int L = 8, bool PK = false>
__global__ void __launch_bounds__(NT, 6) k_resample3_fast(const float* __restrict__ in, int64_t pos0, int ncycles,
                                                        int64_t avail_total, const float* __restrict__ groups,
                                                        int row_stride, float* __restrict__ out, DemodSide dm)
{
    static_assert(NG == 3, "specialised for three polyphase groups");
    static_assert(!DEMOD || NT == 256, "the fused fmDemod loader is written for four waves");
    constexpr int PERIOD = INC0 + INC1 + INC2;
    constexpr int PRE[3] = {0, INC0, INC0 + INC1};
    constexpr int WIN = PRE[2] + NLOOP;                 // floats one thread reads
    constexpr int CYC = DEMOD ? kDemodCycles : NT;      // polyphase cycles per workgroup
    constexpr int SPAN = (CYC - 1) * PERIOD + WIN;      // floats one workgroup reads
    constexpr int SPAN4 = (SPAN + 3) / 4;
    // DEMOD: y of sample p (relative to the tile's first input) lives at lds[2 + p], p = -2 .. 2557: every pair is stored unguarded
    constexpr int LOFF = DEMOD ? 2 : 0;
    __shared__ __attribute__((aligned(16))) float lds[DEMOD ? 2564 : SPAN4 * 4 + 4];

    const int cyc0 = blockIdx.x * CYC;
    const int64_t base = pos0 + (int64_t)cyc0 * PERIOD;  // first input of this workgroup, relative to `in`
    const int64_t av64 = avail_total - (int64_t)cyc0 * PERIOD;
    const int avail = av64 > SPAN ? SPAN : (int)av64;
    if constexpr (DEMOD) {
        static_assert(SPAN <= 4096 && SPAN + 9 <= 2560, "five rounds of 256 pairs cover the tile whatever the 16-byte phase");
        constexpr int R = 5;
        const float2* z = reinterpret_cast<const float2*>(in) + base;
        const int m0 = dm.yseam > 0 ? (int)((dm.y_abs0 + base) % dm.yseam) : 0;
        // first pair: starts at the even (16-byte aligned) sample at or below the tile's predecessor sample base - 1
        const int dd = 1 + (int)(((reinterpret_cast<uintptr_t>(in) >> 3) + (uint64_t)(base - 1)) & 1);     // tile input 0 is sample dd of the pair stream
        const bool interior = av64 >= SPAN + 12 && base >= 4;
        __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
        if (interior) {
            atan_table_fill(atbl, threadIdx.x);
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
            const int j0 = 64 * R * wv + ln;                          // this thread's pair in round 0
            const char* src = reinterpret_cast<const char*>(z - dd);  // sample 0 of the pair stream
            float4 v[R];
#pragma unroll
            for (int i = 0; i < R; i++) v[i] = *reinterpret_cast<const float4*>(src + 16u * (unsigned)(j0 + 64 * i));
            // the wave's first predecessor: a wave-uniform address that must NOT become a scalar load (SMEM returns out of order: with an
            // s_load of HBM data outstanding every LDS wait below would be lgkmcnt(0)); the offset goes through a VGPR
            unsigned po = 16u * (unsigned)(64 * R * wv);
            asm volatile("" : "+v"(po));
            const float2 pl = *reinterpret_cast<const float2*>(src + po - 8);
            auto pred = [&](int i) {
                float2 o = pl;
                if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i - 1].z), 63)),
                                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i - 1].w), 63)));
                return make_float2(dpp_shr1_or(o.x, v[i].z), dpp_shr1_or(o.y, v[i].w));
            };
            float2 y[R];
            bool rare = false;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the table; `interior` is uniform over the workgroup
#pragma unroll
            for (int i = 0; i < R; i++) {
                const float2 A = make_float2(v[i].x, v[i].y), B = make_float2(v[i].z, v[i].w);
                const float2 pv = pred(i);
                bool q0, q1;
                y[i].x = fm_phase_common_tbl(A, pv, q0, atbl);
                __builtin_amdgcn_sched_barrier(0);
                y[i].y = fm_phase_common_tbl(B, A, q1, atbl);
                rare |= q0 | q1;
                __builtin_amdgcn_sched_barrier(0);
            }
            float* yl = lds + LOFF - dd;                               // yl[s] = y of pair-stream sample s
            if (dd == 2) {
#pragma unroll
                for (int i = 0; i < R; i++) *reinterpret_cast<float2*>(yl + 2 * (j0 + 64 * i)) = y[i];
            } else {
#pragma unroll
                for (int i = 0; i < R; i++) { yl[2 * (j0 + 64 * i)] = y[i].x; yl[2 * (j0 + 64 * i) + 1] = y[i].y; }
            }
            if (__any(rare)) {
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const float2 A = make_float2(v[i].x, v[i].y), B = make_float2(v[i].z, v[i].w);
                    const float2 pv = pred(i);
                    y[i] = make_float2(fm_phase_sel(A, pv), fm_phase_sel(B, A));
                    yl[2 * (j0 + 64 * i)] = y[i].x;
                    yl[2 * (j0 + 64 * i) + 1] = y[i].y;
                }
            }
            if (dm.yseam > 0 && (m0 < dm.ykeep || m0 + SPAN > dm.yseam - dm.ykeep)) {
#pragma unroll
                for (int i = 0; i < R; i++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int p = 2 * (j0 + 64 * i) + h - dd;
                        int m = m0 + p;
                        if (m >= dm.yseam) m -= dm.yseam;
                        if (p >= 0 && p < SPAN && (m < dm.ykeep || m >= dm.yseam - dm.ykeep)) dm.y_out[base + p] = h ? y[i].y : y[i].x;
                    }
                }
            }
        } else {
            for (int p = threadIdx.x; p < SPAN; p += NT) {
                float y = 0.0f;
                if (p < av64) {
                    const float2 c = z[p];
                    const float2 pv = (base + p > 0 || dm.has_prev) ? z[p - 1] : make_float2(0.0f, 0.0f);
                    y = fm_phase_sel(c, pv);
                    if (dm.yseam > 0) {
                        int m = m0 + p;
                        if (m >= dm.yseam) m -= dm.yseam;
                        if (m < dm.ykeep || m >= dm.yseam - dm.ykeep) dm.y_out[base + p] = y;
                    }
                }
                lds[LOFF + p] = y;
            }
        }
    } else {
    const float* src = in + base;
    const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    // all of the tile's global loads in flight at once, then one wait: a load-store loop pays one HBM round trip per
    // iteration (2.6 of them here), and those round trips, not the arithmetic, were what a workgroup's life consisted of
    constexpr int NV = (SPAN4 + NT - 1) / NT;
    float4 val[NV];
    if (al && avail >= 4 * SPAN4) {
        // interior tile (all but the last workgroup of an aligned launch): branch-free 16-byte loads
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = threadIdx.x + i * NT;
            val[i] = v < SPAN4 ? *reinterpret_cast<const float4*>(src + 4 * v) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int s = 4 * (threadIdx.x + i * NT);
            val[i].x = s + 0 < avail ? src[s + 0] : 0.0f;
            val[i].y = s + 1 < avail ? src[s + 1] : 0.0f;
            val[i].z = s + 2 < avail ? src[s + 2] : 0.0f;
            val[i].w = s + 3 < avail ? src[s + 3] : 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int v = threadIdx.x + i * NT;
        if (v < SPAN4) *reinterpret_cast<float4*>(&lds[4 * v]) = val[i];
    }
    }
    __syncthreads();

    const int cyc = cyc0 + threadIdx.x;
    if ((int)threadIdx.x >= CYC || cyc >= ncycles) {
        if constexpr (DEMOD) demod_edges(in, dm);
        return;
    }
    // window start = 10*t floats: 8-byte aligned, and a 10-dword lane stride is conflict-free
    // for ds_read_b64 (distinct even banks within each 32-lane group)
    static_assert(PERIOD % 2 == 0, "8-byte aligned thread windows");
    const float* win = lds + LOFF + threadIdx.x * PERIOD;
    float res[3];
    if constexpr (PK && L == 8) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        static_assert(NLOOP % 8 == 0 && WIN % 2 == 1, "pairs below assume an even tap count and windows of 7 + NLOOP floats");
        f2 W2[(WIN + 1) / 2];
#pragma unroll
        for (int i = 0; i < (WIN + 1) / 2; i++) {
            const float2 q = *reinterpret_cast<const float2*>(win + 2 * i);
            W2[i] = f2{q.x, q.y};
        }
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const float* c = groups + g * row_stride;
            float acc[8];
            {
                if (PRE[g] % 2 == 0) {
                    f2 A[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) A[p] = f2{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < NLOOP; j += 2) A[(j % 8) / 2] = A[(j % 8) / 2] + W2[(PRE[g] + j) / 2] * f2{c[j], c[j + 1]};
#pragma unroll
                    for (int p = 0; p < 4; p++) { acc[2 * p] = A[p].x; acc[2 * p + 1] = A[p].y; }
                } else {
                    // B[p] = (partial 2p + 1, partial 2p + 2 mod 8): B[3] = (partial 7, partial 0)
                    f2 B[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) B[p] = f2{0.0f, 0.0f};
                    B[3].y = 0.0f + c[0] * W2[PRE[g] / 2].y;                               // tap 0: sample PRE is the high half of its pair
#pragma unroll
                    for (int j = 1; j + 1 < NLOOP; j += 2) B[((j % 8) - 1) / 2] = B[((j % 8) - 1) / 2] + W2[(PRE[g] + j) / 2] * f2{c[j], c[j + 1]};
                    B[3].x = B[3].x + c[NLOOP - 1] * W2[(PRE[g] + NLOOP - 1) / 2].x;       // tap NLOOP - 1: the low half of the last pair
                    acc[0] = B[3].y; acc[7] = B[3].x;
#pragma unroll
                    for (int p = 0; p < 3; p++) { acc[2 * p + 1] = B[p].x; acc[2 * p + 2] = B[p].y; }
                }
            }
            res[g] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
    } else {
    float w[WIN + 1];
#pragma unroll
    for (int i = 0; i < (WIN + 1) / 2; i++) {
        const float2 q = *reinterpret_cast<const float2*>(win + 2 * i);
        w[2 * i] = q.x;
        w[2 * i + 1] = q.y;
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float* c = groups + g * row_stride;
        // L = 8: the AVX order; L = 4: the SSE order (resampleSSERR, resample.c:52-68 -> sse_dotprod_R, common.h:34-50)
        float acc[L];
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = 0.0f;
#pragma unroll
        for (int j = 0; j < NLOOP; j++) acc[j % L] = acc[j % L] + c[j] * w[PRE[g] + j];
        if constexpr (L == 8) res[g] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        else res[g] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    }
    // one 12-byte store per thread: the wave writes 768 contiguous bytes
    struct __attribute__((packed, aligned(4))) f3 { float a, b, c; };
    f3 v = {res[0], res[1], res[2]};
    *reinterpret_cast<f3*>(out + (int64_t)cyc * 3) = v;
    if constexpr (DEMOD) demod_edges(in, dm);
}


// Everything of a 3/10 launch that is not a whole polyphase cycle, in ONE launch behind the tile kernel (round 6; they were three: a
// lead-in launch of up to two outputs, a tail launch of up to two, the seam fix-up): workgroups [0, nfixwg) rewrite the Cross outputs
// (resample_real_crossfix_wg), the last workgroup computes the lead-in and tail outputs, one thread each, in the SIMD lane order
// (resample.c:70-87) -- unless such an output is itself Cross: the seam workgroup owns it.
struct Stragglers {
    int lead, tail;            // outputs before the first whole cycle / after the last
    int lead_group0;           // polyphase group of output 0
    int64_t lead_pos[2];       // first input of lead output i, relative to `in`
    int64_t tail_pos0;         // first input of the tail's cycle (its outputs are groups 0, 1 at + 0, + 4)
    int done;                  // outputs in front of the tail
    int nloop, row_stride;
};

template <int PER, int UNI, int L>
__global__ void __launch_bounds__(256) k_resample3_stragglers(Geom g, Stragglers st, const float* __restrict__ groups, const float* __restrict__ plain,
                                                               int ntaps, const float* __restrict__ in, float* __restrict__ out, int64_t first_seam,
                                                               int nseams, int64_t in_avail)
{
    const int nfixwg = (nseams + 7) / 8;
    if ((int)blockIdx.x < nfixwg) {
        resample_real_crossfix_wg<PER, UNI, 32>((int)blockIdx.x, g, plain, ntaps, in, out, first_seam, nseams, in_avail, 1.0f, 0);
        return;
    }
    const int tid = threadIdx.x;
    int o, group;
    int64_t pos;
    if (tid < st.lead) {
        o = tid;
        group = (st.lead_group0 + tid) % 3;
        pos = st.lead_pos[tid];
    } else if (tid >= 32 && tid - 32 < st.tail) {
        o = st.done + (tid - 32);
        group = tid - 32;
        pos = st.tail_pos0 + (tid - 32 == 0 ? 0 : 4);
    } else {
        return;
    }
    if (is_cross(g, g.k_begin + o)) return;
    const float* x = in + pos;
    const float* c = groups + (size_t)group * st.row_stride;
    float acc[L];
#pragma unroll
    for (int l = 0; l < L; l++) acc[l] = 0.0f;
    for (int j = 0; j < st.nloop; j += L) {
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = acc[l] + c[j + l] * x[j + l];
    }
    if constexpr (L == 8) out[o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    else out[o] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

}  // namespace

void launch_fm_demod_fast(hipStream_t s, const float* d_in_iq, float* d_out, int64_t count, bool has_prev, float last_re,
                          float last_im)
{
    if (count <= 0) return;
    const int out_vec = ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0) ? 1 : 0;
    int64_t blocks = ((count >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    // grid-stride over at most this many workgroups (measured on 2^26 samples: 1024 / 2048 / 4096 / 8192 / 16384 / 65536 workgroups
    // 199 / 188 / 177 / 174 / 171 / 176 us)
    constexpr int64_t cap = 256 * 64;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_fm_demod_fast, dim3((int)blocks), dim3(256), 0, s, d_in_iq, d_out, count, has_prev ? 1 : 0, last_re, last_im, out_vec);
}

bool launch_fir_real8_fast(hipStream_t s, const Geom& g, bool sym, const float* d_taps, int nk, const float* d_cross_taps,
                          const float* d_in, float* d_out, float gain, bool apply_gain, int lanes)
{
    if (lanes != 8 && lanes != 4) return false;
    // filters only (D == 1), 8 lanes, tap count a multiple of 8 (the AVX constructors guarantee it)
    if (g.I != 1 || g.D != 1 || nk < 8 || nk % 8 != 0 || nk > 4096 || g.count <= 0 || g.seamBI < 0) return false;
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    constexpr int R = 4, NT = 256;
    const int full = sym ? 2 * nk : nk;
    const int64_t x0 = g.k_begin - g.in_base;
    const int aligned = ((reinterpret_cast<uintptr_t>(d_in + x0) & 15) == 0) ? 1 : 0;
    const int tiles = (g.count + NT * R - 1) / (NT * R);
    const size_t lds_bytes = ((size_t)(NT * R + full - 1 + 3) / 4 * 4 + 16) * sizeof(float);
    if (sym && lanes == 8)
        hipLaunchKernelGGL((k_fir_real8_fast<true, R, NT>), dim3(tiles), dim3(NT), lds_bytes, s, d_in, x0, g.count, d_taps, nk, d_out,
                           gain, apply_gain ? 1 : 0, aligned);
    else if (lanes == 8)
        hipLaunchKernelGGL((k_fir_real8_fast<false, R, NT>), dim3(tiles), dim3(NT), lds_bytes, s, d_in, x0, g.count, d_taps, nk, d_out,
                           gain, apply_gain ? 1 : 0, aligned);
    else if (sym)
        hipLaunchKernelGGL((k_fir_real8_fast<true, R, NT, 4>), dim3(tiles), dim3(NT), lds_bytes, s, d_in, x0, g.count, d_taps, nk, d_out,
                           gain, apply_gain ? 1 : 0, aligned);
    else
        hipLaunchKernelGGL((k_fir_real8_fast<false, R, NT, 4>), dim3(tiles), dim3(NT), lds_bytes, s, d_in, x0, g.count, d_taps, nk, d_out,
                           gain, apply_gain ? 1 : 0, aligned);
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            if (full == 128)
                hipLaunchKernelGGL((k_filter_real_crossfix_lds<128>), dim3(nseams), dim3(128), 0, s, g, d_cross_taps, d_in, d_out,
                                   first, gain, apply_gain ? 1 : 0);
            else if (full == 64)
                hipLaunchKernelGGL((k_filter_real_crossfix_lds<64>), dim3(nseams), dim3(64), 0, s, g, d_cross_taps, d_in, d_out,
                                   first, gain, apply_gain ? 1 : 0);
            else {
                const int per = full - 1;
                const int64_t total = (int64_t)nseams * per;
                hipLaunchKernelGGL(k_fir_real_crossfix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_cross_taps, d_in,
                                   d_out, first, nseams, per, gain, apply_gain ? 1 : 0);
            }
        }
    }
    return true;
}

bool launch_filter_cplx4_fast(hipStream_t s, const Geom& g, const float* d_dup_taps, int P, const float* d_cross_taps,
                              const float* d_in, float* d_out)
{
    if (g.I != 1 || g.D != 1 || P < 4 || P % 4 != 0 || P > 2048 || g.count <= 0 || g.seamBI < 0) return false;
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    constexpr int R = 4, NT = 256;
    const int64_t x0 = g.k_begin - g.in_base;
    const int tiles = (g.count + NT * R - 1) / (NT * R);
    const size_t lds_bytes = ((size_t)(NT * R + P - 1) + 16) * sizeof(float2);
    hipLaunchKernelGGL((k_filter_cplx4_fast<R, NT>), dim3(tiles), dim3(NT), lds_bytes, s, d_in, x0, g.count, d_dup_taps, P, d_out);
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = P - 1;
            const int64_t total = (int64_t)nseams * per;
            hipLaunchKernelGGL(k_fir_cplx_crossfix<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_cross_taps, d_in, d_out,
                               first, nseams, per);
        }
    }
    return true;
}

// d_iq != nullptr: fmDemod fused into the tile loader (k_resample3_fast<.., DEMOD>): d_iq[2p], d_iq[2p+1] is the decimator
// output whose phase step is input p of this launch (y_count of them), d_in is then the y BUFFER, which this call fills
// only where other kernels read it: the first and last kEdge inputs (stand-alone fmDemod launches: the few outputs before the
// first / after the last whole polyphase cycle) and the neighbourhood of every seam (written by the tile kernel).
bool launch_resample_3_10_fast(hipStream_t s, const Geom& g, const ResampTable& t, const int* increments,
                               const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out,
                               const float* d_iq, bool iq_has_prev, int64_t y_count, int lanes)
{
    if (!(lanes == 8 || (lanes == 4 && d_iq == nullptr && t.nloop == 64))) return false;
    constexpr int kEdge = 256, kKeep = 160;
    if (d_iq != nullptr) {
        // the fused form serves the FM chain's shape only; anything else: the caller demodulates first
        const int64_t yseam = g.seamBI > 0 ? g.seamBI / g.I : 0;
        if (!(t.nloop == 64 || t.nloop == 16) || g.I != 3 || y_count < 4 * kEdge || t.pos0 + 96 > kEdge) return false;
        if (g.seamBI > 0 && (g.seamBI % g.I != 0 || yseam < 4096 || yseam > (1 << 30))) return false;
    }
    // specialised for the FM chain's resampler: 3 groups, increments {4,3,3}, 64-float rows, AVX order
    if (t.ngroups != 3 || !(t.nloop == 64 || t.nloop == 16) || g.seamBI < 0 || t.force_seq) return false;
    if (!(increments[0] == 4 && increments[1] == 3 && increments[2] == 3)) return false;
    if (g.seamBI != 0 && d_plain_taps == nullptr) return false;
    if (g.count <= 0) return false;
    constexpr int NT = 256;
    // outputs before the first group-0 output and after the last whole cycle go to the generic kernel
    int lead = (3 - t.group0) % 3;
    if (lead > g.count) lead = g.count;
    const int ncycles = (g.count - lead) / 3;
    const int tail = g.count - lead - 3 * ncycles;
    if (d_iq != nullptr) {
        // the tail's windows must lie inside the last kEdge inputs
        const int64_t tail_pos = t.pos0 + (lead > 0 ? t.pre[lead - 1] + increments[(t.group0 + lead - 1) % 3] : 0) + (int64_t)ncycles * 10;
        if (tail_pos < y_count - kEdge + 16 || ncycles < 1) return false;
    }
    if (ncycles > 0) {
        // position of the first group-0 output relative to d_in
        int64_t pos = t.pos0 + (lead > 0 ? t.pre[lead - 1] + increments[(t.group0 + lead - 1) % 3] : 0);
        const int64_t avail_total = (int64_t)(ncycles - 1) * 10 + 7 + t.nloop;
        const int blocks = (ncycles + NT - 1) / NT;
        DemodSide dm = {};
        if (d_iq != nullptr) {
            dm.has_prev = iq_has_prev ? 1 : 0;
            dm.y_out = const_cast<float*>(d_in);
            dm.y_abs0 = g.in_base;
            dm.yseam = g.seamBI > 0 ? (int)(g.seamBI / g.I) : 0;
            dm.ykeep = kKeep;
            dm.nedge = kEdge;
            dm.y_count = y_count;
            // (workgroup size of the fused form, same-box A/B: 128 / 256 / 512 threads 0.197 / 0.196 / 0.2015 ms)
            if (t.nloop == 64)
                hipLaunchKernelGGL((k_resample3_fast<3, 64, 4, 3, 3, NT, true, 8, true>), dim3((ncycles + kDemodCycles - 1) / kDemodCycles), dim3(NT), 0, s,
                                   d_iq, pos, ncycles, avail_total, d_groups, t.row_stride, d_out + lead, dm);
            else        // 16-float groups: the reference example's own 31-tap resampler (examples/fm/Coeffs.hs)
                hipLaunchKernelGGL((k_resample3_fast<3, 16, 4, 3, 3, NT, true, 8, true>), dim3((ncycles + kDemodCycles - 1) / kDemodCycles), dim3(NT), 0, s,
                                   d_iq, pos, ncycles, avail_total, d_groups, t.row_stride, d_out + lead, dm);
        } else if (lanes == 4)
            hipLaunchKernelGGL((k_resample3_fast<3, 64, 4, 3, 3, NT, false, 4>), dim3(blocks), dim3(NT), 0, s, d_in, pos, ncycles, avail_total,
                               d_groups, t.row_stride, d_out + lead, dm);
        else if (t.nloop == 64)         // the packed-pair walk (round 4)
            hipLaunchKernelGGL((k_resample3_fast<3, 64, 4, 3, 3, NT, false, 8, true>), dim3(blocks), dim3(NT), 0, s, d_in, pos, ncycles, avail_total,
                               d_groups, t.row_stride, d_out + lead, dm);
        else
            hipLaunchKernelGGL((k_resample3_fast<3, 16, 4, 3, 3, NT>), dim3(blocks), dim3(NT), 0, s, d_in, pos, ncycles, avail_total,
                               d_groups, t.row_stride, d_out + lead, dm);
    }
    // (after the tile kernel: with fmDemod fused its first and last workgroup write the y the lead-in / tail outputs read)
    {
        int64_t first = 0, last = -1;
        if (g.seamBI != 0) seam_range(g, first, last);
        const int nseams = last >= first ? (int)(last - first + 1) : 0;
        if (nseams > 0 || lead > 0 || tail > 0) {
            Stragglers sg = {};
            sg.lead = lead;
            sg.tail = tail;
            sg.lead_group0 = t.group0;
            for (int i = 0; i < lead && i < 2; i++) sg.lead_pos[i] = t.pos0 + t.pre[i];
            sg.tail_pos0 = t.pos0 + (lead > 0 ? t.pre[lead - 1] + increments[(t.group0 + lead - 1) % 3] : 0) + (int64_t)ncycles * 10;
            sg.done = lead + 3 * ncycles;
            sg.nloop = t.nloop;
            sg.row_stride = t.row_stride;
            // Lp <= 192, D = 10: <= 20 straddlers per seam, 3.33 inputs apart, each reading <= 64 inputs
            constexpr int PER = 20, UNI = 64 + (PER * 10 + 2) / 3 + 4;
            const int64_t last_m = g.k_begin + g.count - 1;
            const int64_t in_avail = (last_m * g.D + g.I - 1) / g.I - g.in_base + t.nloop;          // inputs the caller guarantees
            const dim3 grid((nseams + 7) / 8 + ((lead > 0 || tail > 0) ? 1 : 0));
            if (lanes == 8)
                hipLaunchKernelGGL((k_resample3_stragglers<PER, UNI, 8>), grid, dim3(256), 0, s, g, sg, d_groups, d_plain_taps, t.ntaps_plain, d_in, d_out,
                                   first, nseams, in_avail);
            else
                hipLaunchKernelGGL((k_resample3_stragglers<PER, UNI, 4>), grid, dim3(256), 0, s, g, sg, d_groups, d_plain_taps, t.ntaps_plain, d_in, d_out,
                                   first, nseams, in_avail);
        }
    }
    return true;
}

}  // namespace sdrhip
