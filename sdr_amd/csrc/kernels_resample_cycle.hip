// kernels_resample_cycle.hip -- real polyphase resamplers I/D with an odd decimation (2/3, 5/7, 3/5, ...), AVX / SSE lane order
// (resampleAVXRR / resampleSSERR, c_sources/resample.c:52-87 -> avx_dotprod_R / sse_dotprod_R, common.h:34-72),
// any filter length: the design of the FM chain's 3/10 kernel (kernels_chain.hip:k_resample3_fast) without its compile-time
// tap count.
//
// A launch starts at an output of polyphase group 0, so thread t owns one whole CYCLE: the I outputs I t .. I t + I - 1, one
// per group, which start pre[g] inputs into the thread's window (pre = prefix sums of the groups' increments, compile-time
// from I and D) -- every tap is wave-uniform and comes straight from the group table by scalar loads (s_load_dwordx8), one
// chunk of groups ahead of its use.  The window is walked eight taps per step in a rolled loop over pairs of steps, with a
// sliding register window (PM + 24 values: what the I groups need in two steps plus the next eight in flight from LDS) --
// so the filter length is a run-time number (a multiple of the lane count), and a MAC costs its multiply and its add where
// the lane-split kernel of kernels_split.hip pays one ds_read_b32 on top.  An odd decimation makes the lanes' window reads
// (stride D dwords) bank-conflict free; even ones stay on the split kernel (3/10 has its own).  Tiles are staged with
// aligned 16-byte loads from the 16-byte boundary below the tile's first input; the thread windows start `shift` floats
// into the LDS copy.  Lead / tail outputs (before the first group-0 output, after the last whole cycle) go to the generic
// kernel, Cross outputs to the generic fix-up kernel of crossfix.hpp.
//
// Measured (tools/resamp_cycle_ab.py, 2^24 inputs, 8192-sample seams; lane-split kernel -> this one, G inputs/s):
// 2/3 191 taps AVX 132 -> 235, 5/7 216 -> 406, 3/5 235 -> 344, 4/5 150 taps 219 -> 375, 6/7 700 taps 79 -> 148,
// 1/3 154 -> 164; SSE order 2/3 165 -> 250, 5/7 181 -> 433.
#include <atomic>
#include <type_traits>

#include "crossfix.hpp"
#include "kernels.hpp"

namespace sdrhip {

namespace {

// the polyphase walk of prepareCoeffs (FilterInternal.hs:297-319) at compile time: group g (in walk order from filter
// offset 0) advances the input by inc[g]; pre[g] = inputs between the cycle's first window and group g's
template <int I, int D>
struct Walk {
    int inc[I], pre[I], premax;
    constexpr Walk() : inc{}, pre{}, premax(0)
    {
        int off = 0, acc = 0;
        for (int g = 0; g < I; g++) {
            pre[g] = acc;
            inc[g] = (D - off - 1) / I + 1;
            acc += inc[g];
            off = I - 1 - (D - off - 1) % I;
        }
        premax = pre[I - 1];
    }
};

constexpr int CY_NT = 256;
constexpr int CY_NV = 8;                       // 16-byte vectors a thread stages at most

// CPLX: complex data (resampleAVXRC / resampleSSERC, resample.c:106-142 -> avx_dotprod_C / sse_dotprod_C, common.h:108-155, the
// "RC2" order: L complex partials over taps m, m + L, ..; q_k = p_k + p_{k+L/2}, then (q0 + q1) + (q2 + q3) or q0 + q1).  A complex
// stream is a float stream of twice the length whose elements come in pairs: the staging is the real kernel's, positions x 2.
template <int I, int D, int L, bool CPLX>
__global__ void __launch_bounds__(CY_NT) k_resample_cycle(const float* __restrict__ in, int64_t pos0, int ncycles, int64_t avail_total,
                                                          const float* __restrict__ groups, int row_stride, int nloop,
                                                          float* __restrict__ out)
{
    static_assert(D % 2 == 1 && I < D, "odd decimation (conflict-free window reads), decimation > interpolation");
    static_assert(L == 8 || L == 4, "AVX or SSE lane count");
    constexpr Walk<I, D> W{};
    constexpr int PM = W.premax;
    constexpr int ES = CPLX ? 2 : 1;                                   // floats per element
    extern __shared__ __attribute__((aligned(16))) float cy_lds[];
    const int tid = threadIdx.x;
    const int cyc0 = blockIdx.x * CY_NT;
    const int span = ES * ((CY_NT - 1) * D + PM + nloop + 24);        // floats; + 24: the sliding window runs up to two steps past the last tap
    const int64_t first = ES * (pos0 + (int64_t)cyc0 * D);            // first input float of the tile, relative to `in`
    // aligned staging: the LDS copy starts at the 16-byte boundary at or below the tile's first input
    const int shift = (int)((reinterpret_cast<uintptr_t>(in + first) & 15) >> 2);
    const float* src = in + first - shift;
    const int64_t avail = ES * (avail_total - (int64_t)cyc0 * D) + shift;    // floats that exist from src on (the first `shift` of them
                                                                      // may lie in front of the caller's data: they are in the
                                                                      // same 16-byte line as in[first], never used)
    const int span4 = (span + shift + 3) / 4;
    float* tile = cy_lds;
    {
        float4 val[CY_NV];
#pragma unroll
        for (int i = 0; i < CY_NV; i++) {
            const int v = tid + i * CY_NT;
            float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (v < span4) {
                const int64_t s = 4 * (int64_t)v;
                if (s + 3 < avail) {                                   // an aligned 16-byte line holding one valid float is readable
                    q = *reinterpret_cast<const float4*>(src + s);
                } else {
                    if (s + 0 >= shift && s + 0 < avail) q.x = src[s + 0];
                    if (s + 1 >= shift && s + 1 < avail) q.y = src[s + 1];
                    if (s + 2 >= shift && s + 2 < avail) q.z = src[s + 2];
                    if (s + 3 >= shift && s + 3 < avail) q.w = src[s + 3];
                }
            }
            val[i] = q;
        }
#pragma unroll
        for (int i = 0; i < CY_NV; i++) {
            const int v = tid + i * CY_NT;
            if (v < span4) *reinterpret_cast<float4*>(&tile[4 * v]) = val[i];
        }
    }
    __syncthreads();
    const int cyc = cyc0 + tid;
    if (cyc >= ncycles) return;
    const float* wp = tile + shift + tid * (D * ES);
    float acc[I][L][ES];
#pragma unroll
    for (int g = 0; g < I; g++)
#pragma unroll
        for (int l = 0; l < L; l++)
#pragma unroll
            for (int e = 0; e < ES; e++) acc[g][l][e] = 0.0f;
    // taps: wave-uniform loads straight from the group table (s_load_dwordx8), a chunk of GC groups one chunk ahead of its use;
    // window: PM + 24 registers = what two steps need, the second step's last eight and the next pair's in flight
    constexpr int GC = I <= 4 ? I : (I + 1) / 2, NCH = (I + GC - 1) / GC;
    auto load_taps = [&](float (&c)[GC][8], int ch, int j) {
#pragma unroll
        for (int q = 0; q < GC; q++) {
            const int g = ch * GC + q;
            if (g < I) {
                const float* row = groups + g * row_stride + j;
#pragma unroll
                for (int i = 0; i < 8; i++) c[q][i] = row[i];
            }
        }
    };
    float w[ES * (PM + 24)];
#pragma unroll
    for (int k = 0; k < ES * (PM + 8); k++) w[k] = wp[k];
    float c[GC][8];
    load_taps(c, 0, 0);
    // one step = eight taps of every group; BASE = where the step's window starts in w[]
    auto step = [&](auto base_tag, int j0) {
        constexpr int BASE = decltype(base_tag)::value;
#pragma unroll
        for (int i = 0; i < 8 * ES; i++) w[ES * (BASE + PM + 8) + i] = wp[ES * (j0 + PM + 8) + i];        // next step's values, in flight during this step
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            float cn[GC][8];
            if (ch + 1 < NCH) load_taps(cn, ch + 1, j0);
            else load_taps(cn, 0, j0 + 16 <= nloop ? j0 + 8 : j0);        // (the last prefetch re-reads the step's own taps)
#pragma unroll
            for (int q = 0; q < GC; q++) {
                const int g = ch * GC + q;
                if (g < I) {
#pragma unroll
                    for (int i = 0; i < 8; i++)
#pragma unroll
                        for (int e = 0; e < ES; e++)
                            acc[g][i % L][e] = acc[g][i % L][e] + c[q][i] * w[ES * (BASE + W.pre[g] + i) + e];   // tap j0 + i: lane i % L
                }
            }
#pragma unroll
            for (int q = 0; q < GC; q++)
#pragma unroll
                for (int i = 0; i < 8; i++) c[q][i] = cn[q][i];
        }
    };
    int j0 = 0;
#pragma unroll 1
    for (; j0 + 16 <= nloop; j0 += 16) {
        step(std::integral_constant<int, 0>{}, j0);
        step(std::integral_constant<int, 8>{}, j0 + 8);
#pragma unroll
        for (int k = 0; k < ES * (PM + 8); k++) w[k] = w[k + 16 * ES];
    }
    // SSE order pads the groups to a multiple of four taps only: a last half step
    auto half_step = [&](auto base_tag, int j) {
        constexpr int BASE = decltype(base_tag)::value;
#pragma unroll
        for (int g = 0; g < I; g++) {
            const float* row = groups + g * row_stride + j;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int e = 0; e < ES; e++) acc[g][i % L][e] = acc[g][i % L][e] + row[i] * w[ES * (BASE + W.pre[g] + i) + e];
        }
    };
    if (j0 + 8 <= nloop) {
        step(std::integral_constant<int, 0>{}, j0);
        if constexpr (L == 4) if (j0 + 12 <= nloop) half_step(std::integral_constant<int, 8>{}, j0 + 8);
    } else if constexpr (L == 4) {
        if (j0 + 4 <= nloop) half_step(std::integral_constant<int, 0>{}, j0);
    }
    float* o = out + (int64_t)cyc * (I * ES);
#pragma unroll
    for (int g = 0; g < I; g++)
#pragma unroll
        for (int e = 0; e < ES; e++) {
            float r;
            if constexpr (!CPLX) {
                if constexpr (L == 8) r = ((acc[g][0][e] + acc[g][1][e]) + (acc[g][2][e] + acc[g][3][e])) + ((acc[g][4][e] + acc[g][5][e]) + (acc[g][6][e] + acc[g][7][e]));
                else r = (acc[g][0][e] + acc[g][1][e]) + (acc[g][2][e] + acc[g][3][e]);
            } else {
                if constexpr (L == 8)
                    r = ((acc[g][0][e] + acc[g][4][e]) + (acc[g][1][e] + acc[g][5][e])) + ((acc[g][2][e] + acc[g][6][e]) + (acc[g][3][e] + acc[g][7][e]));
                else r = (acc[g][0][e] + acc[g][2][e]) + (acc[g][1][e] + acc[g][3][e]);
            }
            o[g * ES + e] = r;
        }
}

std::atomic<long long> g_cycle_launches{0};

template <int I, int D, int L, bool CPLX>
bool launch_cycle(hipStream_t s, const float* d_in, int64_t pos, int ncycles, int nloop, const float* d_groups, int row_stride, float* d_out)
{
    constexpr Walk<I, D> W{};
    const int span = (CPLX ? 2 : 1) * ((CY_NT - 1) * D + W.premax + nloop + 24);
    const int span4_max = (span + 3 + 3) / 4;
    if (span4_max > CY_NV * CY_NT) return false;
    const size_t lds_bytes = ((size_t)4 * span4_max + 16) * sizeof(float);
    if (lds_bytes > 60 * 1024) return false;
    auto kern = k_resample_cycle<I, D, L, CPLX>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int64_t avail_total = (int64_t)(ncycles - 1) * D + W.premax + nloop;
    hipLaunchKernelGGL(kern, dim3((ncycles + CY_NT - 1) / CY_NT), dim3(CY_NT), lds_bytes, s, d_in, pos, ncycles, avail_total, d_groups, row_stride, nloop,
                       d_out);
    g_cycle_launches++;
    return true;
}

}  // namespace

long long resample_cycle_launch_count() { return g_cycle_launches.load(); }

bool launch_resample_cycle_fast(hipStream_t s, const Geom& g, int lanes, const ResampTable& t, const int* increments, const float* d_groups,
                                const float* d_plain_taps, const float* d_in, float* d_out, bool cplx, ComplexOrder corder)
{
    if (cplx) {
        if (!(corder == CO_X4 || corder == CO_X2)) return false;          // the "RC2" orders of resampleAVXRC / resampleSSERC
        lanes = corder == CO_X4 ? 8 : 4;
    }
    const int es = cplx ? 2 : 1;
    if (t.force_seq || t.ext != nullptr || g.seamBI < 0 || g.count < 4096) return false;
    if (!(lanes == 8 || lanes == 4) || t.ngroups != g.I || t.nloop < 8 || t.nloop % lanes != 0 || t.nloop > 1024) return false;
    if (g.seamBI != 0 && d_plain_taps == nullptr) return false;
    const int I = g.I, D = g.D;
    // the increments must be the walk's (they are, for a table made by prepare_coeffs; a caller-supplied table is checked)
    {
        int off_ = 0;
        for (int q = 0; q < I; q++) {
            if (increments[q] != (D - off_ - 1) / I + 1) return false;
            off_ = I - 1 - (D - off_ - 1) % I;
        }
    }
    // outputs before the first group-0 output and after the last whole cycle go to the generic kernel
    int lead = (I - t.group0) % I;
    if (lead > g.count) lead = g.count;
    const int ncycles = (g.count - lead) / I;
    const int tail = g.count - lead - I * ncycles;
    if (ncycles < 1) return false;
    const int64_t skip = lead > 0 ? t.pre[lead - 1] + increments[(t.group0 + lead - 1) % I] : 0;
    const int64_t pos = t.pos0 + skip;
    bool took = false;
#define CYC(IV, DV) if (I == IV && D == DV) took = lanes == 8 ? launch_cycle<IV, DV, 8, false>(s, d_in, pos, ncycles, t.nloop, d_groups, t.row_stride, d_out + lead) \
                                                              : launch_cycle<IV, DV, 4, false>(s, d_in, pos, ncycles, t.nloop, d_groups, t.row_stride, d_out + lead)
#define CYCC(IV, DV) if (I == IV && D == DV) took = lanes == 8 ? launch_cycle<IV, DV, 8, true>(s, d_in, pos, ncycles, t.nloop, d_groups, t.row_stride, d_out + 2 * lead) \
                                                               : launch_cycle<IV, DV, 4, true>(s, d_in, pos, ncycles, t.nloop, d_groups, t.row_stride, d_out + 2 * lead)
    if (!cplx) { CYC(1, 3); CYC(2, 3); CYC(1, 5); CYC(2, 5); CYC(3, 5); CYC(4, 5); CYC(2, 7); CYC(3, 7); CYC(4, 7); CYC(5, 7); CYC(6, 7); }
    else { CYCC(1, 3); CYCC(2, 3); CYCC(1, 5); CYCC(2, 5); CYCC(3, 5); CYCC(4, 5); CYCC(2, 7); CYCC(3, 7); CYCC(4, 7); CYCC(5, 7); }   // (6/7: 250+ VGPRs)
#undef CYC
#undef CYCC
    if (!took) return false;
    Geom gs = g;
    gs.seamBI = 0;          // every output as One first; seams are fixed up below
    if (lead > 0) {
        Geom gl = gs;
        gl.count = lead;
        if (cplx) launch_resample_cplx(s, gl, corder, t, d_groups, d_plain_taps, d_in, d_out);
        else launch_resample_real(s, gl, lanes, t, d_groups, d_plain_taps, d_in, d_out);
    }
    if (tail > 0) {
        const int done = lead + I * ncycles;
        Geom gt = gs;
        gt.k_begin = g.k_begin + done;
        gt.count = tail;
        ResampTable tt = t;
        tt.group0 = 0;
        tt.pos0 = pos + (int64_t)ncycles * D;
        int acc = 0;
        for (int q = 0; q < I; q++) { tt.pre[q] = acc; acc += increments[q]; }
        if (cplx) launch_resample_cplx(s, gt, corder, tt, d_groups, d_plain_taps, d_in, d_out + es * done);
        else launch_resample_real(s, gt, lanes, tt, d_groups, d_plain_taps, d_in, d_out + done);
    }
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            // LDS-staged fix-up (one group of 32 / 64 lanes per seam: the straddlers' union of inputs and the taps in LDS)
            // where the seam's straddlers fit, the generic one (global reads) beyond
            const int64_t last_m = g.k_begin + g.count - 1;
            const int64_t in_avail = (last_m * g.D + g.I - 1) / g.I - g.in_base + t.nloop;          // inputs the caller guarantees
            auto uni = [&](int PER) { return t.nloop + (PER * g.D + g.I - 1) / g.I + 4; };
            if (cplx) {
                const int64_t total = (int64_t)nseams * per;
                hipLaunchKernelGGL(k_resample_crossfix<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in,
                                   d_out, first, nseams, per);
            } else if (per <= 32 && uni(32) <= 192)
                hipLaunchKernelGGL((k_resample_real_crossfix<32, 192, 32>), dim3((nseams + 7) / 8), dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in,
                                   d_out, first, nseams, in_avail);
            else if (per <= 64 && uni(64) <= 384)
                hipLaunchKernelGGL((k_resample_real_crossfix<64, 384, 64>), dim3((nseams + 3) / 4), dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in,
                                   d_out, first, nseams, in_avail);
            else {
                const int64_t total = (int64_t)nseams * per;
                hipLaunchKernelGGL(k_resample_crossfix<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in,
                                   d_out, first, nseams, per);
            }
        }
    }
    return true;
}

}  // namespace sdrhip
