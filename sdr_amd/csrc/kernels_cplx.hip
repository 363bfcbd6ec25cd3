// kernels_cplx.hip -- the 3/10 polyphase resampler on COMPLEX data (resampleAVXRC / resampleSSERC, resample.c:106-142 ->
// avx_dotprod_C / sse_dotprod_C, common.h:108-155), specialised like its real twin (kernels_chain.hip: k_resample3_fast):
// three polyphase groups of NLOOP padded taps, input increments {4, 3, 3}.
//
// One thread = one polyphase cycle = 3 consecutive outputs (groups 0, 1, 2), so every tap is wave-uniform (scalar loads,
// SGPR operands).  The three windows of a cycle start 0 / 4 / 7 samples into the same stretch of input, so the thread
// slides ONE 16-sample register window over it, 8 samples (four 16-byte LDS reads) per step of 8 taps: one LDS read feeds
// 3 x 2 complex MACs instead of one (the lane-split kernel reads every operand of every MAC from LDS).
// Summation order ("RC2", plain taps): NP complex partials p_m over taps m, m + NP, ..; AVX: q_k = p_k + p_{k+4},
// (q0 + q1) + (q2 + q3); SSE: q_k = p_k + p_{k+2}, q0 + q1.
#include "crossfix.hpp"
#include "kernels.hpp"

namespace sdrhip {

namespace {

typedef float cplx_f8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ cplx_f8 cplx_taps8(const float* base, int chunk)
{
    typedef const __attribute__((address_space(4))) cplx_f8* cp;
    uint64_t a = reinterpret_cast<uint64_t>(base) + 32u * (uint32_t)chunk;
    asm volatile("" : "+s"(a));
    return *reinterpret_cast<cp>(a);
}

template <int NP>
__device__ __forceinline__ float2 fold_rc2(const float2 (&p)[NP])
{
    if constexpr (NP == 8) {
        float2 q[4];
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = make_float2(p[l].x + p[l + 4].x, p[l].y + p[l + 4].y);
        return make_float2((q[0].x + q[1].x) + (q[2].x + q[3].x), (q[0].y + q[1].y) + (q[2].y + q[3].y));
    } else {
        static_assert(NP == 4, "AVX: 8 partials, SSE: 4");
        const float2 q0 = make_float2(p[0].x + p[2].x, p[0].y + p[2].y), q1 = make_float2(p[1].x + p[3].x, p[1].y + p[3].y);
        return make_float2(q0.x + q1.x, q0.y + q1.y);
    }
}

template <int NLOOP, int NP, int NT>
__global__ void __launch_bounds__(NT, 4) k_resample3c_fast(const float* __restrict__ in, int64_t pos0, int ncycles, int64_t avail_total,
                                                           const float* __restrict__ groups, int row_stride, float* __restrict__ out)
{
    constexpr int PERIOD = 10;
    constexpr int PRE[3] = {0, 4, 7};
    constexpr int WIN = PRE[2] + NLOOP;                   // complex samples one thread reads (71)
    constexpr int SPAN = (NT - 1) * PERIOD + WIN + 9;     // + the last step's reads beyond the last tap (never used)
    constexpr int SPAN2 = (SPAN + 1) / 2;                 // 16-byte vectors
    __shared__ __attribute__((aligned(16))) float2 lds[SPAN2 * 2];
    static_assert(NLOOP % 8 == 0, "taps are walked eight at a time");

    const int cyc0 = blockIdx.x * NT;
    const int64_t base = pos0 + (int64_t)cyc0 * PERIOD;   // first input sample of this workgroup, relative to `in`
    const int64_t av64 = avail_total - (int64_t)cyc0 * PERIOD;
    const int avail = av64 > SPAN ? SPAN : (int)av64;
    const float2* src = reinterpret_cast<const float2*>(in) + base;
    const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (al && av64 >= 2 * SPAN2) {
        // interior tile: every 16-byte load of the workgroup in flight before the first LDS store (a load-store loop pays one
        // HBM round trip per iteration: 5.2 of them here)
        constexpr int NV = (SPAN2 + NT - 1) / NT;
        float4 val[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            int v = threadIdx.x + i * NT;
            v = v < SPAN2 ? v : SPAN2 - 1;
            val[i] = *reinterpret_cast<const float4*>(src + 2 * v);
        }
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = threadIdx.x + i * NT;
            if (v < SPAN2) *reinterpret_cast<float4*>(&lds[2 * v]) = val[i];
        }
    } else {
        for (int v = threadIdx.x; v < SPAN2; v += NT) {
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

    const int cyc = cyc0 + threadIdx.x;
    if (cyc >= ncycles) return;
    // thread window: 10 complex samples = 80 bytes apart: 16-byte aligned, and a 20-dword lane stride is conflict-free for
    // ds_read_b128 (16 lanes x 4 banks tile the 64 banks)
    const float2* win = lds + threadIdx.x * PERIOD;
    float2 acc[3][NP];
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int l = 0; l < NP; l++) acc[g][l] = make_float2(0.0f, 0.0f);
    float2 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float4 t = *reinterpret_cast<const float4*>(win + 2 * i);
        w[2 * i] = make_float2(t.x, t.y);
        w[2 * i + 1] = make_float2(t.z, t.w);
    }
    // taps one step ahead of their use (a scalar-load wait is a full lgkmcnt(0) drain: it must find the loads long issued)
    cplx_f8 tn[3];
#pragma unroll
    for (int g = 0; g < 3; g++) tn[g] = cplx_taps8(groups + g * row_stride, 0);
#pragma unroll
    for (int jc = 0; jc < NLOOP / 8; jc++) {
        // taps 8jc .. 8jc+7 of the three groups meet samples w[PRE[g] + k], k = 0..7 (w[0] = sample 8jc of the window)
        cplx_f8 t8[3];
#pragma unroll
        for (int g = 0; g < 3; g++) t8[g] = tn[g];
        float2 wn[8];                                   // the next 8 samples, read before this step's MACs
        if (jc + 1 < NLOOP / 8) {
#pragma unroll
            for (int g = 0; g < 3; g++) tn[g] = cplx_taps8(groups + g * row_stride, jc + 1);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 t = *reinterpret_cast<const float4*>(win + 8 * (jc + 2) + 2 * i);
                wn[2 * i] = make_float2(t.x, t.y);
                wn[2 * i + 1] = make_float2(t.z, t.w);
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int g = 0; g < 3; g++) {
                const int l = (8 * jc + k) % NP;
                const float h = t8[g][k];
                const float2 x = w[PRE[g] + k];
                acc[g][l].x = acc[g][l].x + h * x.x;
                acc[g][l].y = acc[g][l].y + h * x.y;
            }
        }
        if (jc + 1 < NLOOP / 8) {
            // slide by 8 samples: keep w[8..15], append samples 8(jc+2) .. 8(jc+2)+7
#pragma unroll
            for (int i = 0; i < 8; i++) { w[i] = w[i + 8]; w[8 + i] = wn[i]; }
        }
    }
    float2* o = reinterpret_cast<float2*>(out) + (int64_t)cyc * 3;
#pragma unroll
    for (int g = 0; g < 3; g++) o[g] = fold_rc2<NP>(acc[g]);
}

}  // namespace

bool launch_resample3c_fast(hipStream_t s, const Geom& g, ComplexOrder order, const ResampTable& t, const int* increments, const float* d_groups,
                            const float* d_plain_taps, const float* d_in, float* d_out)
{
    if (t.ngroups != 3 || t.nloop != 64 || g.seamBI < 0 || t.force_seq) return false;
    if (!(order == CO_X4 || order == CO_X2)) return false;
    if (!(increments[0] == 4 && increments[1] == 3 && increments[2] == 3)) return false;
    if (g.seamBI != 0 && d_plain_taps == nullptr) return false;
    if (g.count <= 0) return false;
    constexpr int NT = 256;
    // outputs before the first group-0 output and after the last whole cycle go to the generic kernel
    int lead = (3 - t.group0) % 3;
    if (lead > g.count) lead = g.count;
    const int ncycles = (g.count - lead) / 3;
    const int tail = g.count - lead - 3 * ncycles;
    Geom gs = g;
    gs.seamBI = 0;      // every output as One first; seams are fixed up below
    const int64_t skip = lead > 0 ? t.pre[lead - 1] + increments[(t.group0 + lead - 1) % 3] : 0;
    if (lead > 0) {
        Geom gl = gs;
        gl.count = lead;
        launch_resample_cplx(s, gl, order, t, d_groups, d_plain_taps, d_in, d_out);
    }
    if (ncycles > 0) {
        const int64_t pos = t.pos0 + skip;
        const int64_t avail_total = (int64_t)(ncycles - 1) * 10 + 7 + t.nloop;
        const int blocks = (ncycles + NT - 1) / NT;
        if (order == CO_X4)
            hipLaunchKernelGGL((k_resample3c_fast<64, 8, NT>), dim3(blocks), dim3(NT), 0, s, d_in, pos, ncycles, avail_total, d_groups, t.row_stride,
                               d_out + 2 * lead);
        else
            hipLaunchKernelGGL((k_resample3c_fast<64, 4, NT>), dim3(blocks), dim3(NT), 0, s, d_in, pos, ncycles, avail_total, d_groups, t.row_stride,
                               d_out + 2 * lead);
    }
    if (tail > 0) {
        const int done = lead + 3 * ncycles;
        Geom gt = gs;
        gt.k_begin = g.k_begin + done;
        gt.count = tail;
        ResampTable tt = t;
        tt.group0 = 0;
        tt.pos0 = t.pos0 + skip + (int64_t)ncycles * 10;
        tt.pre[0] = 0; tt.pre[1] = 4; tt.pre[2] = 7;
        launch_resample_cplx(s, gt, order, tt, d_groups, d_plain_taps, d_in, d_out + 2 * done);
    }
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            const int64_t total = (int64_t)nseams * per;
            hipLaunchKernelGGL(k_resample_crossfix<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in,
                               d_out, first, nseams, per);
        }
    }
    return true;
}

}  // namespace sdrhip
