// kernels_fast_filter.hip -- complex FILTERS (decimation 1) on the LDS-tiled decimator of decimate_tile.hpp
// (filterAVXRC, c_sources/filter.c:106-114 -> avx_dotprod_R on duplicated taps -> avx_hadd_C: the AVX "RC" order,
// 4 complex lane partials, (L0 + L1) + (L2 + L3)).
//
// With D = 1 a thread's R = 4 consecutive outputs share a window of P + 3 samples: one 16-byte LDS read (two samples) feeds
// up to 8 complex MACs -- 16 packed instructions per LDS read where the FM chain's decimator has 7.5 -- with the taps in SGPRs
// one chunk ahead and the whole walk unrolled (the rolled kernel this replaces, k_filter_cplx4_fast, pays a scalar tap load and
// its full drain per four taps).  Measured, 2^24 samples, 128 taps: 234 -> 160 us without seams (26.9 T real MAC/s, the FM
// decimator's rate), 272 -> 170 with the 8192-sample seams; eight outputs per thread (180 VGPRs, two waves per SIMD): 170 / 205.  Exact tap counts only (128 and 64: what the guarded walk of the
// decimator does for shorter filters needs the decimation to be a multiple of the tap chunk); everything else stays on
// k_filter_cplx4_fast.  Cross outputs by the generic fix-up kernel, as there.
#include "decimate_tile.hpp"

namespace sdrhip {

namespace {

// Cross outputs of a complex FILTER (filterCrossHighLevel, FilterInternal.hs:404-408: sequential order over the plain taps):
// the LP - 1 straddlers of a seam have windows one sample apart; one workgroup stages their union (2 LP - 2 complex samples,
// coalesced) and the taps in LDS and thread c walks window c -- the complex twin of k_filter_real_crossfix_lds (the generic
// one-thread-per-straddler kernel reads every sample of every window from global memory: 37 us per 2^24 samples against 6).
template <int LP>
__global__ void __launch_bounds__(LP) k_filter_cplx_crossfix_lds(Geom g, const float* __restrict__ xtaps, const float* __restrict__ in,
                                                                  float* __restrict__ out, int64_t first_seam)
{
    constexpr int UNI = 2 * LP - 2;
    __shared__ float2 lds[2 * LP];
    __shared__ float tl[LP];
    const int tid = threadIdx.x;
    const int64_t edge = (first_seam + blockIdx.x) * g.seamBI;
    const int64_t v0 = edge - (LP - 1);                               // first straddler starts here
    const int64_t lo = g.k_begin - g.in_base, hi = g.k_begin + g.count - 1 + LP - g.in_base;
    const float2* in2 = reinterpret_cast<const float2*>(in);
    float2 v[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int e = tid + k * LP;
        const int64_t idx = v0 + e - g.in_base;
        v[k] = (e < UNI && idx >= lo && idx < hi) ? in2[idx] : make_float2(0.0f, 0.0f);
    }
    tl[tid] = xtaps[tid];
#pragma unroll
    for (int k = 0; k < 2; k++) lds[tid + k * LP] = v[k];
    __syncthreads();
    const int64_t m = v0 + tid;                                        // candidates v0 .. v0 + LP - 2
    if (tid >= LP - 1 || m < g.k_begin || m >= g.k_begin + g.count) return;
    float re = 0.0f, im = 0.0f;
#pragma unroll 16
    for (int j = 0; j < LP; j++) {
        const float2 x = lds[tid + j];
        const float h = tl[j];
        re = re + x.x * h;
        im = im + x.y * h;
    }
    *reinterpret_cast<float2*>(out + 2 * (m - g.k_begin)) = make_float2(re, im);
}

}  // namespace

bool launch_filter_c4_tile(hipStream_t s, const Geom& g, const float* d_plain_taps, int P, const float* d_cross_taps, const float* d_in,
                           float* d_out)
{
    if (g.I != 1 || g.D != 1 || g.count <= 0 || g.seamBI < 0 || g.Lp != P) return false;
    if (!(P == 128 || P == 64)) return false;
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    if (g.count < 16384) return false;                       // small launches: the rolled kernel's tile is a quarter of this one's
    const int64_t x0 = g.k_begin - g.in_base;
    if (((reinterpret_cast<uintptr_t>(d_in) + 8 * (uintptr_t)x0) & 15) != 0) return false;      // 16-byte aligned tile starts
    if ((reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return false;
    if (P == 128) launch_c4<1, 128, 4, 256, false>(s, g, d_plain_taps, d_in, d_out);
    else launch_c4<1, 64, 4, 256, false>(s, g, d_plain_taps, d_in, d_out);
    if (g.seamBI != 0) {
        // Cross outputs: sequential order over the plain taps (filterCrossHighLevel, FilterInternal.hs:404-408)
        const int64_t v_lo = g.k_begin, v_hi = g.k_begin + g.count - 1 + g.Lp;
        const int64_t first = v_lo / g.seamBI + 1, last = (v_hi - 1) / g.seamBI;
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            if (P == 128) hipLaunchKernelGGL((k_filter_cplx_crossfix_lds<128>), dim3(nseams), dim3(128), 0, s, g, d_cross_taps, d_in, d_out, first);
            else hipLaunchKernelGGL((k_filter_cplx_crossfix_lds<64>), dim3(nseams), dim3(64), 0, s, g, d_cross_taps, d_in, d_out, first);
        }
    }
    return true;
}

}  // namespace sdrhip
