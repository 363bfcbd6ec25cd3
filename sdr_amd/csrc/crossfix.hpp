// crossfix.hpp -- generic "Cross" (seam) fix-up kernels shared by the tiled kernels (internal header).
//
// The tiled kernels compute every output in the SIMD ("One") order; the few outputs whose window straddles a
// seam of the reference's input buffers are then rewritten in the sequential order the reference's Haskell
// fallbacks use (FilterInternal.hs:397-423).  These versions take any D / Lp / I and read global memory
// directly; the hot configurations have LDS-staged specialisations next to their kernels.
#pragma once
#include "kernels.hpp"

namespace sdrhip {
namespace {

// Cross outputs of a complex filter / decimator: sequential over the Lp plain taps
// (filterCrossHighLevel with Mult (Complex a) a, FilterInternal.hs:397-408, Util.hs:87-88).
template <bool U8 = false>
__global__ void __launch_bounds__(256) k_fir_cplx_crossfix(Geom g, const float* __restrict__ xtaps,
                                                            const void* __restrict__ in, float* __restrict__ out,
                                                            int64_t first_seam, int nseams, int per_seam)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseams * per_seam) return;
    const int si = t / per_seam, ci = t - si * per_seam;
    const int64_t edge = (first_seam + si) * g.seamBI;
    const int64_t m = (edge + g.D - 1) / g.D - 1 - ci;
    if (m < g.k_begin || m >= g.k_begin + g.count) return;
    const int64_t v = m * g.D;
    if (!(v < edge && v + g.Lp > edge)) return;
    float re = 0.0f, im = 0.0f;
    if constexpr (U8) {   // interleaved u8 IQ: convert.c's (u - 128) / 128 on the way in (exact)
        const uchar2* x = reinterpret_cast<const uchar2*>(in) + (v - g.in_base);
        for (int j = 0; j < g.Lp; j++) {
            const uchar2 u = x[j];
            re = re + (((float)u.x - 128.0f) * (1.0f / 128.0f)) * xtaps[j];
            im = im + (((float)u.y - 128.0f) * (1.0f / 128.0f)) * xtaps[j];
        }
    } else {
        const float2* x = reinterpret_cast<const float2*>(in) + (v - g.in_base);
        for (int j = 0; j < g.Lp; j++) {
            const float2 s = x[j];
            re = re + s.x * xtaps[j];
            im = im + s.y * xtaps[j];
        }
    }
    *reinterpret_cast<float2*>(out + 2 * (m - g.k_begin)) = make_float2(re, im);
}

// Cross outputs of a real FIR / decimator: sequential over the Lp plain taps
// (filterCrossHighLevel / decimateCrossHighLevel, FilterInternal.hs:397-408).
__global__ void __launch_bounds__(256) k_fir_real_crossfix(Geom g, const float* __restrict__ xtaps,
                                                            const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t first_seam, int nseams, int per_seam, float gain,
                                                            int apply_gain)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseams * per_seam) return;
    const int si = t / per_seam, ci = t - si * per_seam;
    const int64_t edge = (first_seam + si) * g.seamBI;
    const int64_t m = (edge + g.D - 1) / g.D - 1 - ci;
    if (m < g.k_begin || m >= g.k_begin + g.count) return;
    const int64_t v = m * g.D;
    if (!(v < edge && v + g.Lp > edge)) return;
    const float* x = in + (v - g.in_base);
    float r = 0.0f;
    for (int j = 0; j < g.Lp; j++) r = r + x[j] * xtaps[j];
    if (apply_gain) r = r * gain;
    out[m - g.k_begin] = r;
}

// Cross outputs of a resampler, real or complex data (resampleCrossHighLevel, FilterInternal.hs:410-423):
// taps = stride I (drop filterOffset coeffs) over the UNPADDED taps, sequential.
template <bool CPLX>
__global__ void __launch_bounds__(256) k_resample_crossfix(Geom g, const float* __restrict__ plain, int ntaps,
                                                            const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t first_seam, int nseams, int per_seam)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseams * per_seam) return;
    const int si = t / per_seam, ci = t - si * per_seam;
    const int64_t edge = (first_seam + si) * g.seamBI;           // upsampled units
    const int64_t m = (edge + g.D - 1) / g.D - 1 - ci;
    if (m < g.k_begin || m >= g.k_begin + g.count) return;
    const int64_t v = m * g.D;
    if (!(v < edge && v + g.Lp > edge)) return;
    if (!seam_has_crossover(edge, g.I, g.D, g.Lp)) return;       // the Pipe goes straight to the next buffer here
    if (late_output_is_one(m, edge, g.I, g.D, g.outB)) return;   // first output of an output block, first input beyond the seam
    const int64_t pos = (v + g.I - 1) / g.I;                     // inOff(m)
    const int fo = (int)(pos * g.I - v);
    if constexpr (CPLX) {
        const float2* x = reinterpret_cast<const float2*>(in) + (pos - g.in_base);
        float re = 0.0f, im = 0.0f;
        for (int l = 0, j = fo; j < ntaps; l++, j += g.I) {
            const float2 sv = x[l];
            re = re + sv.x * plain[j];
            im = im + sv.y * plain[j];
        }
        *reinterpret_cast<float2*>(out + 2 * (m - g.k_begin)) = make_float2(re, im);
    } else {
        const float* x = in + (pos - g.in_base);
        float r = 0.0f;
        for (int l = 0, j = fo; j < ntaps; l++, j += g.I) r = r + x[l] * plain[j];
        out[m - g.k_begin] = r;
    }
}

// seams (multiples of seamBI) strictly inside the launch's window range
inline void seam_range(const Geom& g, int64_t& first, int64_t& last)
{
    const int64_t v_lo = g.k_begin * g.D, v_hi = (g.k_begin + g.count - 1) * g.D + g.Lp;
    first = v_lo / g.seamBI + 1;
    last = (v_hi - 1) / g.seamBI;
}

// Cross outputs of the real resampler (resampleCrossHighLevel, FilterInternal.hs:410-423):
// taps = stride I (drop filterOffset coeffs) over the UNPADDED taps, sequential -- and with I = 1 the Cross outputs of a real
// decimator / filter (decimateCrossHighLevel / filterCrossHighLevel, FilterInternal.hs:397-408).  A group of LPG
// (32 or 64) lanes serves one seam: its <= PER straddlers read a union of <= UNI consecutive inputs, staged in LDS.
// the work of workgroup `wg` (256 threads): seams [wg * SPW, wg * SPW + SPW).  A kernel of its own below; round 6: also the body of
// the seam workgroups of kernels_chain.hip's k_resample3_stragglers (seams + lead-in + tail of a 3/10 launch in ONE launch)
template <int PER, int UNI, int LPG = 32>
__device__ __forceinline__ void resample_real_crossfix_wg(int wg, const Geom& g, const float* __restrict__ plain, int ntaps,
                                                          const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t first_seam, int nseams, int64_t in_avail, float gain, int apply_gain)
{
    static_assert((LPG == 32 || LPG == 64) && PER <= LPG, "one group of LPG lanes per seam");
    constexpr int SPW = 256 / LPG;                                       // seams per workgroup
    __shared__ float lds[SPW][UNI];
    __shared__ float tl[256];
    const int tid = threadIdx.x, sl = tid / LPG, ci = tid % LPG;
    const int si = wg * SPW + sl;
    const bool live = si < nseams;
    int64_t edge = 0, m_lo = 0, p_lo = 0;
    if (live) {
        edge = (first_seam + si) * g.seamBI;                     // in upsampled units
        const int64_t m_hi = (edge + g.D - 1) / g.D - 1;         // last output starting before the edge
        m_lo = m_hi - (PER - 1);
        if (m_lo < 0) m_lo = 0;
        p_lo = (m_lo * g.D + g.I - 1) / g.I;                     // inOff(m_lo): first input of the union
        // branch-free (clamped index + select): the five loads of a lane are in flight together
        constexpr int NE = (UNI + LPG - 1) / LPG;
        float ve[NE];
#pragma unroll
        for (int k = 0; k < NE; k++) {
            const int64_t idx = p_lo + (ci + LPG * k) - g.in_base;
            const bool ok = idx >= 0 && idx < in_avail;
            const float x = in[ok ? idx : 0];
            ve[k] = ok ? x : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NE; k++)
            if (ci + LPG * k < UNI) lds[sl][ci + LPG * k] = ve[k];
    }
    // the unpadded taps, once per workgroup (the sequential loop below would otherwise wait for a global load per tap)
    const bool taps_in_lds = ntaps <= 256;
    if (taps_in_lds) tl[tid] = tid < ntaps ? plain[tid] : 0.0f;
    __syncthreads();
    if (!live || ci >= PER) return;
    const int64_t m = m_lo + ci;
    if (m < g.k_begin || m >= g.k_begin + g.count) return;
    const int64_t v = m * g.D;
    if (!(v < edge && v + g.Lp > edge)) return;
    if (!seam_has_crossover(edge, g.I, g.D, g.Lp)) return;       // the Pipe goes straight to the next buffer here
    if (late_output_is_one(m, edge, g.I, g.D, g.outB)) return;   // first output of an output block, first input beyond the seam
    const int64_t pos = (v + g.I - 1) / g.I;                     // inOff(m)
    const int fo = (int)(pos * g.I - v);
    const float* x = lds[sl] + (pos - p_lo);
    float r = 0.0f;
    if (taps_in_lds && g.I == 1) {
        // decimators / filters: unit stride, so the reads of eight taps ahead are issued together (the chain of dependent
        // additions is the sequential order itself; what can be hidden is the LDS latency in front of each of them)
        int j = 0;
        for (; j + 8 <= ntaps; j += 8) {
            float xv[8], tv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { xv[i] = x[j + i]; tv[i] = tl[j + i]; }
#pragma unroll
            for (int i = 0; i < 8; i++) r = r + xv[i] * tv[i];
        }
        for (; j < ntaps; j++) r = r + x[j] * tl[j];
    } else if (taps_in_lds) {
        for (int l = 0, j = fo; j < ntaps; l++, j += g.I) r = r + x[l] * tl[j];
    } else {
        for (int l = 0, j = fo; j < ntaps; l++, j += g.I) r = r + x[l] * plain[j];
    }
    if (apply_gain) r = r * gain;
    out[m - g.k_begin] = r;
}

template <int PER, int UNI, int LPG = 32>
__global__ void __launch_bounds__(256) k_resample_real_crossfix(Geom g, const float* __restrict__ plain, int ntaps,
                                                                 const float* __restrict__ in, float* __restrict__ out,
                                                                 int64_t first_seam, int nseams, int64_t in_avail, float gain = 1.0f,
                                                                 int apply_gain = 0)
{
    resample_real_crossfix_wg<PER, UNI, LPG>((int)blockIdx.x, g, plain, ntaps, in, out, first_seam, nseams, in_avail, gain, apply_gain);
}

}  // namespace
}  // namespace sdrhip
