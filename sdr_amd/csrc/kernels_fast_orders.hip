// kernels_fast_orders.hip -- the LDS-tiled complex decimator of kernels_fast.hip (decimate_tile.hpp) for the OTHER summation
// orders of the reference's complex kernels:
//     SSE "RC"   decimateSSERC  (decimate.c:84-93):   2 complex partials, l0 + l1
//     AVX "RC2"  decimateAVXRC2 (common.h:129-155):   8 partials folded p_k + p_{k+4}, then (q0+q1)+(q2+q3)
//     SSE "RC2"  decimateSSERC2 (common.h:108-127):   4 partials folded p_k + p_{k+2}, then q0 + q1
// Same tile, same register-level reuse (two outputs per thread, two samples per 16-byte LDS read, taps in SGPRs); only the
// number of partial accumulators and the final fold differ, so these families run at the speed of the FM chain's first
// stage instead of the lane-split kernel's one-LDS-read-per-MAC (complex SSE order, decimation 8, 128 taps: 119 -> see
// profiles/README.md).  Guarded instantiations only (any tap count up to the template's, multiples of 4); decimation 4, 8, 16.
#include "decimate_tile.hpp"

namespace sdrhip {

namespace {

template <int NP, int ORD, bool U8OK>
bool launch_order(hipStream_t s, const Geom& g, const float* taps, int P, const void* in, bool in_is_u8, float* out)
{
    if (in_is_u8 && !U8OK) return false;
    constexpr bool TC4OK = NP <= 4;
    const bool tc8 = P % 8 == 0;
    if (!tc8 && !TC4OK) return false;
#define GO(DV, PV, TCV, U8V) launch_c4<DV, PV, 2, 256, U8V, TCV, true, NP, ORD>(s, g, taps, in, out)
#define BY_TC(DV, PV, U8V) do { if (tc8) GO(DV, PV, 8, U8V); else if constexpr (TC4OK) GO(DV, PV, 4, U8V); } while (0)
#define BY_U8(DV, PV) do { if (in_is_u8) { if constexpr (U8OK) BY_TC(DV, PV, true); } else BY_TC(DV, PV, false); } while (0)
    if (P <= 128) {
        if (g.D == 4) {
            // decimation 4 walks its window in blocks of 4 taps (D % TC == 0): not for the 8-partial order
            if constexpr (TC4OK) { if (in_is_u8) { if constexpr (U8OK) GO(4, 128, 4, true); } else GO(4, 128, 4, false); }
            else return false;
        } else if (g.D == 8) BY_U8(8, 128);
        else BY_U8(16, 128);
        return true;
    }
    // 129 .. 256 taps: cfloat input, the SSE "RC" order only (what SDR.Filter can construct)
    if constexpr (NP == 2) {
        if (in_is_u8 || g.D == 4) return false;
        if (g.D == 8) BY_TC(8, 256, false);
        else BY_TC(16, 256, false);
        return true;
    }
    return false;
#undef BY_U8
#undef BY_TC
#undef GO
}

}  // namespace

bool launch_decimate_c_orders_fast(hipStream_t s, const Geom& g, ComplexOrder order, const float* d_plain_taps, int P, const float* d_cross_taps,
                                   const void* d_in, bool in_is_u8, float* d_out)
{
    if (g.I != 1 || g.count <= 0 || g.seamBI < 0) return false;
    if (!((g.D == 8 || g.D == 4 || g.D == 16) && P >= 8 && P <= 256 && P % 4 == 0 && g.Lp == P && P > g.D)) return false;
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    const int64_t x0 = g.k_begin * g.D - g.in_base;
    const uintptr_t base = reinterpret_cast<uintptr_t>(d_in);
    if (((base + (in_is_u8 ? 2 : 8) * (uintptr_t)x0) & 15) != 0) return false;      // 16-byte aligned tile starts
    if ((reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return false;
    bool took = false;
    switch (order) {
        case CO_L2: took = launch_order<2, 0, true>(s, g, d_plain_taps, P, d_in, in_is_u8, d_out); break;
        case CO_X4: took = P % 8 == 0 && launch_order<8, 1, false>(s, g, d_plain_taps, P, d_in, in_is_u8, d_out); break;
        case CO_X2: took = launch_order<4, 1, false>(s, g, d_plain_taps, P, d_in, in_is_u8, d_out); break;
        default: return false;
    }
    if (!took) return false;
    if (g.seamBI != 0) {
        // Cross outputs: sequential order over the plain taps, the same for every SIMD order (FilterInternal.hs:397-402)
        const int64_t v_lo = g.k_begin * g.D, v_hi = (g.k_begin + g.count - 1) * g.D + g.Lp;
        const int64_t first = v_lo / g.seamBI + 1, last = (v_hi - 1) / g.seamBI;
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            const int64_t total = (int64_t)nseams * per;
            const dim3 grid((unsigned)((total + 255) / 256));
            if (in_is_u8) hipLaunchKernelGGL(k_fir_cplx_crossfix<true>, grid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per);
            else hipLaunchKernelGGL(k_fir_cplx_crossfix<false>, grid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per);
        }
    }
    return true;
}

}  // namespace sdrhip
