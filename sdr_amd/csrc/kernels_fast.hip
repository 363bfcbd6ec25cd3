// kernels_fast.hip -- LDS-tiled kernels for the hot configurations.
//
// K2: complex FIR decimator, real taps, the AVX "RC" summation order
//     (decimateAVXRC, c_sources/decimate.c:105-113 -> avx_dotprod_R common.h:58-72 on
//     duplicated taps -> avx_hadd_C common.h:82-90):
//        out[o] = (L0 + L1) + (L2 + L3),   L_k = sum_{j = k, k+4, ..} h[j] * x[o*D + j]
//     each L_k accumulated from +0 in increasing j, separate mul and add.
//
// Design (CDNA4, not a translation of the AVX loop):
//   * one workgroup = NT threads = a tile of NT*R consecutive outputs; the tile's
//     input span ((NT*R-1)*D + P samples) is staged ONCE in LDS as float2, loaded
//     from HBM with 16-byte coalesced loads (u8 IQ input: 8 samples per load,
//     converted on the way in -- convert.c's (u-128)/128 is exact, so fusing it
//     cannot change a bit);
//   * each thread produces R consecutive outputs, so one LDS read of a sample
//     feeds up to R MACs (R = 4: 38 sample reads per output instead of 128) --
//     register-level reuse keeps the LDS pipe at ~30 % while the VALU does the work;
//   * per-thread LDS stride is padded (2 float2 per D*R samples) so ds_read_b128
//     is bank-conflict free;
//   * the 4 complex lane partials of every output live in registers: 4*R float2
//     accumulators = 8*R independent dependency chains, no cross-lane traffic;
//   * taps are wave-uniform -> scalar loads / SGPR operands.
//   "Cross" outputs (seam straddlers, sequential order) are rewritten afterwards by
//   a tiny fix-up kernel on the same stream: ~1.5 % of outputs.
#include "decimate_tile.hpp"

namespace sdrhip {

bool launch_decimate_c4_fast(hipStream_t s, const Geom& g, const float* d_plain_taps, int P, const float* d_cross_taps,
                             const void* d_in, bool in_is_u8, float* d_out, bool last_tap_zero)
{
    if (g.I != 1 || g.count <= 0 || g.seamBI < 0) return false;
    // Decimation 8: exact kernels for 128 and 52 taps; every other length up to 128 (multiples of 4: mkDecimatorC pads
    // to that) runs on the 128-tap kernel with run-time guards that skip the tap blocks the shorter filter does not have.
    // Decimation 4 and 16: the guarded 128-tap kernel as well.
    // Decimation 8 and 16 with 129..256 taps: guarded instantiations of a 256-tap kernel.
    const int pmax = g.D == 4 ? 128 : 256;
    if (!((g.D == 8 || g.D == 4 || g.D == 16) && P >= 8 && P <= pmax && P % 4 == 0 && g.Lp == P && P > g.D)) return false;
    const bool guarded = !(g.D == 8 && (P == 128 || P == 52));
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    int64_t x0 = g.k_begin * g.D - g.in_base;
    // vector loads need 16-byte aligned tile starts (tiles begin at multiples of 8 samples from x0)
    uintptr_t base = reinterpret_cast<uintptr_t>(d_in);
    if (in_is_u8) {
        if (((base + 2 * (uintptr_t)x0) & 15) != 0) return false;
    } else {
        if (((base + 8 * (uintptr_t)x0) & 15) != 0) return false;
    }
    if ((reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return false;
    // Launch-bound sizes (a host block per push .. a 2^20-sample shard) compute their Cross outputs inside the tile kernel:
    // one launch instead of two (abi_device.cpp: the 5v row).  At full size that loses to the fix-up launch (0.87-1.0 ms
    // against 0.71 per 2^29 samples: the straddlers' sequential loops hold whole workgroups' resources).
    const int64_t inl_max = 5 * (int64_t)small_launch_outputs();
    const bool inl = g.seamBI > 0 && g.count <= inl_max;
    bool inlined = false;
    // R = 2 outputs per thread, 256 threads, 4 workgroups per CU.  Alternatives measured on MI355X and dropped:
    // R = 4 (the compiler's SGPR allocation for the sliding tap window collapses into spills), 512-thread
    // workgroups (same rate), and a persistent kernel that prefetches the next tile into registers during the
    // MAC phase (same rate: with random data the kernel is power/clock-limited, the exposed load phase is ~4 %).
    if (guarded) {
#define GUARDED(DV, PV, TCV) do { if (in_is_u8) launch_c4<DV, PV, 2, 256, true, TCV, true>(s, g, d_plain_taps, d_in, d_out, inl, &inlined); \
                                  else launch_c4<DV, PV, 2, 256, false, TCV, true>(s, g, d_plain_taps, d_in, d_out, inl, &inlined); } while (0)
        if (g.D == 4) GUARDED(4, 128, 4);
        else if (g.D == 16 && P <= 128) { if (P % 8 == 0) GUARDED(16, 128, 8); else GUARDED(16, 128, 4); }
        else if (g.D == 16) { if (P % 8 == 0) GUARDED(16, 256, 8); else GUARDED(16, 256, 4); }
        else if (P <= 128) { if (P % 8 == 0) GUARDED(8, 128, 8); else GUARDED(8, 128, 4); }
        else { if (P % 8 == 0) GUARDED(8, 256, 8); else GUARDED(8, 256, 4); }
#undef GUARDED
    } else if (P == 52) {
        // the tap count of the reference FM example's RF decimation filter (51 -> 52): u8 launches that are not launch-bound take the
        // systolic kernel's 64-tap instantiation (round 6)
        if (in_is_u8 && !inl && launch_decimate_c4_systolic(s, g, d_plain_taps, P, d_in, true, d_out, false, d_cross_taps, &inlined)) {
        } else if (in_is_u8) launch_c4<8, 52, 2, 256, true>(s, g, d_plain_taps, d_in, d_out, inl, &inlined);
        else launch_c4<8, 52, 2, 256, false>(s, g, d_plain_taps, d_in, d_out, inl, &inlined);
    } else {
        // 127 taps padded to 128 (the FM chain's decimator).  Launches that are not launch-bound take the register-resident
        // systolic kernel (kernels_systolic.hip, round 4: no LDS in the MAC loop; same bits); the padding tap's MACs are skipped
        // on u8 input (decimate_tile.hpp: PSKIP)
        if (!inl && launch_decimate_c4_systolic(s, g, d_plain_taps, P, d_in, in_is_u8, d_out, in_is_u8 && last_tap_zero, d_cross_taps, &inlined)) {
        } else if (in_is_u8 && last_tap_zero) launch_c4<8, 128, 2, 256, true, 8, false, 4, 0, 1>(s, g, d_plain_taps, d_in, d_out, inl, &inlined);
        else if (in_is_u8) launch_c4<8, 128, 2, 256, true>(s, g, d_plain_taps, d_in, d_out, inl, &inlined);
        else launch_c4<8, 128, 2, 256, false>(s, g, d_plain_taps, d_in, d_out, inl, &inlined);
    }

    if (g.seamBI != 0 && !inlined) {
        // seams whose straddling outputs may fall in [k_begin, k_end)
        int64_t v_lo = g.k_begin * g.D, v_hi = (g.k_begin + g.count - 1) * g.D + g.Lp;
        int64_t first = v_lo / g.seamBI + 1;          // first boundary strictly above v_lo
        int64_t last = (v_hi - 1) / g.seamBI;         // last boundary strictly below v_hi
        if (last >= first) {
            int nseams = (int)(last - first + 1);
            constexpr int PER = 16, SPW = 16;          // 16 candidate slots per seam (ceil((128-1)/8))
            dim3 grid((nseams + SPW - 1) / SPW), block(PER * SPW);
#define FIX(U, LPV) hipLaunchKernelGGL((k_decimate_c_crossfix<U, 8, LPV, PER, SPW>), grid, block, 0, s, g, d_cross_taps, d_in, d_out, first, nseams)
            if (g.D != 8 || P > 128) {   // the generic one-thread-per-straddler kernel (crossfix.hpp)
                const int per = (g.Lp - 1 + g.D - 1) / g.D;
                const int64_t total = (int64_t)nseams * per;
                const dim3 ggrid((unsigned)((total + 255) / 256));
                if (in_is_u8) hipLaunchKernelGGL(k_fir_cplx_crossfix<true>, ggrid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per);
                else hipLaunchKernelGGL(k_fir_cplx_crossfix<false>, ggrid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per);
            } else if (guarded) {
                if (in_is_u8) hipLaunchKernelGGL((k_decimate_c_crossfix<true, 8, 128, PER, SPW, true>), grid, block, 0, s, g, d_cross_taps, d_in, d_out, first, nseams);
                else hipLaunchKernelGGL((k_decimate_c_crossfix<false, 8, 128, PER, SPW, true>), grid, block, 0, s, g, d_cross_taps, d_in, d_out, first, nseams);
            } else if (P == 52) { if (in_is_u8) FIX(true, 52); else FIX(false, 52); }
            else { if (in_is_u8) FIX(true, 128); else FIX(false, 128); }
#undef FIX
        }
    }
    return true;
}

}  // namespace sdrhip
