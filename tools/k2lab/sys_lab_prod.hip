// tools/k2lab/sys_lab_prod.hip -- the production tile kernel for tools/k2lab/sys_lab.hip, in a translation unit of its own so that
// it is compiled exactly as the library compiles it (SLP vectoriser on; the lab's own kernels want it off), plus a probed twin
// (shader cycles / 100 MHz ticks of one wave in 64 workgroups) of the FULL body.
#include <hip/hip_runtime.h>

#include "../../sdr_amd/csrc/kernels_fast.hip"
#include "sys_lab.hpp"

namespace sdrhip {
int small_launch_outputs() { return 32768; }
// kernels_fast.hip's launcher refers to the production systolic kernel; the lab launches the tile kernel directly
bool launch_decimate_c4_systolic(hipStream_t, const Geom&, const float*, int, const void*, bool, float*, bool) { return false; }
}
using namespace sdrhip;

template <bool U8, int PSKIP>
__global__ void __launch_bounds__(256) k_prod_probe(const void* __restrict__ in, int64_t x0, int count, const float* __restrict__ taps, float* __restrict__ out,
                                                    ClkProbe* pr)
{
    using T = Tile<8, 128, 2, 256>;
    const int ntiles = (count + T::OUTS - 1) / T::OUTS;
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    decimate_c4_tile<8, 128, 2, 256, U8, 8, false, 4, 0, PSKIP, true>(tile, in, x0, count, taps, out, 128, 0, 0);
    probe_end(pr, c0, r0);
}

using TP = Tile<8, 128, 2, 256>;
void lab_prod_launch(bool u8, const void* in, int64_t nout, const float* taps, float* out, ClkProbe* pr)
{
    static bool set = false;
    auto k_u8 = k_decimate_c4<8, 128, 2, 256, true, 8, false, 4, 0, 1, true>;
    auto k_cf = k_decimate_c4<8, 128, 2, 256, false, 8, false, 4, 0, 0, true>;
    auto p_u8 = k_prod_probe<true, 1>;
    auto p_cf = k_prod_probe<false, 0>;
    if (!set) {
        for (const void* k : {(const void*)k_u8, (const void*)k_cf, (const void*)p_u8, (const void*)p_cf})
            (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TP::LDS_BYTES);
        set = true;
    }
    const int ntiles = (int)((nout + TP::OUTS - 1) / TP::OUTS);
    const int grid = ((ntiles + 63) / 64) * 64;
    const int nfull = (int)(nout / TP::OUTS);
    if (pr) {
        if (u8) hipLaunchKernelGGL(p_u8, dim3(grid), dim3(256), TP::LDS_BYTES, 0, in, (int64_t)0, (int)nout, taps, out, pr);
        else hipLaunchKernelGGL(p_cf, dim3(grid), dim3(256), TP::LDS_BYTES, 0, in, (int64_t)0, (int)nout, taps, out, pr);
    } else {
        if (u8) hipLaunchKernelGGL(k_u8, dim3(grid), dim3(256), TP::LDS_BYTES, 0, in, (int64_t)0, (int)nout, taps, out, 128, 0, 0, nfull);
        else hipLaunchKernelGGL(k_cf, dim3(grid), dim3(256), TP::LDS_BYTES, 0, in, (int64_t)0, (int)nout, taps, out, 128, 0, 0, nfull);
    }
}
