// tools/k2lab/sys_lab.hpp -- shared by sys_lab.hip and sys_lab_prod.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ClkProbe { unsigned long long cyc, rt; unsigned int n; unsigned int pad; };

__device__ __forceinline__ void probe_begin(unsigned long long& c0, unsigned long long& r0)
{
    c0 = __builtin_readcyclecounter();          // s_memtime: shader clock
    r0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
}
__device__ __forceinline__ void probe_end(ClkProbe* pr, unsigned long long c0, unsigned long long r0)
{
    if (pr != nullptr && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&pr->cyc, c1 - c0);
        atomicAdd(&pr->rt, r1 - r0);
        atomicAdd(&pr->n, 1u);
    }
}

// the production tile kernel (FULL body) on 2^k-sample buffers: `pr` != nullptr runs the probed twin
void lab_prod_launch(bool u8, const void* in, int64_t nout, const float* taps, float* out, ClkProbe* pr);
