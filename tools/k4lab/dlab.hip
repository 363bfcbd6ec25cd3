// dlab.hip -- is the fused loader's DEMODULATOR phase faster with more workgroups per CU?  (Round 5 lab: the tile kernel k_resample3_fast<.., DEMOD>
// runs six waves per SIMD at 78-80 VGPRs; its waves wait on memory for 37 % of their cycles.)  Two loaders for the same tile -- 256 polyphase
// cycles = 2621 phases from 2621 complex samples, then a trivial "resampler" (three phases per thread) so that only the loader is timed:
//   A: the production loader: thread t takes samples t, t + 256, ..: two 8-byte loads per sample (itself, its predecessor), 11 rounds;
//   B: pairs: 16-byte loads, the predecessor by DPP from the lane below, lane 0's from a scalar load; five rounds + a ragged round of singles;
// each at 6 and at 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I../../sdr_amd/csrc -I../../include -o dlab dlab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "demod.hpp"

using namespace sdrhip;

constexpr int NT = 256, PERIOD = 10, SPAN = 255 * 10 + 71;     // 2621

__device__ __forceinline__ float dpp_shr1_or(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}

template <int W>
__global__ void __launch_bounds__(NT, W) k_a(const float* __restrict__ in, float* __restrict__ out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) float lds[SPAN + 3];
    __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
    constexpr int NP = (SPAN + NT - 1) / NT;
    const int tile = blockIdx.x;
    const float2* z = reinterpret_cast<const float2*>(in) + 3 + (int64_t)tile * NT * PERIOD;
    atan_table_fill(atbl, threadIdx.x);
    float2 cur[NP], prv[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        int p = threadIdx.x + i * NT;
        p = p < SPAN ? p : SPAN - 1;
        cur[i] = z[p];
        prv[i] = z[p - 1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float y[NP];
    bool rare = false;
#pragma unroll
    for (int i = 0; i < NP - 1; i++) {
        bool q;
        y[i] = fm_phase_common_tbl(cur[i], prv[i], q, atbl);
        rare |= q;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NP - 1; i++) lds[threadIdx.x + i * NT] = y[i];
    if (__any(rare)) {
#pragma unroll
        for (int i = 0; i < NP - 1; i++) lds[threadIdx.x + i * NT] = fm_phase_sel(cur[i], prv[i]);
    }
    if ((int)threadIdx.x + (NP - 1) * NT < SPAN) lds[threadIdx.x + (NP - 1) * NT] = fm_phase_sel(cur[NP - 1], prv[NP - 1]);
    __syncthreads();
    const float* w = lds + threadIdx.x * PERIOD;
    float* o = out + ((int64_t)tile * NT + threadIdx.x) * 3;
    o[0] = w[0] + w[70]; o[1] = w[4] + w[33]; o[2] = w[7] + w[50];
}

template <int W>
__global__ void __launch_bounds__(NT, W) k_b(const float* __restrict__ in, float* __restrict__ out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) float lds[SPAN + 3];
    __shared__ __attribute__((aligned(16))) float atbl[kAtanRows * kAtanRowFloats];
    constexpr int NR = 5;                                       // whole pair rounds: 2560 samples
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave_t0 = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63;
    // sample 0 of the tile sits at an ODD float2 of `in` (in + 1): pairs start one sample earlier (E = 1), as in the library when the
    // run's first input is odd; pair k = samples 2k - 1, 2k of the tile
    const float2* z = reinterpret_cast<const float2*>(in) + 3 + (int64_t)tile * NT * PERIOD;
    const float4* z4 = reinterpret_cast<const float4*>(z - 1);
    atan_table_fill(atbl, threadIdx.x);
    const int lp0 = NR * wave_t0 + lane;                        // wave-contiguous rounds: the lane below holds the predecessor pair
    float4 cur[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) cur[i] = z4[lp0 + 64 * i];
    // the sample before the wave's first pair: a scalar load (nothing else of this kernel is an s_load of HBM data in flight during LDS waits:
    // every load is waited for before the first phase is computed)
    const float2 pl = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(z - 2) + 2 * (2 * NR * (int64_t)wave_t0));
    // the ragged end: samples 2559 .. 2620 (62) as singles on the first lanes of wave 0
    const int pt = 2 * NR * NT - 1 + (int)threadIdx.x;
    float2 tc = make_float2(0.f, 0.f), tp = make_float2(0.f, 0.f);
    if (pt < SPAN) { tc = z[pt]; tp = z[pt - 1]; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float2 y[NR];
    bool rare = false;
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
        float2 o = pl;
        if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].z), 63)),
                                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].w), 63)));
        const float2 pv = make_float2(dpp_shr1_or(o.x, cur[i].z), dpp_shr1_or(o.y, cur[i].w));
        bool q0, q1;
        y[i].x = fm_phase_common_tbl(A, pv, q0, atbl);
        __builtin_amdgcn_sched_barrier(0);
        y[i].y = fm_phase_common_tbl(B, A, q1, atbl);
        rare |= q0 | q1;
        __builtin_amdgcn_sched_barrier(0);
    }
    // phases of samples 2k - 1, 2k -> lds[2k - 1], lds[2k]: the pair's first float sits one position before an even index
    float* lw = lds + 1;                                        // lw[-1 + 2k] .. : keep it simple, two 4-byte stores
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const int k = lp0 + 64 * i;
        if (k > 0) lw[2 * k - 2] = y[i].x;
        lw[2 * k - 1] = y[i].y;
    }
    if (__any(rare)) {
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const float2 A = make_float2(cur[i].x, cur[i].y), B = make_float2(cur[i].z, cur[i].w);
            float2 o = pl;
            if (i > 0) o = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].z), 63)),
                                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[i - 1].w), 63)));
            const float2 pv = make_float2(dpp_shr1_or(o.x, cur[i].z), dpp_shr1_or(o.y, cur[i].w));
            const int k = lp0 + 64 * i;
            if (k > 0) lw[2 * k - 2] = fm_phase_sel(A, pv);
            lw[2 * k - 1] = fm_phase_sel(B, A);
        }
    }
    if (pt < SPAN) lds[pt] = fm_phase_sel(tc, tp);
    __syncthreads();
    const float* w = lds + threadIdx.x * PERIOD;
    float* o = out + ((int64_t)tile * NT + threadIdx.x) * 3;
    o[0] = w[0] + w[70]; o[1] = w[4] + w[33]; o[2] = w[7] + w[50];
}

template <class K>
float run(K kern, const float* in, float* out, int ntiles, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), 0, 0, in, out, ntiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), 0, 0, in, out, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const int64_t n = (int64_t)1 << 26;
    const int ntiles = (int)(n / (NT * PERIOD)) - 1;
    std::vector<float> h(2 * (n + 16));
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)((s >> 9) & 0xffff) - 32768) / 32768.0f; }
    float *in, *oa, *ob;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&oa, (size_t)ntiles * NT * 3 * 4);
    hipMalloc(&ob, (size_t)ntiles * NT * 3 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_a<6>, dim3(ntiles), dim3(NT), 0, 0, in, oa, ntiles);
    hipLaunchKernelGGL(k_b<8>, dim3(ntiles), dim3(NT), 0, 0, in, ob, ntiles);
    hipDeviceSynchronize();
    std::vector<float> a((size_t)ntiles * NT * 3), b(a.size());
    hipMemcpy(a.data(), oa, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), ob, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); i++) bad += memcmp(&a[i], &b[i], 4) != 0;
    printf("loader A against loader B: %zu of %zu outputs differ\n", bad, a.size());
    for (int rnd = 0; rnd < 3; rnd++) {
        printf("round %d: A w=6 %.4f ms   A w=8(spills?) %.4f   B w=6 %.4f   B w=8 %.4f   B w=4 %.4f\n", rnd, run(k_a<6>, in, oa, ntiles, 50), run(k_a<8>, in, oa, ntiles, 50),
               run(k_b<6>, in, ob, ntiles, 50), run(k_b<8>, in, ob, ntiles, 50), run(k_b<4>, in, ob, ntiles, 50));
    }
    return 0;
}
