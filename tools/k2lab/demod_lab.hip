// tools/k2lab/demod_lab.hip -- VALU-only throughput of fmDemod's per-sample arithmetic (demod.hpp), select form and
// ternary (branchy) form, against waves per SIMD and data (noise-like IQ: every atan range in every wave; a clean FM
// tone: small phase steps only).  No memory traffic in the timed loop: what the arithmetic alone sustains.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Isdr_amd/csrc tools/k2lab/demod_lab.hip -o tools/k2lab/demod_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../sdr_amd/csrc/demod.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace sdrhip;

template <int FORM, int SPT>
__global__ void __launch_bounds__(256) k_demod_only(const float2* __restrict__ in, float* __restrict__ out, int iters)
{
    extern __shared__ float pad[];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float2 s[SPT + 1];
#pragma unroll
    for (int i = 0; i <= SPT; i++) s[i] = in[(size_t)gid * (SPT + 1) + i];
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            asm volatile("" : "+v"(s[i].x), "+v"(s[i].y));       // opaque: no hoisting out of the loop
            const float p = FORM ? fm_phase_sel(s[i + 1], s[i]) : fm_phase_tern(s[i + 1], s[i]);
            acc = acc + p;
        }
    }
    if (iters < 0) pad[threadIdx.x] = acc;
    out[gid] = acc;
}

template <int FORM, int SPT>
void run(const char* name, const float2* d_in, float* d_out, int wgs_per_cu, int iters, const char* data)
{
    // one WG = 4 waves = one per SIMD; dynamic LDS caps the workgroups per CU
    const size_t lds = wgs_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wgs_per_cu) - 1024;
    CK(hipFuncSetAttribute((const void*)k_demod_only<FORM, SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_demod_only<FORM, SPT>), dim3(grid), dim3(256), lds, 0, d_in, d_out, iters);
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_demod_only<FORM, SPT>), dim3(grid), dim3(256), lds, 0, d_in, d_out, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double samples = (double)grid * 256 * SPT * iters;
    printf("%-8s SPT %d %-6s %d wave(s)/SIMD: %7.3f ms, %7.1f G samples/s  (2^26 samples in %.3f ms)\n", name, SPT, data, wgs_per_cu, ms,
           samples / ms / 1e6, 67108864.0 / (samples / ms) );
}

int main()
{
    const int n = 256 * 8 * 256 * 9;
    std::vector<float> noise((size_t)2 * n), tone((size_t)2 * n);
    uint64_t s = 12345;
    for (auto& v : noise) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0) * 0.3f; }
    double ph = 0.0;
    for (int i = 0; i < n; i++) { ph += 0.15 * sin(i * 0.001) + 0.02; tone[2 * i] = (float)(0.4 * cos(ph)); tone[2 * i + 1] = (float)(0.4 * sin(ph)); }
    float2 *d_noise, *d_tone; float* d_out;
    CK(hipMalloc(&d_noise, noise.size() * 4)); CK(hipMalloc(&d_tone, tone.size() * 4)); CK(hipMalloc(&d_out, (size_t)n * 4));
    CK(hipMemcpy(d_noise, noise.data(), noise.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tone, tone.data(), tone.size() * 4, hipMemcpyHostToDevice));
    const int iters = 200;
    for (int w : {1, 2, 4, 8}) {
        run<0, 4>("ternary", d_noise, d_out, w, iters, "noise");
        run<1, 4>("select", d_noise, d_out, w, iters, "noise");
        run<0, 4>("ternary", d_tone, d_out, w, iters, "tone");
        run<1, 4>("select", d_tone, d_out, w, iters, "tone");
    }
    run<1, 1>("select", d_noise, d_out, 8, iters, "noise");
    run<1, 2>("select", d_noise, d_out, 8, iters, "noise");
    run<1, 8>("select", d_noise, d_out, 4, iters, "noise");
    return 0;
}
