// tools/mfma_mul_bench.hip -- development aid (gfx950): can the matrix pipe serve as an EXACT f32 multiplier array
// beside the VALU?  v_mfma_f32_4x4x1_16b_f32 with C = 0 is D[i][j] = fma(A[i], B[j], +0) = round(A[i] * B[j]): one rounding,
// i.e. the bits of v_mul_f32 (a -0 product comes out as +0, which an accumulation that started at +0 cannot tell apart).
//   part 1  exactness: 2^26 random bit patterns (denormals, zeros, infinities, NaNs included) against v_mul_f32
//   part 2  issue rates: MFMA alone, v_add_f32 alone, MFMA + its 4 adds (scalar or packed), v_add_f32 with a DPP
//           wave_shr:1 operand -- at 1..4 waves per SIMD
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mfma_mul_bench.hip -o tools/mfma_mul_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// classes of operands: 0 = any bit pattern, 1 = normal mid-range, 2 = products near the denormal boundary, 3 = small integers
// (u8 - 128) against taps
__device__ __forceinline__ float operand(uint32_t h, int cls, bool is_a)
{
    if (cls == 0) return __uint_as_float(h);
    if (cls == 1) return __uint_as_float((h & 0x807fffffu) | ((100u + (h >> 23) % 56u) << 23));
    if (cls == 2) return __uint_as_float((h & 0x807fffffu) | ((is_a ? 20u : 90u) + ((h >> 23) % 24u)) << 23);
    if (is_a) return __uint_as_float((h & 0x807fffffu) | ((96u + (h >> 23) % 30u) << 23));     // a tap / 128
    return (float)((int)(h & 255u) - 128);
}

__global__ void __launch_bounds__(256, 2) k_exact(uint32_t seed, int cls, unsigned long long* bad, unsigned long long* negzero, uint32_t* first_bad)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const float a = operand(mix(gid * 2u + seed), cls, true), b = operand(mix(gid * 2u + 1u + seed * 77u), cls, false);
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
    const int lane = threadIdx.x & 63;
    unsigned long long nb = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float ai = __shfl(a, (lane & ~3) + i, 64);
        const float want = ai * b;
        const uint32_t w = __float_as_uint(want), g = __float_as_uint(d[i]);
        if (w == g) continue;
        if (w == 0x80000000u && g == 0u) { nz++; continue; }        // -0 product: fma(a, b, +0) = +0
        if (want != want && d[i] != d[i]) continue;                   // both NaN (payload / sign of a NaN is not arithmetic)
        if (nb == 0 && atomicAdd(bad, 0ull) == 0ull) { first_bad[0] = __float_as_uint(ai); first_bad[1] = __float_as_uint(b); first_bad[2] = w; first_bad[3] = g; }
        nb++;
    }
    if (nb) atomicAdd(bad, nb);
    if (nz) atomicAdd(negzero, nz);
}

// MODE 0: MFMA only   1: v_add only (4 per step)   2: MFMA + 4 adds of its results   3: v_add with a DPP wave_shr:1 operand
// 4: MFMA + 4 adds, the adds one step behind the MFMA (software pipelined)   5: v_mul + v_add (today's arithmetic), 4 pairs per step
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_rate(float* out, const float* in, int iters)
{
    float s[8], t[8];
    for (int u = 0; u < 8; u++) { s[u] = in[threadIdx.x + 64 * u]; t[u] = in[threadIdx.x + 64 * u + 512]; }
    float acc[32];
    for (int i = 0; i < 32; i++) acc[i] = (float)i;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 prev = z;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile("" : "+v"(s[u]), "+v"(t[u]));
            if constexpr (MODE == 0) {
                const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(t[u], s[u], z, 0, 0, 0);
                asm volatile("" ::"v"(p));
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[4 * u + i] = acc[4 * u + i] + s[(u + i) & 7];
            } else if constexpr (MODE == 2) {
                const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(t[u], s[u], z, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[4 * u + i] = acc[4 * u + i] + p[i];
            } else if constexpr (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[4 * u + i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[4 * u + i]), 0x138, 0xf, 0xf, false)) + s[(u + i) & 7];
            } else if constexpr (MODE == 4) {
                const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(t[u], s[u], z, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[4 * u + i] = acc[4 * u + i] + prev[i];
                prev = p;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[4 * u + i] = acc[4 * u + i] + s[(u + i) & 7] * t[u];
            }
        }
    }
    float r = prev[0];
    for (int i = 0; i < 32; i++) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void rate(const char* name, int wg_per_cu, float* dout, const float* din, int ncu)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = ncu * wg_per_cu;
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, dout, din, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // one "step" = 1 MFMA (256 products) and/or 4 wave-wide adds; waves per SIMD = wg_per_cu
    const double steps_per_simd = (double)iters * 8 * wg_per_cu;
    const double ns_per_step = ms * 1e6 / steps_per_simd;
    printf("  %-44s %d waves/SIMD  %8.3f ms  %6.2f ns/step/SIMD = %5.2f cycles at 2.4 GHz\n", name, wg_per_cu, ms, ns_per_step, ns_per_step * 2.4);
}


typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
template <int NR> struct RV;
template <> struct RV<4> { typedef f4 type; };
template <> struct RV<16> { typedef f16v type; };
template <> struct RV<32> { typedef f32v type; };

// SHAPE 4 / 16 / 32 = v_mfma_f32_4x4x1_16b / 16x16x1_4b / 32x32x1_2b (256 / 1024 / 2048 products, 8 / 32 / 64 cycles);
// NADD plain v_add_f32 per MFMA, placed right BEHIND the MFMA in program order and consuming the results of the MFMA issued
// two MFMAs earlier (three rotating result sets; inline asm pins the order, which the scheduler otherwise undoes).
// clk[0] += shader cycles (s_memtime), clk[1] += 100 MHz ticks (s_memrealtime) of lane 0 of every wave
template <int SHAPE, int NADD, bool WITH_MFMA>
__global__ void __launch_bounds__(256, 2) k_rate2(float* out, const float* in, int iters, unsigned long long* clk)
{
    constexpr int NR = SHAPE == 4 ? 4 : SHAPE;
    typedef typename RV<NR>::type rv;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float acc[32];
    for (int i = 0; i < 32; i++) acc[i] = (float)i;
    rv P[3];
    for (int u = 0; u < 3; u++)
        for (int i = 0; i < NR; i++) P[u][i] = in[threadIdx.x + 512 + i + u];
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if constexpr (WITH_MFMA) {
                if constexpr (SHAPE == 4) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(P[u]) : "v"(a), "v"(b));
                else if constexpr (SHAPE == 16) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, 0" : "=v"(P[u]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, 0" : "=v"(P[u]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int i = 0; i < NADD; i++) {
                float src = P[(u + 1) % 3][i % NR];
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[(NADD * u + i) & 31]) : "v"(src));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&clk[0], t1 - t0); atomicAdd(&clk[1], r1 - r0); }
    float r = 0.f;
    for (int u = 0; u < 3; u++)
        for (int i = 0; i < NR; i++) r += P[u][i];
    for (int i = 0; i < 32; i++) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE, int NADD, bool WITH_MFMA>
static void rate2(int wg_per_cu, float* dout, const float* din, int ncu, unsigned long long* dclk)
{
    const int iters = 8192 / SHAPE * 4 / 3 * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = ncu * wg_per_cu;
    hipLaunchKernelGGL((k_rate2<SHAPE, NADD, WITH_MFMA>), dim3(blocks), dim3(256), 0, 0, dout, din, iters / 4, dclk);
    CK(hipDeviceSynchronize());
    CK(hipMemset(dclk, 0, 16));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 8; rep++) hipLaunchKernelGGL((k_rate2<SHAPE, NADD, WITH_MFMA>), dim3(blocks), dim3(256), 0, 0, dout, din, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
    const double steps_per_simd = 8.0 * iters * 3 * wg_per_cu;
    const double ns_per_step = ms * 1e6 / steps_per_simd;
    printf("  %s%-9s + %2d v_add  %d waves/SIMD  %7.3f ms  clock %.2f GHz  %6.2f ns = %6.1f shader cycles per step   (%5.2f cycles per 256 products)\n",
           WITH_MFMA ? "MFMA " : "no MFMA, ", WITH_MFMA ? (SHAPE == 4 ? "4x4x1" : SHAPE == 16 ? "16x16x1" : "32x32x1") : "", NADD, wg_per_cu, ms, ghz, ns_per_step,
           ns_per_step * ghz, ns_per_step * ghz / (SHAPE == 4 ? 1 : SHAPE == 16 ? 4 : 8));
}


// ---- energy: the same number of multiply-adds per second at the power cap?  Sustained rows (seconds each) with the socket power
// and sclk of THIS device sampled from hwmon; operands are random floats that rotate through 8 registers (the energy of a
// multiplier depends on how its inputs toggle).
// KIND 0: VALU only: 16 x (v_pk_mul_f32 by an SGPR-held tap + v_pk_add_f32)            = 2048 MACs per step and wave
// KIND 1: 2 x MFMA 16x16x1_4b + 32 v_add_f32 of results issued two MFMAs earlier       = 2048 MACs
// KIND 2: 1 x MFMA 32x32x1_2b + 32 v_add_f32                                           = 2048 MACs
// KIND 3: 8 x MFMA 4x4x1_16b + 32 v_add_f32                                            = 2048 MACs
// KIND 4: VALU only, scalar: 32 x (v_mul_f32 + v_add_f32)                              = 2048 MACs
typedef float f2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(256, 2) k_energy(float* out, const float* in, int iters, unsigned long long* clk)
{
    float a[8], b[8];
    for (int u = 0; u < 8; u++) { a[u] = in[threadIdx.x + 256 * u]; b[u] = in[threadIdx.x + 256 * u + 2048]; }
    f2v x[8];
    for (int u = 0; u < 8; u++) x[u] = f2v{a[u], b[u]};
    const float t0s = in[4096], t1s = in[4097], t2s = in[4098], t3s = in[4099];      // wave-uniform "taps" -> SGPRs
    float acc[32];
    f2v acc2[16];
    for (int i = 0; i < 32; i++) acc[i] = 0.f;
    for (int i = 0; i < 16; i++) acc2[i] = f2v{0.f, 0.f};
    f32v P32[3];
    f16v P16[3];
    f4 P4[3];
    for (int u = 0; u < 3; u++) {
        for (int i = 0; i < 32; i++) P32[u][i] = 0.f;
        for (int i = 0; i < 16; i++) P16[u][i] = 0.f;
        for (int i = 0; i < 4; i++) P4[u][i] = 0.f;
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    f2v p;
                    const f2v tp = (i & 2) ? f2v{t2s, t3s} : f2v{t0s, t1s};
                    if (i & 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(x[(i + u) & 7]), "s"(tp));
                    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(x[(i + u) & 7]), "s"(tp));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(p));
                }
            } else if constexpr (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    float p;
                    const float tp = (i & 3) == 0 ? t0s : (i & 3) == 1 ? t1s : (i & 3) == 2 ? t2s : t3s;
                    asm volatile("v_mul_f32 %0, %2, %1" : "=v"(p) : "v"(i & 1 ? a[(i / 2 + u) & 7] : b[(i / 2 + u) & 7]), "s"(tp));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(p));
                }
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int w = (2 * u + h) % 3;
                    asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, 0" : "=v"(P16[w]) : "v"(a[(2 * u + h) & 7]), "v"(b[(u + 3 * h) & 7]));
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        float src = P16[(w + 1) % 3][i];
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[16 * h + i]) : "v"(src));
                    }
                }
            } else if constexpr (KIND == 2) {
                asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, 0" : "=v"(P32[u]) : "v"(a[(2 * u) & 7]), "v"(b[(u + 3) & 7]));
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    float src = P32[(u + 1) % 3][i];
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(src));
                }
            } else {
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const int w = (8 * u + h) % 3;
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(P4[w]) : "v"(a[(u + h) & 7]), "v"(b[(u + 3 * h) & 7]));
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float src = P4[(w + 1) % 3][i];
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[4 * h + i]) : "v"(src));
                    }
                }
            }
        }
        // keep the sums bounded: halve them now and then (exact, and the same few instructions for every kind)
        if ((it & 15) == 15) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] *= 0.03125f;
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[i] *= 0.03125f;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&clk[0], c1 - c0); atomicAdd(&clk[1], r1 - r0); }
    float r = 0.f;
    for (int u = 0; u < 3; u++) {
        for (int i = 0; i < 32; i++) r += P32[u][i];
        for (int i = 0; i < 16; i++) r += P16[u][i];
        for (int i = 0; i < 4; i++) r += P4[u][i];
    }
    for (int i = 0; i < 32; i++) r += acc[i];
    for (int i = 0; i < 16; i++) r += acc2[i].x + acc2[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

#include "power_sampler.hpp"
template <int KIND>
static void energy_row(const char* name, float* dout, const float* din, int ncu, unsigned long long* dclk, PowerSampler& ps, double seconds)
{
    const int blocks = ncu * 4, iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_energy<KIND>), dim3(blocks), dim3(256), 0, 0, dout, din, iters, dclk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_energy<KIND>), dim3(blocks), dim3(256), 0, 0, dout, din, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms1;
    CK(hipEventElapsedTime(&ms1, e0, e1));
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    CK(hipMemset(dclk, 0, 16));
    ps.start();
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_energy<KIND>), dim3(blocks), dim3(256), 0, 0, dout, din, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    const PowerStats st = ps.finish(0.3);
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
    const double macs = (double)reps * blocks * 4 * iters * 3 * 2048.0 * 1.0;      // per wave and step: 2048 MACs (64 lanes x 32)
    printf("  %-46s %6.2f T MAC/s  %4.0f W (%4.0f..%4.0f)  sclk %4.0f MHz (probe %4.0f)  %5.1f pJ per MAC above idle 250 W\n", name, macs / (ms * 1e-3) / 1e12, st.mean_w,
           st.min_w, st.max_w, st.mean_sclk_mhz, clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0, (st.mean_w - 250.0) / (macs / (ms * 1e-3)) * 1e12);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    const bool only_energy = getenv("ONLY_ENERGY") != nullptr;
    unsigned long long *dbad, *dnz;
    uint32_t* dfirst;
    CK(hipMalloc(&dbad, 8));
    CK(hipMalloc(&dnz, 8));
    CK(hipMalloc(&dfirst, 16));
    const char* cname[4] = {"any bit pattern", "normal mid-range", "denormal-range products", "taps/128 x (u8-128)"};
    for (int cls = 0; cls < 4 && !only_energy; cls++) {
        CK(hipMemset(dbad, 0, 8));
        CK(hipMemset(dnz, 0, 8));
        for (uint32_t seed = 1; seed <= 4; seed++) hipLaunchKernelGGL(k_exact, dim3(1 << 16), dim3(256), 0, 0, seed * 0x9e3779b9u, cls, dbad, dnz, dfirst);
        CK(hipDeviceSynchronize());
        unsigned long long bad, nz;
        uint32_t fb[4];
        CK(hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&nz, dnz, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(fb, dfirst, 16, hipMemcpyDeviceToHost));
        printf("exactness, %-24s: %llu of %llu products differ from v_mul_f32 (%llu are -0 -> +0, not counted)", cname[cls], bad, 4ull * (1ull << 26), nz);
        if (bad) printf("  first: a=%08x b=%08x want=%08x got=%08x", fb[0], fb[1], fb[2], fb[3]);
        printf("\n");
    }
    float *dout, *din;
    CK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4));
    CK(hipMalloc(&din, 2048 * 4));
    std::vector<float> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = 0.5f + (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    CK(hipMemcpy(din, h.data(), 2048 * 4, hipMemcpyHostToDevice));
    for (int w : {1, 2, 3, 4}) {
        if (only_energy) break;
        printf("-- %d workgroup(s) of 256 per CU\n", w);
        rate<0>("MFMA 4x4x1 alone", w, dout, din, ncu);
        rate<1>("4 v_add_f32 alone", w, dout, din, ncu);
        rate<3>("4 v_add_f32 dpp wave_shr:1 alone", w, dout, din, ncu);
        rate<5>("4 v_mul_f32 + 4 v_add_f32 (today)", w, dout, din, ncu);
        rate<2>("MFMA + 4 adds of its own results", w, dout, din, ncu);
        rate<4>("MFMA + 4 adds of the previous results", w, dout, din, ncu);
    }
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, 16));
    for (int w : {1, 2, 4}) {
        if (only_energy) break;
        printf("-- %d workgroup(s) of 256 per CU: MFMA shape x plain adds of the previous results\n", w);
        rate2<4, 4, false>(w, dout, din, ncu, dclk);
        rate2<4, 0, true>(w, dout, din, ncu, dclk);
        rate2<4, 1, true>(w, dout, din, ncu, dclk);
        rate2<4, 2, true>(w, dout, din, ncu, dclk);
        rate2<4, 4, true>(w, dout, din, ncu, dclk);
        rate2<16, 16, false>(w, dout, din, ncu, dclk);
        rate2<16, 0, true>(w, dout, din, ncu, dclk);
        rate2<16, 8, true>(w, dout, din, ncu, dclk);
        rate2<16, 12, true>(w, dout, din, ncu, dclk);
        rate2<16, 16, true>(w, dout, din, ncu, dclk);
        rate2<32, 32, false>(w, dout, din, ncu, dclk);
        rate2<32, 0, true>(w, dout, din, ncu, dclk);
        rate2<32, 16, true>(w, dout, din, ncu, dclk);
        rate2<32, 24, true>(w, dout, din, ncu, dclk);
        rate2<32, 32, true>(w, dout, din, ncu, dclk);
    }
    {
        char bdf[64] = {0};
        CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
        PowerSampler ps(bdf);
        const double secs = getenv("ENERGY_SECONDS") ? atof(getenv("ENERGY_SECONDS")) : 3.0;
        printf("-- energy rows: 4 waves per SIMD, %.1f s each, random operands; hwmon of %s %s\n", secs, bdf, ps.available() ? "found" : "NOT found");
        float* din2;
        CK(hipMalloc(&din2, 4100 * 4));
        std::vector<float> h2(4100);
        uint32_t sd = 12345;
        for (auto& v : h2) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) * (1.0f / 8388608.0f) - 1.0f); }
        CK(hipMemcpy(din2, h2.data(), 4100 * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            energy_row<0>("VALU: 16 x (v_pk_mul SGPR tap + v_pk_add)", dout, din2, ncu, dclk, ps, secs);
            energy_row<4>("VALU: 32 x (v_mul SGPR tap + v_add)", dout, din2, ncu, dclk, ps, secs);
            energy_row<1>("2 x MFMA 16x16x1 + 32 v_add", dout, din2, ncu, dclk, ps, secs);
            energy_row<2>("1 x MFMA 32x32x1 + 32 v_add", dout, din2, ncu, dclk, ps, secs);
            energy_row<3>("8 x MFMA 4x4x1 + 32 v_add", dout, din2, ncu, dclk, ps, secs);
        }
    }
    return 0;
}
