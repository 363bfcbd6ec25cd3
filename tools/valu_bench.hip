// VALU issue-rate microbenchmark for gfx950 (development aid): unfused f32
// multiply+add, packed vs scalar, VGPR vs SGPR tap operand, vs FMA.  Inline asm so
// the compiler cannot hoist or re-associate anything.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* taps, int iters)
{
    f2 hs = f2{taps[0], taps[1]};          // uniform -> SGPR pair
    f2 hv = f2{taps[0] + threadIdx.x * 1e-9f, taps[1]};  // VGPR pair
    f2 acc[16], t[16], x[4];
    for (int i = 0; i < 16; i++) acc[i] = f2{(float)threadIdx.x * 1e-3f + i, 1.0f - i * 1e-2f};
    for (int i = 0; i < 4; i++) x[i] = f2{(float)threadIdx.x * 1e-3f + i, (float)threadIdx.x * 2e-3f + i};
    for (int it = 0; it < (MODE == 6 ? 0 : iters); it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if constexpr (MODE == 0)
                    asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %1, %0" : "=&v"(t[i]), "+v"(acc[i]) : "v"(hv), "v"(x[i & 3]));
                else if constexpr (MODE == 1)
                    asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %0" : "=&v"(t[i]), "+v"(acc[i]) : "s"(hs), "v"(x[i & 3]));
                else if constexpr (MODE == 2)
                    asm volatile("v_mul_f32 %0, %4, %6\n\tv_mul_f32 %1, %5, %7\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1"
                                 : "=&v"(t[i].x), "=&v"(t[i].y), "+v"(acc[i].x), "+v"(acc[i].y)
                                 : "s"(hs.x), "s"(hs.y), "v"(x[i & 3].x), "v"(x[i & 3].y));
                else if constexpr (MODE == 3)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(hv), "v"(x[i & 3]));
                else if constexpr (MODE == 4)
                    asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %5, %1" : "+v"(acc[i].x), "+v"(acc[i].y)
                                 : "s"(hs.x), "s"(hs.y), "v"(x[i & 3].x), "v"(x[i & 3].y));
                else if constexpr (MODE == 5)  // scalar, VOP2 forms with VGPR tap
                    asm volatile("v_mul_f32 %0, %4, %6\n\tv_mul_f32 %1, %5, %7\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1"
                                 : "=&v"(t[i].x), "=&v"(t[i].y), "+v"(acc[i].x), "+v"(acc[i].y)
                                 : "v"(hv.x), "v"(hv.y), "v"(x[i & 3].x), "v"(x[i & 3].y));
            }
        }
    }
    if constexpr (MODE == 6) {
        // software-pipelined: 16 independent multiplies, then the 16 adds that consume them
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int rep = 0; rep < 4; rep++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t[i]) : "s"(hs), "v"(x[i & 3]));
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(t[i]));
            }
        }
    }
    f2 s = f2{0.f, 0.f};
    for (int i = 0; i < 16; i++) s = s + acc[i] + t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
void run(const char* name, int blocks, int inst_per_cmac)
{
    float *out, *taps;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&taps, 16);
    float h[2] = {0.999f, 1.001f};
    (void)hipMemcpy(taps, h, 8, hipMemcpyHostToDevice);
    int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3;
    double cmacs = (double)blocks * 256 * iters * 64;          // complex (2-float) multiply-add pairs
    double waves = (double)blocks * 4, simds = 1024.0;
    double inst_per_wave = (double)iters * 64 * inst_per_cmac;
    double cyc = t * 2.4e9 * simds / (waves * inst_per_wave);   // if waves were spread evenly
    printf("%-34s blocks=%5d %8.3f ms  %6.2f T f32 mul-add pairs/s  %5.2f cyc@2.4GHz/wave-instr/SIMD\n", name, blocks, ms,
           2 * cmacs / t / 1e12, cyc);
    (void)hipFree(out);
    (void)hipFree(taps);
}

int main()
{
    for (int blocks : {256, 512, 1024, 2048, 8192}) {
        run<6>("pk_mul x16 then pk_add x16 (SGPR)", blocks, 2);
        run<0>("pk_mul+pk_add (VGPR tap)", blocks, 2);
        run<1>("pk_mul+pk_add (SGPR tap, op_sel)", blocks, 2);
        run<2>("v_mul+v_add x2 (SGPR tap)", blocks, 4);
        run<5>("v_mul+v_add x2 (VGPR tap)", blocks, 4);
        run<3>("pk_fma", blocks, 1);
        run<4>("v_fma x2", blocks, 2);
    }
    return 0;
}
