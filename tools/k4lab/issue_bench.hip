// issue_bench.hip -- how many f32 VALU instructions per second does ONE SIMD issue as a function of the waves resident on it,
// for scalar (v_mul_f32 / v_add_f32, a 2-cycle pass on gfx950) and packed (v_pk_mul_f32 / v_pk_add_f32) operations, as one
// dependent chain per wave or as eight independent chains?  (Round 5: the streaming fmDemod + resampler runs three waves per SIMD
// and its arithmetic alone takes twice what an instruction count x 2 cycles predicts.)
//   hipcc --offload-arch=gfx950 -O3 -o issue_bench issue_bench.hip && ./issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float c = 1.0000001f, d = 1e-7f;
    const f2 c2 = f2{c, c}, d2 = f2{d, d};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if constexpr (MODE == 0) {          // scalar, dependent: 16 instructions on one chain
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[0]) : "v"(c), "v"(d));
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[0]) : "v"(c), "v"(d));
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[0]) : "v"(c), "v"(d));
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[0]) : "v"(c), "v"(d));
            } else if constexpr (MODE == 1) {   // scalar, eight independent chains: 16 instructions
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
            } else if constexpr (MODE == 2) {   // packed, dependent: 16 instructions
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
            } else if constexpr (MODE == 3) {   // packed, eight independent chains
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(d2));
            } else if constexpr (MODE == 4) {   // scalar FMA, independent
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            } else if constexpr (MODE == 5) {   // v_cndmask + v_cmp mix, independent (the demodulator's selects)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(c), "v"(d) : "vcc");
            } else if constexpr (MODE == 7) {   // packed fma, eight independent chains: 16 instructions
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
            } else if constexpr (MODE == 8) {   // packed fma, one dependent chain
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
            } else if constexpr (MODE == 9) {   // alternating packed mul / scalar add, independent: 16 instructions (8 packed + 8 scalar)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(p[i]), "+v"(a[i]) : "v"(c2), "v"(d));
            } else if constexpr (MODE == 10) {  // v_pk_mul with op_sel / neg modifiers (the complex product's forms)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(c2));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]" : "+v"(p[i]) : "v"(d2));
            } else if constexpr (MODE == 11) {  // scalar dependent chain with a literal constant operand (the polynomial's shape)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, 0x3e124925, %0" : "+v"(a[0]) : "v"(c));
            } else if constexpr (MODE == 12) {  // packed dependent chain with a register-pair constant
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[0]) : "v"(c2), "v"(d2));
            } else if constexpr (MODE == 6) {   // v_rcp_f32, independent: 16 instructions
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int cus)
{
    float* out;
    hipMalloc(&out, 256 * cus * 16 * sizeof(float));
    const int iters = 20000;
    printf("%-44s", name);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * w), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * w), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // instructions per SIMD: w waves x iters x 8 x 16
        const double inst = (double)w * iters * 8 * 16;
        printf("  w=%d %6.3f Ginst/s/SIMD", w, inst / (ms * 1e-3) / 1e9);
    }
    printf("\n");
    hipFree(out);
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d; wave64 instructions per second per SIMD (at 2.4 GHz a 2-cycle pass = 1.2 G/s, a 4-cycle pass 0.6 G/s)\n", cus);
    run<0>("scalar mul/add, one dependent chain", cus);
    run<1>("scalar mul/add, 8 independent chains", cus);
    run<2>("packed mul/add, one dependent chain", cus);
    run<3>("packed mul/add, 8 independent chains", cus);
    run<4>("scalar fma, 8 independent chains", cus);
    run<5>("v_cmp + v_cndmask pairs, independent", cus);
    run<6>("v_rcp_f32, independent", cus);
    run<7>("packed fma, 8 independent chains", cus);
    run<8>("packed fma, one dependent chain", cus);
    run<9>("packed mul + scalar add alternating", cus);
    run<10>("packed mul/add with op_sel / neg modifiers", cus);
    run<11>("scalar chain, literal constant operand", cus);
    run<12>("packed chain, register constants", cus);
    return 0;
}
