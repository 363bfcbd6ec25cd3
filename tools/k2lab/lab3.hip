// tools/k2lab/lab3.hip -- round 3 of the decimator lab: tile shapes (R, NT), scheduler occupancy hints, store policy and
// input data classes, for cfloat and u8 IQ input, each with an in-kernel shader-clock probe (the kernel is power-limited:
// what matters is energy per MAC, visible as the clock the chip sustains).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isdr_amd/csrc tools/k2lab/lab3.hip -o tools/k2lab/lab3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../sdr_amd/csrc/kernels_fast.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace sdrhip;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ClkProbe { unsigned long long cyc, rt; unsigned int n; unsigned int pad; };
__device__ __forceinline__ void probe_begin(unsigned long long& c0, unsigned long long& r0)
{
    c0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
}
__device__ __forceinline__ void probe_end(ClkProbe* pr, unsigned long long c0, unsigned long long r0)
{
    if (pr != nullptr && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
        atomicAdd(&pr->cyc, c1 - c0);
        atomicAdd(&pr->rt, r1 - r0);
        atomicAdd(&pr->n, 1u);
    }
}

template <int NT, bool NTL, int NIN>
__global__ void __launch_bounds__(NT) k_stream(const uint4* __restrict__ in, uint4* __restrict__ out)
{
    const size_t base = (size_t)blockIdx.x * NT * NIN + threadIdx.x;
    uint4 v[NIN];
#pragma unroll
    for (int i = 0; i < NIN; i++) {
        if constexpr (NTL) {
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + base + (size_t)i * NT));
            v[i] = make_uint4(t.x, t.y, t.z, t.w);
        } else {
            v[i] = in[base + (size_t)i * NT];
        }
    }
    uint4 r = v[0];
#pragma unroll
    for (int i = 1; i < NIN; i++) { r.x ^= v[i].x; r.y ^= v[i].y; r.z ^= v[i].z; r.w ^= v[i].w; }
    out[(size_t)blockIdx.x * NT + threadIdx.x] = r;
}

template <int NT>
__global__ void __launch_bounds__(NT) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <int D, int P, int R, class T, bool NTS>
__device__ __forceinline__ void compute_tile(const float2* __restrict__ lds, const float* __restrict__ taps, float* __restrict__ out, int out0)
{
    const float2* win = lds + T::lds_idx(threadIdx.x * T::CHUNK);
    float2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[r][k] = make_float2(0.0f, 0.0f);
    mac_window<D, P, R, T, 8, false>(win, taps, acc, 0);
    const int o = out0 + threadIdx.x * R;
    float2 res[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        res[r].x = (acc[r][0].x + acc[r][1].x) + (acc[r][2].x + acc[r][3].x);
        res[r].y = (acc[r][0].y + acc[r][1].y) + (acc[r][2].y + acc[r][3].y);
    }
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) {
        if constexpr (NTS) {
            const f32x4 v = {res[r].x, res[r].y, res[r + 1].x, res[r + 1].y};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + 2 * (int64_t)o) + r / 2);
        } else {
            reinterpret_cast<float4*>(out + 2 * (int64_t)o)[r / 2] = make_float4(res[r].x, res[r].y, res[r + 1].x, res[r + 1].y);
        }
    }
}

// the production structure (one tile per workgroup, register staging) with the staging registers as plain local arrays
// and no ragged-end code in the kernel (whole tiles only)
template <int D, int P, int R, int NT, bool U8, int MINW, bool NTS>
__global__ void __launch_bounds__(NT, MINW) k_dec_x(const void* __restrict__ in, int ntiles, const float* __restrict__ taps, float* __restrict__ out,
                                                    ClkProbe* pr)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    constexpr int SPV = U8 ? 8 : 2;
    constexpr int NV = (T::SPAN + SPV - 1) / SPV;
    constexpr int PER = (NV + NT - 1) / NT;
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(in) + (U8 ? 2 : 8) * (int64_t)tile * T::OUTS * D);
    uint4 r[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        r[i] = (i + 1 < PER || v < NV) ? src[v] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        const int s = v * SPV;
        if (v < NV) {
            if constexpr (!U8) {
                *reinterpret_cast<uint4*>(&lds[T::lds_idx(s)]) = r[i];
            } else {
                const uint32_t w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float4 f;
                    f.x = __builtin_fmaf((float)(w[k] & 0xff), 1.0f / 128.0f, -1.0f);
                    f.y = __builtin_fmaf((float)((w[k] >> 8) & 0xff), 1.0f / 128.0f, -1.0f);
                    f.z = __builtin_fmaf((float)((w[k] >> 16) & 0xff), 1.0f / 128.0f, -1.0f);
                    f.w = __builtin_fmaf((float)(w[k] >> 24), 1.0f / 128.0f, -1.0f);
                    const int ss = s + 2 * k;
                    if (ss < T::SPAN + 1) *reinterpret_cast<float4*>(&lds[T::lds_idx(ss)]) = f;
                }
            }
        }
    }
    __syncthreads();
    compute_tile<D, P, R, T, NTS>(lds, taps, out, tile * T::OUTS);
    probe_end(pr, c0, r0);
}

template <int D, int P, int R, int NT, int MINW>
__global__ void __launch_bounds__(NT, MINW) k_mac_x(int ntiles, const float* __restrict__ taps, float* __restrict__ out, ClkProbe* pr)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    compute_tile<D, P, R, T, false>(lds, taps, out, tile * T::OUTS);
    probe_end(pr, c0, r0);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double us(F f, int reps, int warm = 2)
    {
        for (int i = 0; i < warm; i++) f();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        return ms * 1e3 / reps;
    }
};

static uint64_t sm64(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

static int g_reps = 20;
static int64_t g_n = 0, g_nout = 0;
static Timer* g_tm;
static ClkProbe* g_dpr;
static float *g_dt, *g_dout, *g_dref;
static std::vector<uint64_t> g_href, g_hout;
static double g_rd_per_sample = 8.0;

static void report(const char* name, double us, bool probe)
{
    printf("%-46s %8.1f us  %7.1f Gsamp/s  read %5.3f TB/s (%.3f of 8)", name, us, g_n / us / 1e3, g_rd_per_sample * g_n / us / 1e6,
           g_rd_per_sample * g_n / us / 1e6 / 8.0);
    if (probe) {
        ClkProbe h;
        CK(hipMemcpy(&h, g_dpr, sizeof h, hipMemcpyDeviceToHost));
        if (h.rt) printf("  clk %4.0f MHz, %.2f us/WG", (double)h.cyc / (double)h.rt * 100.0, (double)h.rt / h.n / 100.0);
    }
    printf("\n");
    fflush(stdout);
}

static void check(const char* name)
{
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(g_hout.data(), g_dout, (size_t)g_nout * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < (size_t)g_nout; i++) bad += g_hout[i] != g_href[i];
    if (bad) printf("    check %-40s MISMATCH (%zu of %lld outputs differ)\n", name, bad, (long long)g_nout);
    CK(hipMemset(g_dout, 0xff, (size_t)g_nout * 8));
}

template <int R, int NT, bool U8, int MINW, bool NTS>
static void run_dec(const void* din, const char* name, bool chk)
{
    constexpr int D = 8, P = 128;
    using T = Tile<D, P, R, NT>;
    auto k = k_dec_x<D, P, R, NT, U8, MINW, NTS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    const int ntiles = (int)(g_nout / T::OUTS), grid = ((ntiles + 63) / 64) * 64;
    CK(hipMemset(g_dpr, 0, sizeof(ClkProbe)));
    const double us = g_tm->us([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(NT), T::LDS_BYTES, 0, din, ntiles, g_dt, g_dout, g_dpr); }, g_reps);
    report(name, us, true);
    if (chk) check(name);
}

template <int R, int NT, int MINW>
static void run_mac(const char* name)
{
    constexpr int D = 8, P = 128;
    using T = Tile<D, P, R, NT>;
    auto k = k_mac_x<D, P, R, NT, MINW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    const int ntiles = (int)(g_nout / T::OUTS), grid = ((ntiles + 63) / 64) * 64;
    CK(hipMemset(g_dpr, 0, sizeof(ClkProbe)));
    const double us = g_tm->us([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(NT), T::LDS_BYTES, 0, ntiles, g_dt, g_dout, g_dpr); }, g_reps);
    report(name, us, true);
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 27;
    g_reps = argc > 2 ? atoi(argv[2]) : 20;
    const int rounds = argc > 3 ? atoi(argv[3]) : 2;
    constexpr int D = 8, P = 128;
    const int64_t n = (int64_t)1 << log2n;
    g_n = n;
    const int64_t nout = n / D - 1024;                       // whole tiles of 512 and 1024 outputs, windows inside the buffer
    g_nout = nout;
    const int64_t n_alloc = n + 8192;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("n = 2^%d samples, %lld outputs\n", log2n, (long long)nout);

    // data classes
    std::vector<float> hx((size_t)2 * n_alloc), hq((size_t)2 * n_alloc);
    std::vector<uint8_t> hu((size_t)2 * n_alloc);
    uint64_t seed = 1002;
    for (size_t i = 0; i < hx.size(); i++) {
        const uint64_t z = sm64(seed);
        hx[i] = (float)((double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0);
        hu[i] = (uint8_t)(z & 0xff);
        hq[i] = ((float)hu[i] - 128.0f) * (1.0f / 128.0f);
    }
    std::vector<float> ht(P, 0.0f);
    for (int j = 0; j < 127; j++) {
        const double m = j - 63.0, fc = 1.0 / 16.0;
        const double s = m == 0 ? 2 * fc : sin(2 * M_PI * fc * m) / (M_PI * m);
        ht[j] = (float)(s * (0.54 - 0.46 * cos(2 * M_PI * j / 126.0)));
    }
    float *dx, *dq;
    uint8_t* du;
    CK(hipMalloc(&dx, hx.size() * 4));
    CK(hipMalloc(&dq, hq.size() * 4));
    CK(hipMalloc(&du, hu.size()));
    CK(hipMalloc(&g_dt, P * 4));
    CK(hipMalloc(&g_dref, (size_t)nout * 8));
    CK(hipMalloc(&g_dout, (size_t)nout * 8));
    CK(hipMalloc(&g_dpr, sizeof(ClkProbe)));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, hu.data(), hu.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(g_dt, ht.data(), P * 4, hipMemcpyHostToDevice));
    Timer tm;
    g_tm = &tm;
    g_href.resize((size_t)nout);
    g_hout.resize((size_t)nout);

    using T0 = Tile<D, P, 2, 256>;
    auto k0 = k_decimate_c4<D, P, 2, 256, false>;
    auto k0u = k_decimate_c4<D, P, 2, 256, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T0::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k0u), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T0::LDS_BYTES));
    const int nt0 = (int)(nout / T0::OUTS), grid0 = ((nt0 + 63) / 64) * 64;
    auto run0 = [&](const void* in, bool u8, float* o) {
        if (u8) hipLaunchKernelGGL(k0u, dim3(grid0), dim3(256), T0::LDS_BYTES, 0, in, (int64_t)0, (int)nout, g_dt, o, P, 0, 0, 0);
        else hipLaunchKernelGGL(k0, dim3(grid0), dim3(256), T0::LDS_BYTES, 0, in, (int64_t)0, (int)nout, g_dt, o, P, 0, 0, 0);
    };
    auto set_ref = [&](const void* in, bool u8) {
        run0(in, u8, g_dref);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(g_href.data(), g_dref, (size_t)nout * 8, hipMemcpyDeviceToHost));
    };
    const int ncu = prop.multiProcessorCount;

    for (int round = 0; round < rounds; round++) {
        printf("==== round %d\n", round);
        // ---------------- cfloat in, uniform f32 data
        g_rd_per_sample = 8.0;
        printf("-- cfloat IQ in, uniform [-1,1) f32 data\n");
        set_ref(dx, false);
        {
            const int nchunks = (int)(n * 8 / 16 / (256 * 8));
            report("stream 8:1 plain", tm.us([&] { hipLaunchKernelGGL((k_stream<256, false, 8>), dim3(nchunks), dim3(256), 0, 0, (const uint4*)dx, (uint4*)g_dout); }, g_reps), false);
            report("stream 8:1 nt", tm.us([&] { hipLaunchKernelGGL((k_stream<256, true, 8>), dim3(nchunks), dim3(256), 0, 0, (const uint4*)dx, (uint4*)g_dout); }, g_reps), false);
        }
        CK(hipMemset(g_dout, 0xff, (size_t)nout * 8));
        report("V0 production", tm.us([&] { run0(dx, false, g_dout); }, g_reps), false);
        check("V0");
        run_dec<2, 256, false, 1, false>(dx, "R2 NT256 minw1", true);
        run_dec<2, 256, false, 4, false>(dx, "R2 NT256 minw4", true);
        run_dec<2, 256, false, 4, true>(dx, "R2 NT256 minw4 nt-stores", true);
        run_dec<4, 256, false, 2, false>(dx, "R4 NT256 minw2 (2 WG/CU)", true);
        run_dec<4, 256, false, 2, true>(dx, "R4 NT256 minw2 nt-stores", true);
        run_dec<4, 128, false, 2, false>(dx, "R4 NT128 minw2 (4 WG/CU)", true);
        run0(dx, false, g_dout);
        run_mac<2, 256, 1>("MAC only R2 NT256 minw1");
        run_mac<2, 256, 4>("MAC only R2 NT256 minw4");
        run_mac<4, 256, 2>("MAC only R4 NT256 minw2");
        run_mac<4, 128, 2>("MAC only R4 NT128 minw2");
        // ---------------- cfloat in, data = convert(u8 IQ)
        printf("-- cfloat IQ in, data = convert(u8 IQ) (what the FM pipeline feeds this stage)\n");
        set_ref(dq, false);
        CK(hipMemset(g_dout, 0xff, (size_t)nout * 8));
        report("V0 production", tm.us([&] { run0(dq, false, g_dout); }, g_reps), false);
        check("V0");
        run_dec<2, 256, false, 4, false>(dq, "R2 NT256 minw4", true);
        run_dec<4, 256, false, 2, false>(dq, "R4 NT256 minw2 (2 WG/CU)", true);
        run0(dq, false, g_dout);
        run_mac<2, 256, 4>("MAC only R2 NT256 minw4");
        run_mac<4, 256, 2>("MAC only R4 NT256 minw2");
        // ---------------- u8 in
        g_rd_per_sample = 2.0;
        printf("-- u8 IQ in (convert fused)\n");
        set_ref(du, true);
        {
            const int nchunks = (int)(n * 2 / 16 / (256 * 2));
            report("stream 2:1 plain", tm.us([&] { hipLaunchKernelGGL((k_stream<256, false, 2>), dim3(nchunks), dim3(256), 0, 0, (const uint4*)du, (uint4*)g_dout); }, g_reps), false);
        }
        CK(hipMemset(g_dout, 0xff, (size_t)nout * 8));
        report("V0 production u8", tm.us([&] { run0(du, true, g_dout); }, g_reps), false);
        check("V0 u8");
        run_dec<2, 256, true, 1, false>(du, "u8 R2 NT256 minw1", true);
        run_dec<2, 256, true, 4, false>(du, "u8 R2 NT256 minw4", true);
        run_dec<4, 256, true, 2, false>(du, "u8 R4 NT256 minw2 (2 WG/CU)", true);
        run_dec<4, 128, true, 2, false>(du, "u8 R4 NT128 minw2 (4 WG/CU)", true);
        run_dec<2, 128, true, 4, false>(du, "u8 R2 NT128 minw4 (8 WG/CU)", true);
    }
    {
        const size_t nv = (size_t)(9.0 * n / 2 / 16);
        uint4 *ca, *cb;
        CK(hipMalloc(&ca, nv * 16));
        CK(hipMalloc(&cb, nv * 16));
        CK(hipMemset(ca, 1, nv * 16));
        for (int per : {8, 32}) {
            const double us = tm.us([&] { hipLaunchKernelGGL((k_copy<256>), dim3(ncu * per), dim3(256), 0, 0, ca, cb, nv); }, g_reps);
            printf("float4 copy %zu MiB -> same, grid x%d/CU: %8.1f us  total %5.3f TB/s\n", nv * 16 >> 20, per, us, 2.0 * nv * 16 / us / 1e6);
        }
    }
    return 0;
}
