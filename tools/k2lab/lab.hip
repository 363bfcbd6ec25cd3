// tools/k2lab/lab.hip -- development bench for the decimate-by-8 kernel fed cfloat IQ (BASELINE configs[1]).
// Not part of the product: variants of the staging pipeline around the SAME mac_window arithmetic, each checked
// bit-for-bit against the production kernel's output, next to streaming kernels of the same traffic shape
// (8 bytes read : 1 byte written) that measure what the memory system delivers on this box.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isdr_amd/csrc tools/k2lab/lab.hip -o tools/k2lab/lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../sdr_amd/csrc/kernels_fast.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace sdrhip;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NTL>
__device__ __forceinline__ uint4 ld16(const uint4* p)
{
    if constexpr (NTL) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}

// ------------------------------------------------------------------------------------------------
// streaming kernels, 8 : 1
// ------------------------------------------------------------------------------------------------
// one-shot: thread reads 8 uint4 (wave-contiguous 1 KiB each), xors them, writes one uint4
template <int NT, bool NTL>
__global__ void __launch_bounds__(NT) k_stream_oneshot(const uint4* __restrict__ in, uint4* __restrict__ out)
{
    const size_t base = (size_t)blockIdx.x * NT * 8 + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = ld16<NTL>(in + base + (size_t)i * NT);
    uint4 r = v[0];
#pragma unroll
    for (int i = 1; i < 8; i++) { r.x ^= v[i].x; r.y ^= v[i].y; r.z ^= v[i].z; r.w ^= v[i].w; }
    out[(size_t)blockIdx.x * NT + threadIdx.x] = r;
}

// persistent: `gridDim.x` workgroups walk the buffer in the XCD-aware order of the decimator, next chunk's loads
// in flight while the current one is reduced
template <int NT, bool NTL>
__global__ void __launch_bounds__(NT) k_stream_persist(const uint4* __restrict__ in, uint4* __restrict__ out, int nchunks)
{
    const int nwg = gridDim.x;
    auto chunk_of = [&](int it) { const int g = it * nwg + blockIdx.x; return (g & ~63) + ((g & 7) << 3) + ((g >> 3) & 7); };
    int it = 0;
    int c = chunk_of(0);
    uint4 v[8];
    if (c < nchunks) {
        const size_t base = (size_t)c * NT * 8 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ld16<NTL>(in + base + (size_t)i * NT);
    }
    while (c < nchunks) {
        const int n = chunk_of(++it);
        uint4 w[8];
        if (n < nchunks) {
            const size_t base = (size_t)n * NT * 8 + threadIdx.x;
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = ld16<NTL>(in + base + (size_t)i * NT);
        }
        uint4 r = v[0];
#pragma unroll
        for (int i = 1; i < 8; i++) { r.x ^= v[i].x; r.y ^= v[i].y; r.z ^= v[i].z; r.w ^= v[i].w; }
        out[(size_t)c * NT + threadIdx.x] = r;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = w[i];
        c = n;
    }
}

// LDS-DMA: every wave streams 8 KiB per chunk straight into LDS (no VGPR round trip), double-buffered
template <int NT>
__global__ void __launch_bounds__(NT) k_stream_dma(const uint4* __restrict__ in, uint4* __restrict__ out, int nchunks)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[2][8 * NT];
    const int nwg = gridDim.x;
    auto chunk_of = [&](int it) { const int g = it * nwg + blockIdx.x; return (g & ~63) + ((g & 7) << 3) + ((g >> 3) & 7); };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto issue = [&](int c, int b) {
        const uint4* src = in + (size_t)c * NT * 8 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 8; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)i * NT), (lptr_t)(&lds[b][i * NT + wave * 64]), 16, 0, 0);
    };
    int it = 0;
    int c = chunk_of(0);
    if (c < nchunks) issue(c, 0);
    __syncthreads();
    while (c < nchunks) {
        const int n = chunk_of(++it);
        if (n < nchunks) issue(n, it & 1);
        const uint4* b = lds[(it - 1) & 1];
        uint4 r = b[threadIdx.x];
#pragma unroll
        for (int i = 1; i < 8; i++) { const uint4 v = b[i * NT + threadIdx.x]; r.x ^= v.x; r.y ^= v.y; r.z ^= v.z; r.w ^= v.w; }
        out[(size_t)c * NT + threadIdx.x] = r;
        __syncthreads();
        c = n;
    }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// ------------------------------------------------------------------------------------------------
// decimator variants.  All assume whole tiles and SPAN samples available behind every tile start (the lab
// pads the input); the production kernel handles the ragged end.
// ------------------------------------------------------------------------------------------------
template <int D, int P, int R, class T>
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
    float4* dst = reinterpret_cast<float4*>(out + 2 * (int64_t)o);
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) dst[r / 2] = make_float4(res[r].x, res[r].y, res[r + 1].x, res[r + 1].y);
}

// V1: persistent workgroups, next tile's loads in flight in registers during the MAC phase
template <int D, int P, int R, int NT, bool NTL, int WPE>
__global__ void __launch_bounds__(NT, WPE) k_dec_persist_reg(const float* __restrict__ in, int ntiles, const float* __restrict__ taps,
                                                        float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int nwg = gridDim.x;
    auto tile_of = [&](int it) { const int g = it * nwg + blockIdx.x; return (g & ~63) + ((g & 7) << 3) + ((g >> 3) & 7); };
    constexpr int NV = (T::SPAN + 1) / 2;
    constexpr int PER = (NV + NT - 1) / NT;
    uint4 r[PER];
    auto load = [&](int tile) {
        const uint4* src = reinterpret_cast<const uint4*>(in + 2 * (int64_t)tile * T::OUTS * D);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int v = threadIdx.x + i * NT;
            if (i + 1 < PER || v < NV) r[i] = ld16<NTL>(src + v);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int v = threadIdx.x + i * NT;
            if (v < NV) *reinterpret_cast<uint4*>(&lds[T::lds_idx(2 * v)]) = r[i];
        }
    };
    int it = 0;
    int tile = tile_of(0);
    if (tile < ntiles) { load(tile); store(); }
    __syncthreads();
    while (tile < ntiles) {
        const int next = tile_of(++it);
        if (next < ntiles) load(next);
        compute_tile<D, P, R, T>(lds, taps, out, tile * T::OUTS);
        __syncthreads();
        if (next < ntiles) store();
        __syncthreads();
        tile = next;
    }
}

// V2: persistent workgroups, LDS-DMA into a double-buffered tile.  The padded LDS layout (one 16-byte pad per eight
// 16-byte vectors) is produced on the SOURCE side: LDS vector slot q holds source vector (q/9)*8 + q%9; the pad slots
// (q%9 == 8) re-fetch their left neighbour's vector (same cache line, never read back).
template <int D, int P, int R, int NT, int AUX>
__global__ void __launch_bounds__(NT) k_dec_persist_dma(const float* __restrict__ in, int ntiles, const float* __restrict__ taps,
                                                        float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    static_assert(T::CHUNK == 16, "pad pattern below: one pad vector per 8 data vectors");
    constexpr int NVL = (T::LDS_F2 + 1) / 2;                 // 16-byte slots of the padded tile
    constexpr int NI = (NVL + NT - 1) / NT;                  // DMA instructions per thread
    constexpr int BUF_F2 = NI * NT * 2;                      // float2 per buffer (whole instructions)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int nwg = gridDim.x;
    auto tile_of = [&](int it) { const int g = it * nwg + blockIdx.x; return (g & ~63) + ((g & 7) << 3) + ((g >> 3) & 7); };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NV = (T::SPAN + 1) / 2;
    int voff[NI];                                            // source vector of this lane's slot in instruction i
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int q = i * NT + threadIdx.x;
        const int rr = q % 9;
        int v = (q / 9) * 8 + (rr == 8 ? 7 : rr);
        if (v > NV - 1) v = NV - 1;
        voff[i] = v;
    }
    auto issue = [&](int tile, int b) {
        const uint4* src = reinterpret_cast<const uint4*>(in + 2 * (int64_t)tile * T::OUTS * D);
#pragma unroll
        for (int i = 0; i < NI; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + voff[i]), (lptr_t)(lds + b * BUF_F2 + (i * NT + wave * 64) * 2), 16, 0, AUX);
    };
    int it = 0;
    int tile = tile_of(0);
    if (tile < ntiles) issue(tile, 0);
    __syncthreads();
    while (tile < ntiles) {
        const int next = tile_of(++it);
        if (next < ntiles) issue(next, it & 1);
        compute_tile<D, P, R, T>(lds + ((it - 1) & 1) * BUF_F2, taps, out, tile * T::OUTS);
        __syncthreads();      // drains this wave's DMA (vmcnt(0)) and orders every wave's reads before the next overwrite
        tile = next;
    }
}

// MAC phase alone: no global loads, LDS holds whatever the previous kernel left (the V0 run just before: real samples)
template <int D, int P, int R, int NT>
__global__ void __launch_bounds__(NT) k_dec_nold(int ntiles, const float* __restrict__ taps, float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    compute_tile<D, P, R, T>(lds, taps, out, tile * T::OUTS);
}

// V3: non-persistent, LDS-DMA single buffer (what the DMA alone buys over register staging)
template <int D, int P, int R, int NT, int AUX>
__global__ void __launch_bounds__(NT) k_dec_oneshot_dma(const float* __restrict__ in, int ntiles, const float* __restrict__ taps,
                                                        float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    constexpr int NVL = (T::LDS_F2 + 1) / 2;
    constexpr int NI = (NVL + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NV = (T::SPAN + 1) / 2;
    const uint4* src = reinterpret_cast<const uint4*>(in + 2 * (int64_t)tile * T::OUTS * D);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int q = i * NT + threadIdx.x;
        const int rr = q % 9;
        int v = (q / 9) * 8 + (rr == 8 ? 7 : rr);
        if (v > NV - 1) v = NV - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + v), (lptr_t)(lds + (i * NT + wave * 64) * 2), 16, 0, AUX);
    }
    __syncthreads();
    compute_tile<D, P, R, T>(lds, taps, out, tile * T::OUTS);
}

// ------------------------------------------------------------------------------------------------
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

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 27;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int rounds = argc > 3 ? atoi(argv[3]) : 2;
    constexpr int D = 8, P = 128, R = 2, NT = 256;
    using T = Tile<D, P, R, NT>;
    const int64_t n = (int64_t)1 << log2n;
    const int ntiles = (int)(n / (T::OUTS * D));             // whole tiles; the last tile's window runs 120 samples past n (padded)
    const int64_t n_alloc = n + 8192;
    const int64_t nout = (int64_t)ntiles * T::OUTS;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; n = 2^%d samples, %d tiles, LDS/tile %zu B\n", prop.name, prop.multiProcessorCount, log2n, ntiles, T::LDS_BYTES);

    std::vector<float> hx((size_t)2 * n_alloc);
    uint64_t seed = 1002;
    for (auto& v : hx) v = (float)((double)(sm64(seed) >> 11) * (2.0 / 9007199254740992.0) - 1.0);
    std::vector<float> ht(P, 0.0f);
    for (int j = 0; j < 127; j++) {
        const double m = j - 63.0, fc = 1.0 / 16.0;
        const double s = m == 0 ? 2 * fc : sin(2 * M_PI * fc * m) / (M_PI * m);
        ht[j] = (float)(s * (0.54 - 0.46 * cos(2 * M_PI * j / 126.0)));
    }
    float *dx, *dt, *dref, *dout;
    CK(hipMalloc(&dx, hx.size() * 4));
    CK(hipMalloc(&dt, P * 4));
    CK(hipMalloc(&dref, (size_t)nout * 8));
    CK(hipMalloc(&dout, (size_t)nout * 8));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), P * 4, hipMemcpyHostToDevice));
    Timer tm;
    std::vector<uint64_t> href((size_t)nout), hout((size_t)nout);

    // ---- production kernel = reference output
    auto k0 = k_decimate_c4<D, P, R, NT, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    const int grid0 = ((ntiles + 63) / 64) * 64;
    auto run0 = [&](float* o) { hipLaunchKernelGGL(k0, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, o, P); };
    run0(dref);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(href.data(), dref, (size_t)nout * 8, hipMemcpyDeviceToHost));

    auto check = [&](const char* name) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), dout, (size_t)nout * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)nout; i++) bad += hout[i] != href[i];
        printf("    check %-28s %s (%zu of %lld outputs differ)\n", name, bad ? "MISMATCH" : "bit-exact", bad, (long long)nout);
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
    };
    const double rd_bytes = 8.0 * n, wr_bytes = 1.0 * n;
    auto report = [&](const char* name, double us) {
        printf("%-34s %8.1f us  %7.1f Gsamp/s  read %5.3f TB/s (%.3f of 8)  total %5.3f TB/s\n", name, us, n / us / 1e3, rd_bytes / us / 1e6,
               rd_bytes / us / 1e6 / 8.0, (rd_bytes + wr_bytes) / us / 1e6);
        fflush(stdout);
    };

    // LDS sizes of the DMA variants
    constexpr int NVL = (T::LDS_F2 + 1) / 2, NI = (NVL + NT - 1) / NT;
    constexpr size_t DMA_BUF = (size_t)NI * NT * 16;
    const int ncu = prop.multiProcessorCount;

    // streaming shapes: 8 uint4 in, 1 out per thread
    const int nchunks = (int)(n * 8 / 16 / (NT * 8));
    uint4* sin_ = reinterpret_cast<uint4*>(dx);
    uint4* sout = reinterpret_cast<uint4*>(dout);

#define SETLDS(k, bytes) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
    auto kreg3 = k_dec_persist_reg<D, P, R, NT, false, 3>;
    auto kregnt3 = k_dec_persist_reg<D, P, R, NT, true, 3>;
    auto kreg4 = k_dec_persist_reg<D, P, R, NT, false, 4>;
    auto kregnt4 = k_dec_persist_reg<D, P, R, NT, true, 4>;
    auto knold = k_dec_nold<D, P, R, NT>;
    SETLDS(knold, T::LDS_BYTES);
    auto kdma = k_dec_persist_dma<D, P, R, NT, 0>;
    auto kdmant = k_dec_persist_dma<D, P, R, NT, 2>;
    auto kone = k_dec_oneshot_dma<D, P, R, NT, 0>;
    SETLDS(kreg3, T::LDS_BYTES); SETLDS(kregnt3, T::LDS_BYTES); SETLDS(kreg4, T::LDS_BYTES); SETLDS(kregnt4, T::LDS_BYTES); SETLDS(kdma, 2 * DMA_BUF); SETLDS(kdmant, 2 * DMA_BUF); SETLDS(kone, DMA_BUF);

    for (int round = 0; round < rounds; round++) {
        printf("---- round %d\n", round);
        report("stream oneshot 8:1", tm.us([&] { hipLaunchKernelGGL((k_stream_oneshot<NT, false>), dim3(nchunks), dim3(NT), 0, 0, sin_, sout); }, reps));
        report("stream oneshot 8:1 nt", tm.us([&] { hipLaunchKernelGGL((k_stream_oneshot<NT, true>), dim3(nchunks), dim3(NT), 0, 0, sin_, sout); }, reps));
        for (int per : {4, 6, 8}) {
            char nm[64];
            snprintf(nm, sizeof nm, "stream persist 8:1 x%d/CU", per);
            report(nm, tm.us([&] { hipLaunchKernelGGL((k_stream_persist<NT, false>), dim3(ncu * per), dim3(NT), 0, 0, sin_, sout, nchunks); }, reps));
            snprintf(nm, sizeof nm, "stream persist 8:1 nt x%d/CU", per);
            report(nm, tm.us([&] { hipLaunchKernelGGL((k_stream_persist<NT, true>), dim3(ncu * per), dim3(NT), 0, 0, sin_, sout, nchunks); }, reps));
        }
        for (int per : {2, 4}) {
            char nm[64];
            snprintf(nm, sizeof nm, "stream LDS-DMA 8:1 x%d/CU", per);
            report(nm, tm.us([&] { hipLaunchKernelGGL((k_stream_dma<NT>), dim3(ncu * per), dim3(NT), 0, 0, sin_, sout, nchunks); }, reps));
        }
        {
            // plain float4 copy moving the same total bytes (576 MiB -> 576 MiB at n = 2^27)
            const size_t nv = (size_t)((rd_bytes + wr_bytes) / 2 / 16);
            uint4* cdst = reinterpret_cast<uint4*>(dx) + nv;      // second half of the input buffer (restored below: not needed, lab data only read after)
            (void)cdst;
        }
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));

        report("V0 production (4 WG/CU, reg stage)", tm.us([&] { run0(dout); }, reps));
        check("V0");
        report("MAC phase only (no loads, stale LDS)", tm.us([&] { hipLaunchKernelGGL(knold, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, ntiles, dt, dout); }, reps));
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        for (int per : {3, 4}) {
            char nm[64];
            snprintf(nm, sizeof nm, "V1 persist reg-prefetch x%d/CU", per);
            report(nm, tm.us([&] { hipLaunchKernelGGL(per == 3 ? kreg3 : kreg4, dim3(ncu * per), dim3(NT), T::LDS_BYTES, 0, dx, ntiles, dt, dout); }, reps));
            check(nm);
            snprintf(nm, sizeof nm, "V1 persist reg-prefetch nt x%d/CU", per);
            report(nm, tm.us([&] { hipLaunchKernelGGL(per == 3 ? kregnt3 : kregnt4, dim3(ncu * per), dim3(NT), T::LDS_BYTES, 0, dx, ntiles, dt, dout); }, reps));
            check(nm);
        }
        {
            using T1 = Tile<D, P, 2, 128>;
            auto k = k_decimate_c4<D, P, 2, 128, false>;
            SETLDS(k, T1::LDS_BYTES);
            const int nt1 = (int)(nout / T1::OUTS), g1 = ((nt1 + 63) / 64) * 64;
            report("V0 shape NT=128 R=2 (8 WG/CU)", tm.us([&] { hipLaunchKernelGGL(k, dim3(g1), dim3(128), T1::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, dout, P); }, reps));
            check("V0 NT=128 R=2");
        }
        {
            using T1 = Tile<D, P, 4, 128>;
            auto k = k_decimate_c4<D, P, 4, 128, false>;
            SETLDS(k, T1::LDS_BYTES);
            const int nt1 = (int)(nout / T1::OUTS), g1 = ((nt1 + 63) / 64) * 64;
            report("V0 shape NT=128 R=4 (4 WG/CU)", tm.us([&] { hipLaunchKernelGGL(k, dim3(g1), dim3(128), T1::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, dout, P); }, reps));
            check("V0 NT=128 R=4");
        }
        report("V2 persist LDS-DMA dbuf x2/CU", tm.us([&] { hipLaunchKernelGGL(kdma, dim3(ncu * 2), dim3(NT), 2 * DMA_BUF, 0, dx, ntiles, dt, dout); }, reps));
        check("V2");
        report("V2 persist LDS-DMA dbuf nt x2/CU", tm.us([&] { hipLaunchKernelGGL(kdmant, dim3(ncu * 2), dim3(NT), 2 * DMA_BUF, 0, dx, ntiles, dt, dout); }, reps));
        check("V2 nt");
        report("V3 oneshot LDS-DMA (4 WG/CU)", tm.us([&] { hipLaunchKernelGGL(kone, dim3(grid0), dim3(NT), DMA_BUF, 0, dx, ntiles, dt, dout); }, reps));
        check("V3");
    }
    // plain copy, same total bytes, separate buffers
    {
        const size_t nv = (size_t)((rd_bytes + wr_bytes) / 2 / 16);
        uint4 *ca, *cb;
        CK(hipMalloc(&ca, nv * 16));
        CK(hipMalloc(&cb, nv * 16));
        CK(hipMemset(ca, 1, nv * 16));
        for (int per : {8, 16, 32}) {
            const double us = tm.us([&] { hipLaunchKernelGGL((k_copy<NT>), dim3(ncu * per), dim3(NT), 0, 0, ca, cb, nv); }, reps);
            printf("float4 copy %zu MiB -> same, grid x%d/CU: %8.1f us  total %5.3f TB/s\n", nv * 16 >> 20, per, us, 2.0 * nv * 16 / us / 1e6);
        }
        const double us = tm.us([&] { CK(hipMemcpyAsync(cb, ca, nv * 16, hipMemcpyDeviceToDevice, 0)); }, reps);
        printf("hipMemcpyAsync D2D same size: %8.1f us  total %5.3f TB/s\n", us, 2.0 * nv * 16 / us / 1e6);
    }
    return 0;
}
