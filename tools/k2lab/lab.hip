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

namespace sdrhip { int small_launch_outputs() { return 32768; } }   // lives in abi_device.cpp in the library
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

// one-shot copy: every thread moves 4 x 16 B, all four loads in flight before the first store
template <int NT, bool NTL>
__global__ void __launch_bounds__(NT) k_copy4(const uint4* __restrict__ a, uint4* __restrict__ b)
{
    const size_t base = (size_t)blockIdx.x * NT * 4 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = ld16<NTL>(a + base + (size_t)i * NT);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        if (NTL) __builtin_nontemporal_store(u4v{v[i].x, v[i].y, v[i].z, v[i].w}, reinterpret_cast<u4v*>(b + base + (size_t)i * NT));
        else b[base + (size_t)i * NT] = v[i];
    }
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


// ------------------------------------------------------------------------------------------------
// mac_window2: the same arithmetic as mac_window (kernels_fast.hip) with the wait/issue order pinned.
//   block b:  [W] pin buf[b] and tc[b] (the compiler must put its s_waitcnt HERE: everything outstanding was issued a whole
//             block of MACs ago)  ->  issue tc[b+1], LDS(b+DEPTH)  ->  MACs of block b.
// mac_window issues the next reads first and then meets the wait for the CURRENT block's operands, and because a scalar
// load is outstanding that wait is lgkmcnt(0): it also waits for the reads it has just issued.
// Probes: NOLDS (the window is read once, every block reuses it) and NOSMEM (taps are literals) time the loop without
// one of its two operand streams; their results are wrong on purpose.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void pin_f4(float4 (&v)[N])
{
    static_assert(N == 4, "four float4 per block");
    asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z), "+v"(v[1].w),
                      "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w)
                 :: "memory");
}

template <int D, int P, int R, class T, int DEPTH, bool PIN, bool NOLDS, bool NOSMEM>
__device__ __forceinline__ void mac_window2(const float2* __restrict__ win, const float* __restrict__ taps, float2 (&acc)[R][4])
{
    constexpr int TC = 8;
    constexpr int NCH = P / TC;
    constexpr int NB = T::WIN / TC;
    constexpr int NBUF = DEPTH + 1;
    typedef typename TapVec<TC>::type tapv;
    tapv tc[NCH];
    float4 buf[NBUF][TC / 2];
    auto tap_chunk = [&](int c) -> tapv {
        if constexpr (NOSMEM) {
            tapv t;
#pragma unroll
            for (int k = 0; k < TC; k++) t[k] = 0.001f * (float)(c * TC + k + 1);
            return t;
        } else {
            return load_tap_chunk<TC>(taps, c);
        }
    };
#define LDS_BLOCK(BB, DST)                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < TC / 2; i_++) {                                                 \
        const int s_ = TC * (NOLDS ? 0 : (BB)) + 2 * i_;                                                    \
        (DST)[i_] = *reinterpret_cast<const float4*>(&win[s_ + 2 * (s_ / T::CHUNK)]);                       \
    }
    tc[0] = tap_chunk(0);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { LDS_BLOCK(d, buf[d % NBUF]); }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if constexpr (PIN) {
            pin_f4(buf[b % NBUF]);
            if constexpr (!NOSMEM) { if (b < NCH) asm volatile("" : "+s"(tc[b]) :: "memory"); }
        }
        if (b + 1 < NCH) tc[b + 1] = tap_chunk(b + 1);
        if constexpr (!NOLDS) {
            if (b + DEPTH < NB) { LDS_BLOCK(b + DEPTH, buf[(b + DEPTH) % NBUF]); }
        } else {
            if (b + DEPTH < NB) {
#pragma unroll
                for (int i = 0; i < TC / 2; i++) buf[(b + DEPTH) % NBUF][i] = buf[b % NBUF][i];
            }
        }
        if constexpr (PIN) asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < TC / 2; i++) {
            const float4 v2 = buf[b % NBUF][i];
            const float2 v[2] = {make_float2(v2.x, v2.y), make_float2(v2.z, v2.w)};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int ss = TC * b + 2 * i + e;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = ss - r * D;
                    if (j >= 0 && j < P) {
                        const float h = tc[j / TC][j % TC];
                        acc[r][j & 3].x = acc[r][j & 3].x + h * v[e].x;
                        acc[r][j & 3].y = acc[r][j & 3].y + h * v[e].y;
                    }
                }
            }
        }
    }
#undef LDS_BLOCK
}

template <int D, int P, int R, class T, int VAR>
__device__ __forceinline__ void compute_tile_v(const float2* __restrict__ lds, const float* __restrict__ taps, float* __restrict__ out, int out0)
{
    const float2* win = lds + T::lds_idx(threadIdx.x * T::CHUNK);
    float2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[r][k] = make_float2(0.0f, 0.0f);
    if constexpr (VAR == 0) mac_window<D, P, R, T, 8, false>(win, taps, acc, 0);
    else if constexpr (VAR == 1) mac_window2<D, P, R, T, 1, true, false, false>(win, taps, acc);
    else if constexpr (VAR == 2) mac_window2<D, P, R, T, 2, true, false, false>(win, taps, acc);
    else if constexpr (VAR == 3) mac_window2<D, P, R, T, 1, false, false, false>(win, taps, acc);
    else if constexpr (VAR == 4) mac_window2<D, P, R, T, 1, true, true, false>(win, taps, acc);    // probe: no LDS stream
    else if constexpr (VAR == 5) mac_window2<D, P, R, T, 1, true, false, true>(win, taps, acc);    // probe: no scalar loads
    else if constexpr (VAR == 6) mac_window2<D, P, R, T, 1, true, true, true>(win, taps, acc);     // probe: neither
    else mac_window2<D, P, R, T, 3, true, false, false>(win, taps, acc);
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

// MAC phase alone (stale LDS), variant VAR
struct ClkProbe { unsigned long long cyc, rt; unsigned int n; unsigned int pad; };
__device__ __forceinline__ void probe_begin(unsigned long long& c0, unsigned long long& r0)
{
    c0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    r0 = wall_clock64();                     // s_memrealtime: constant 100 MHz
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

template <int D, int P, int R, int NT, int VAR>
__global__ void __launch_bounds__(NT) k_mac_only(int ntiles, const float* __restrict__ taps, float* __restrict__ out, ClkProbe* pr)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    unsigned long long c0, r0;
    probe_begin(c0, r0);
    compute_tile_v<D, P, R, T, VAR>(lds, taps, out, tile * T::OUTS);
    probe_end(pr, c0, r0);
}

// fills every workgroup's LDS with zeros (the next MAC-only kernel then multiplies constant data: low switching activity)
template <int NT>
__global__ void __launch_bounds__(NT) k_lds_fill(int nwords, unsigned int pattern_mul)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned int* w = reinterpret_cast<unsigned int*>(smem_raw);
    for (int i = threadIdx.x; i < nwords; i += NT) {
        unsigned int x = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        // random floats in [-1, 1): sign + exponent of 0.5..1 + random mantissa, or zero
        w[i] = pattern_mul ? ((x & 0x807fffffu) | 0x3f000000u) : 0u;
    }
    __syncthreads();
    if (w[threadIdx.x] == 0x12345678u) w[0] = 1;   // keep the stores
}

// the production structure (one tile per workgroup, register staging) around MAC variant VAR; NTL: non-temporal loads
template <int D, int P, int R, int NT, int VAR, bool NTL>
__global__ void __launch_bounds__(NT) k_dec_v(const float* __restrict__ in, int ntiles, const float* __restrict__ taps, float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    constexpr int NV = (T::SPAN + 1) / 2;
    constexpr int PER = (NV + NT - 1) / NT;
    const uint4* src = reinterpret_cast<const uint4*>(in + 2 * (int64_t)tile * T::OUTS * D);
    uint4 r[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        if (i + 1 < PER || v < NV) r[i] = ld16<NTL>(src + v);
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        if (v < NV) *reinterpret_cast<uint4*>(&lds[T::lds_idx(2 * v)]) = r[i];
    }
    __syncthreads();
    compute_tile_v<D, P, R, T, VAR>(lds, taps, out, tile * T::OUTS);
}

// the production structure with wave priorities: PRIO 1 = MAC phase at raised priority (waves that have their tile in LDS win the
// issue arbitration over waves still converting / storing theirs), PRIO 2 = the opposite (the load phase is favoured so that the
// next tiles' HBM requests go out as early as possible), PRIO 3 = odd workgroups sleep ~2 us before loading (co-resident workgroups
// start out of phase: one loads while the other multiplies)
template <int D, int P, int R, int NT, int VAR, int PRIO>
__global__ void __launch_bounds__(NT) k_dec_prio(const float* __restrict__ in, int ntiles, const float* __restrict__ taps, float* __restrict__ out)
{
    using T = Tile<D, P, R, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int tile = (b & ~63) + ((b & 7) << 3) + ((b >> 3) & 7);
    if (tile >= ntiles) return;
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 3 && ((b >> 8) & 1)) {
#pragma unroll 1
        for (int i = 0; i < 32; i++) __builtin_amdgcn_s_sleep(127);       // 32 x 127 x 64 clocks ~ a few hundred ns each
    }
    constexpr int NV = (T::SPAN + 1) / 2;
    constexpr int PER = (NV + NT - 1) / NT;
    const uint4* src = reinterpret_cast<const uint4*>(in + 2 * (int64_t)tile * T::OUTS * D);
    uint4 r[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        if (i + 1 < PER || v < NV) r[i] = ld16<false>(src + v);
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = threadIdx.x + i * NT;
        if (v < NV) *reinterpret_cast<uint4*>(&lds[T::lds_idx(2 * v)]) = r[i];
    }
    __syncthreads();
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    compute_tile_v<D, P, R, T, VAR>(lds, taps, out, tile * T::OUTS);
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
    auto run0 = [&](float* o) { hipLaunchKernelGGL(k0, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, o, P, 0, 0, 0); };
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

    const int ncu = prop.multiProcessorCount;
    (void)ncu;
    // streaming shapes: 8 uint4 in, 1 out per thread
    const int nchunks = (int)(n * 8 / 16 / (NT * 8));
    uint4* sin_ = reinterpret_cast<uint4*>(dx);
    uint4* sout = reinterpret_cast<uint4*>(dout);
#define SETLDS(k, bytes) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))

    ClkProbe* dpr;
    CK(hipMalloc(&dpr, sizeof(ClkProbe)));
    auto run_mac_lds = [&](auto kern, const char* nm, size_t lds_bytes) {
        SETLDS(kern, lds_bytes);
        CK(hipMemset(dpr, 0, sizeof(ClkProbe)));
        const double us = tm.us([&] { hipLaunchKernelGGL(kern, dim3(grid0), dim3(NT), lds_bytes, 0, ntiles, dt, dout, dpr); }, reps);
        ClkProbe h;
        CK(hipMemcpy(&h, dpr, sizeof h, hipMemcpyDeviceToHost));
        report(nm, us);
        if (h.rt) printf("    clock probe: %.0f MHz shader clock (avg over %u workgroups), %.2f us per workgroup\n", (double)h.cyc / (double)h.rt * 100.0, h.n,
                         (double)h.rt / h.n / 100.0);
    };
    auto run_mac = [&](auto kern, const char* nm) { run_mac_lds(kern, nm, T::LDS_BYTES); };
    auto run_dec = [&](auto kern, const char* nm, bool chk) {
        SETLDS(kern, T::LDS_BYTES);
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        report(nm, tm.us([&] { hipLaunchKernelGGL(kern, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, dx, ntiles, dt, dout); }, reps));
        if (chk) check(nm);
    };
    auto run_dec_lds = [&](auto kern, const char* nm, size_t lds_bytes, bool chk) {
        SETLDS(kern, lds_bytes);
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        report(nm, tm.us([&] { hipLaunchKernelGGL(kern, dim3(grid0), dim3(NT), lds_bytes, 0, dx, ntiles, dt, dout); }, reps));
        if (chk) check(nm);
    };
    for (int round = 0; round < rounds; round++) {
        printf("---- round %d\n", round);
        report("stream oneshot 8:1", tm.us([&] { hipLaunchKernelGGL((k_stream_oneshot<NT, false>), dim3(nchunks), dim3(NT), 0, 0, sin_, sout); }, reps));
        report("stream oneshot 8:1 nt", tm.us([&] { hipLaunchKernelGGL((k_stream_oneshot<NT, true>), dim3(nchunks), dim3(NT), 0, 0, sin_, sout); }, reps));
        CK(hipMemset(dout, 0xff, (size_t)nout * 8));
        report("V0 production (4 WG/CU, reg stage)", tm.us([&] { run0(dout); }, reps));
        check("V0");
        {
            auto kfull = k_decimate_c4<D, P, R, NT, false, 8, false, 4, 0, 0, true>;
            SETLDS(kfull, T::LDS_BYTES);
            CK(hipMemset(dout, 0xff, (size_t)nout * 8));
            report("V0 FULL instantiation (production)", tm.us([&] { hipLaunchKernelGGL(kfull, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, dout, P, 0, 0, ntiles); }, reps));
            check("V0 FULL");
        }
        run0(dout);   // leave real samples in LDS for the MAC-only kernels
        run_mac(k_mac_only<D, P, R, NT, 0>, "MAC only: mac_window (production)");
        run_mac_lds(k_mac_only<D, P, R, NT, 0>, "MAC only: production, 3 WG/CU", 50 * 1024);
        run_mac_lds(k_mac_only<D, P, R, NT, 0>, "MAC only: production, 2 WG/CU", 70 * 1024);
        run_mac_lds(k_mac_only<D, P, R, NT, 0>, "MAC only: production, 1 WG/CU", 100 * 1024);
        {
            auto kf = k_lds_fill<NT>;
            SETLDS(kf, 160 * 1024);
            hipLaunchKernelGGL(kf, dim3(ncu * 4), dim3(NT), 160 * 1024, 0, 160 * 256, 0u);
            run_mac(k_mac_only<D, P, R, NT, 0>, "MAC only: production, LDS all zero");
            hipLaunchKernelGGL(kf, dim3(ncu * 4), dim3(NT), 160 * 1024, 0, 160 * 256, 1u);
            run_mac(k_mac_only<D, P, R, NT, 0>, "MAC only: production, LDS random");
        }
        run_mac(k_mac_only<D, P, R, NT, 3>, "MAC only: mac_window2 unpinned d1");
        run_mac(k_mac_only<D, P, R, NT, 1>, "MAC only: pinned wait, depth 1");
        run_mac(k_mac_only<D, P, R, NT, 2>, "MAC only: pinned wait, depth 2");
        run_mac(k_mac_only<D, P, R, NT, 7>, "MAC only: pinned wait, depth 3");
        run_mac(k_mac_only<D, P, R, NT, 4>, "MAC only probe: no LDS stream");
        run_mac(k_mac_only<D, P, R, NT, 5>, "MAC only probe: no scalar loads");
        run_mac(k_mac_only<D, P, R, NT, 6>, "MAC only probe: neither");
        run_dec(k_dec_v<D, P, R, NT, 0, false>, "dec: mac_window", true);
        run_dec(k_dec_v<D, P, R, NT, 0, true>, "dec: mac_window, nt loads", true);
        run_dec(k_dec_v<D, P, R, NT, 1, false>, "dec: pinned d1", true);
        run_dec(k_dec_v<D, P, R, NT, 1, true>, "dec: pinned d1, nt loads", true);
        run_dec(k_dec_v<D, P, R, NT, 2, false>, "dec: pinned d2", true);
        run_dec(k_dec_v<D, P, R, NT, 2, true>, "dec: pinned d2, nt loads", true);
        run_dec(k_dec_v<D, P, R, NT, 7, true>, "dec: pinned d3, nt loads", true);
        run_dec(k_dec_v<D, P, R, NT, 3, true>, "dec: unpinned mac_window2, nt loads", true);
        // round 3: the combinations VERDICT r02 found missing
        run_dec(k_dec_v<D, P, R, NT, 3, false>, "dec: unpinned mac_window2, plain loads", true);
        run_dec_lds(k_dec_v<D, P, R, NT, 0, false>, "dec: mac_window, 3 WG/CU", 50 * 1024, true);
        run_dec_lds(k_dec_v<D, P, R, NT, 3, false>, "dec: unpinned mac_window2, 3 WG/CU", 50 * 1024, true);
        run_dec_lds(k_dec_v<D, P, R, NT, 0, false>, "dec: mac_window, 2 WG/CU", 70 * 1024, true);
        run_dec(k_dec_prio<D, P, R, NT, 0, 1>, "dec: mac_window, MAC phase prio 3", true);
        run_dec(k_dec_prio<D, P, R, NT, 0, 2>, "dec: mac_window, load phase prio 3", true);
        run_dec(k_dec_prio<D, P, R, NT, 0, 3>, "dec: mac_window, odd WGs start late", true);
        run_dec(k_dec_prio<D, P, R, NT, 3, 1>, "dec: unpinned mw2, MAC phase prio 3", true);
        // order check: the same three kernels back to back, twice (a row's place in the run decides the power state it meets)
        for (int ab = 0; ab < 2; ab++) {
            report("ABAB V0 production", tm.us([&] { run0(dout); }, reps));
            run_dec(k_dec_v<D, P, R, NT, 0, false>, "ABAB dec: mac_window", false);
            auto kfull = k_decimate_c4<D, P, R, NT, false, 8, false, 4, 0, 0, true>;
            report("ABAB V0 FULL", tm.us([&] { hipLaunchKernelGGL(kfull, dim3(grid0), dim3(NT), T::LDS_BYTES, 0, (const void*)dx, (int64_t)0, (int)nout, dt, dout, P, 0, 0, ntiles); }, reps));
        }
    }
    // copies with 4 x 16 B in flight per thread (the one-load-per-iteration copy is a weak ceiling), plain and nt
    {
        const size_t nv = (size_t)((rd_bytes + wr_bytes) / 2 / 16) / (NT * 4) * (NT * 4);
        uint4 *ca, *cb;
        CK(hipMalloc(&ca, nv * 16));
        CK(hipMalloc(&cb, nv * 16));
        CK(hipMemset(ca, 1, nv * 16));
        for (int rep2 = 0; rep2 < 2; rep2++) {
            double us = tm.us([&] { hipLaunchKernelGGL((k_copy4<NT, false>), dim3((unsigned)(nv / (NT * 4))), dim3(NT), 0, 0, ca, cb); }, reps);
            printf("copy, 4 x 16 B per thread, %zu MiB -> same: %8.1f us  total %5.3f TB/s\n", nv * 16 >> 20, us, 2.0 * nv * 16 / us / 1e6);
            us = tm.us([&] { hipLaunchKernelGGL((k_copy4<NT, true>), dim3((unsigned)(nv / (NT * 4))), dim3(NT), 0, 0, ca, cb); }, reps);
            printf("copy, 4 x 16 B per thread, nt loads + nt stores: %8.1f us  total %5.3f TB/s\n", us, 2.0 * nv * 16 / us / 1e6);
        }
        CK(hipFree(ca));
        CK(hipFree(cb));
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
