// kernels_decimate_real.hip -- real FIR decimators by 2 / 4 / 8 / 16, AVX / SSE lane order, any filter length
// (decimateAVXRR / decimateSSERR, c_sources/decimate.c:36-66 -> avx_dotprod_R / sse_dotprod_R, common.h:34-72).
//
// The thread-per-cycle design of kernels_resample_cycle.hip with "cycle" = R = 16 / D consecutive outputs: thread t owns
// outputs R t .. R t + R - 1, whose windows start D floats apart inside one register window; every tap is wave-uniform and
// arrives by scalar loads (16 taps one step ahead), a MAC costs its multiply and its add (the lane-split kernel pays one
// ds_read_b32 per MAC on top).  The filter is walked 16 taps per rolled step; the window slides through 16-byte LDS reads.
// Thread windows start 16 floats apart, which b128 reads would hit four ways into the same banks: the tile is stored with four
// floats of padding after every 16 (thread stride 20 dwords: 16 consecutive lanes cover all 64 banks once), and because a
// step is as long as a thread's chunk the padding sits at the same place in every step.  Tiles are staged with 16-byte
// global loads (4-byte aligned: the stream API hands over any float offset).  Seams: every output in the SIMD order first,
// then the generic sequential fix-up of crossfix.hpp.
#include <atomic>
#include <type_traits>

#include "crossfix.hpp"
#include "kernels.hpp"

namespace sdrhip {

namespace {

constexpr int DR_NT = 256;
constexpr int DR_NV = 8;                       // 16-byte vectors a thread stages at most

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

// LDS position of float k of a window / of the tile: four floats of padding after every 16
__host__ __device__ constexpr int dr_idx(int k) { return k + 4 * (k / 16); }

// stage span4 16-byte vectors from src (4-byte aligned; `avail` floats exist) into the padded tile: all of a thread's global loads
// in flight before the first wait
__device__ __forceinline__ void dr_stage_tile(const float* __restrict__ src, int64_t avail, int span4, int tid, float* __restrict__ dr_lds)
{
    float4 val[DR_NV];
#pragma unroll
    for (int i = 0; i < DR_NV; i++) {
        const int v = tid + i * DR_NT;
        float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (v < span4) {
            const int64_t s = 4 * (int64_t)v;
            if (s + 3 < avail) {
                const f4u u = *reinterpret_cast<const f4u*>(src + s);
                q = make_float4(u.x, u.y, u.z, u.w);
            } else {
                if (s + 0 < avail) q.x = src[s + 0];
                if (s + 1 < avail) q.y = src[s + 1];
                if (s + 2 < avail) q.z = src[s + 2];
            }
        }
        val[i] = q;
    }
#pragma unroll
    for (int i = 0; i < DR_NV; i++) {
        const int v = tid + i * DR_NT;
        if (v < span4) *reinterpret_cast<float4*>(&dr_lds[4 * v + 4 * (v / 4)]) = val[i];
    }
}

template <int D, int L, bool RINGED>
__global__ void __launch_bounds__(DR_NT) k_decimate_real16(const float* __restrict__ in, int64_t pos0, int count, int64_t avail_total,
                                                            const float* __restrict__ taps, int nloop, float gain, int apply_gain,
                                                            float* __restrict__ out)
{
    static_assert(D == 2 || D == 4 || D == 8 || D == 16, "a thread's outputs span 16 inputs");
    static_assert(L == 8 || L == 4, "AVX or SSE lane count");
    constexpr int R = 16 / D, PM = (R - 1) * D;
    constexpr int A = (PM + 3) / 4 * 4 + 16;                          // floats of the window a step reads (16-byte granules)
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * DR_NT;                                // first thread-chunk of the tile
    const int span = (DR_NT - 1) * 16 + A + nloop + 16;               // floats the tile reads (the window runs one step ahead)
    const int span4 = (span + 3) / 4;
    dr_stage_tile(in + pos0 + (int64_t)t0 * 16, avail_total - (int64_t)t0 * 16, span4, tid, dr_lds);
    __syncthreads();
    const int64_t m0 = (int64_t)(t0 + tid) * R;                       // first output of the thread, relative to the launch
    if (m0 >= count) return;
    const float* wp = dr_lds + tid * 20;
    float acc[R][L];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int l = 0; l < L; l++) acc[r][l] = 0.0f;
    // the register window is a ring of RING floats walked 16 per step: U = RING / 16 steps are unrolled in the loop body, so
    // every step addresses fixed registers and nothing is ever moved (a sliding window would cost A moves per 16 R MACs)
    constexpr int RING = RINGED ? (A + 16 + 15) / 16 * 16 : A + 16, U = RINGED ? RING / 16 : 1;
    static_assert(!RINGED || U == 2 || U == 3, "ring of 32 or 48 floats");
    float w[RING];
#pragma unroll
    for (int q = 0; q < A / 4; q++) {
        const float4 v = *reinterpret_cast<const float4*>(wp + dr_idx(4 * q));
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    if (nloop >= 16) {                                                 // (a filter shorter than one step is all tail code)
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = taps[i];
    }
    // one step = 16 taps of the thread's R outputs; PH = where the step's window starts in the ring (in steps)
    auto step = [&](auto ph_tag, int j0) {
        constexpr int B0 = 16 * decltype(ph_tag)::value;
        asm volatile("" ::: "memory");                                 // the steps' loads stay in their own steps (registers)
        const float* wn = wp + 20 * (j0 >> 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {                                  // the next step's 16 values, in flight during this step
            const float4 v = *reinterpret_cast<const float4*>(wn + dr_idx(A + 4 * q));
            w[(B0 + A + 4 * q) % RING] = v.x; w[(B0 + A + 4 * q + 1) % RING] = v.y;
            w[(B0 + A + 4 * q + 2) % RING] = v.z; w[(B0 + A + 4 * q + 3) % RING] = v.w;
        }
        float cn[16];
        const float* tn = taps + (j0 + 32 <= nloop ? j0 + 16 : j0);   // (the last prefetch re-reads the step's own taps)
#pragma unroll
        for (int i = 0; i < 16; i++) cn[i] = tn[i];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[r][i % L] = acc[r][i % L] + c[i] * w[(B0 + r * D + i) % RING];       // tap j0 + i: lane i % L
            // one output's products at a time (left alone the scheduler multiplies for every output first: 100+ temporaries)
            if constexpr (L == 8)
                asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]), "+v"(acc[r][4]), "+v"(acc[r][5]),
                             "+v"(acc[r][6]), "+v"(acc[r][7]));
            else asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = cn[i];
        if constexpr (!RINGED) {                                       // sliding window: everything moves down one step
#pragma unroll
            for (int k = 0; k < A; k++) w[k] = w[k + 16];
        }
    };
    // the last 4 / 8 / 12 taps (the lane count divides the padded filter length, 16 need not)
    auto tail = [&](auto ph_tag, int j0) {
        constexpr int B0 = 16 * decltype(ph_tag)::value;
#pragma unroll
        for (int q = 0; q < 16; q += L) {
            if (j0 + q < nloop) {
                const float* tq = taps + j0 + q;
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < L; i++) acc[r][i] = acc[r][i] + tq[i] * w[(B0 + r * D + q + i) % RING];
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    int j0 = 0;
    if constexpr (RINGED) {
#pragma unroll 1
        for (; j0 + 16 * U <= nloop; j0 += 16 * U) {
            step(P0{}, j0);
            step(P1{}, j0 + 16);
            if constexpr (U == 3) step(P2{}, j0 + 32);
        }
        if (j0 + 16 <= nloop) {
            step(P0{}, j0);
            j0 += 16;
            if (U == 3 && j0 + 16 <= nloop) {
                step(P1{}, j0);
                j0 += 16;
                if constexpr (U == 3) tail(P2{}, j0);
            } else tail(P1{}, j0);
        } else tail(P0{}, j0);
    } else {
#pragma unroll 1
        for (; j0 + 16 <= nloop; j0 += 16) step(P0{}, j0);
        tail(P0{}, j0);
    }
    float res[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if constexpr (L == 8) res[r] = ((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3])) + ((acc[r][4] + acc[r][5]) + (acc[r][6] + acc[r][7]));
        else res[r] = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        if (apply_gain) res[r] = res[r] * gain;
    }
    float* o = out + m0;
#pragma unroll
    for (int r = 0; r < R; r++)
        if (m0 + r < count) o[r] = res[r];
}


// The symmetric form (decimateAVXSymmetricRR / decimateSSESymmetricRR, decimate.c:53-83 -> avx_sym_dotprod_R, common.h:58-72):
// half-tap j multiplies x[j] + x[2N-1-j] -- the pair is added first, then multiplied, then accumulated in lane j % L.  A second
// register window slides DOWN from the far end of the thread's window; nhalf a multiple of 8, so that it starts on a 16-float
// granule and the tile's padding sits at fixed offsets for it as well.
template <int D, int L>
__global__ void __launch_bounds__(DR_NT) k_decimate_real16_sym(const float* __restrict__ in, int64_t pos0, int count, int64_t avail_total,
                                                                const float* __restrict__ taps, int nhalf, float gain, int apply_gain,
                                                                float* __restrict__ out)
{
    static_assert(D == 2 || D == 4 || D == 8 || D == 16, "a thread's outputs span 16 inputs");
    static_assert(L == 8 || L == 4, "AVX or SSE lane count");
    constexpr int R = 16 / D, PM = (R - 1) * D;
    constexpr int A = (PM + 3) / 4 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * DR_NT;
    const int span = (DR_NT - 1) * 16 + A + 2 * nhalf + 16;
    const int span4 = (span + 3) / 4;
    dr_stage_tile(in + pos0 + (int64_t)t0 * 16, avail_total - (int64_t)t0 * 16, span4, tid, dr_lds);
    __syncthreads();
    const int64_t m0 = (int64_t)(t0 + tid) * R;
    if (m0 >= count) return;
    const float* wp = dr_lds + tid * 20;
    float acc[R][L];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int l = 0; l < L; l++) acc[r][l] = 0.0f;
    // forward window: w[k] = x[j0 + k]; backward window: wb[16 + k] = x[kb + k], kb = 2 nhalf - 16 - j0 (wb[0 .. 16): the next
    // step's, sixteen floats further down)
    float w[A + 16], wb[A + 16];
    const float* wbp = wp + 20 * ((2 * nhalf - 16) >> 4);             // (2 nhalf - 16 is a multiple of 16)
#pragma unroll
    for (int q = 0; q < A / 4; q++) {
        const float4 v = *reinterpret_cast<const float4*>(wp + dr_idx(4 * q));
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        const float4 u = *reinterpret_cast<const float4*>(wbp + dr_idx(4 * q));
        wb[16 + 4 * q] = u.x; wb[16 + 4 * q + 1] = u.y; wb[16 + 4 * q + 2] = u.z; wb[16 + 4 * q + 3] = u.w;
    }
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    if (nhalf >= 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = taps[i];
    }
    int j0 = 0;
#pragma unroll 1
    for (; j0 + 16 <= nhalf; j0 += 16) {
        const float* wn = wp + 20 * (j0 >> 4);
        const float* wbn = wbp - 20 * ((j0 >> 4) + 1);               // sixteen floats below the backward window (>= the thread's
                                                                    // own window start: j0 + 16 <= nhalf)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(wn + dr_idx(A + 4 * q));
            w[A + 4 * q] = v.x; w[A + 4 * q + 1] = v.y; w[A + 4 * q + 2] = v.z; w[A + 4 * q + 3] = v.w;
            const float4 u = *reinterpret_cast<const float4*>(wbn + 4 * q);
            wb[4 * q] = u.x; wb[4 * q + 1] = u.y; wb[4 * q + 2] = u.z; wb[4 * q + 3] = u.w;
        }
        float cn[16];
        const float* tn = taps + (j0 + 32 <= nhalf ? j0 + 16 : j0);
#pragma unroll
        for (int i = 0; i < 16; i++) cn[i] = tn[i];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[r][i % L] = acc[r][i % L] + (w[r * D + i] + wb[16 + r * D + 15 - i]) * c[i];
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = cn[i];
#pragma unroll
        for (int k = 0; k < A; k++) w[k] = w[k + 16];
#pragma unroll
        for (int k = A + 15; k >= 16; k--) wb[k] = wb[k - 16];
    }
    // the last 8 half-taps (nhalf is a multiple of 8, 16 need not divide it)
#pragma unroll
    for (int q = 0; q < 16; q += L) {
        if (j0 + q < nhalf) {
            const float* tq = taps + j0 + q;
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int i = 0; i < L; i++) acc[r][i] = acc[r][i] + (w[r * D + q + i] + wb[16 + r * D + 15 - (q + i)]) * tq[i];
        }
    }
    float* o = out + m0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float res;
        if constexpr (L == 8) res = ((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3])) + ((acc[r][4] + acc[r][5]) + (acc[r][6] + acc[r][7]));
        else res = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        if (apply_gain) res = res * gain;
        if (m0 + r < count) o[r] = res;
    }
}

std::atomic<long long> g_decreal_launches{0};

template <int D, int L, bool RINGED>
bool launch_dr(hipStream_t s, const Geom& g, const float* d_taps, int nk, const float* d_in, float* d_out, float gain, bool apply_gain)
{
    constexpr int R = 16 / D, PM = (R - 1) * D, A = (PM + 3) / 4 * 4 + 16;
    const int span = (DR_NT - 1) * 16 + A + nk + 16;
    const int span4 = (span + 3) / 4;
    if (span4 > DR_NV * DR_NT) return false;
    const size_t lds_bytes = (size_t)(4 * span4 + 4 * (span4 / 4) + 8) * sizeof(float);
    if (lds_bytes > 60 * 1024) return false;
    const int64_t pos0 = g.k_begin * g.D - g.in_base;                                  // window of the launch's first output
    const int64_t avail_total = (int64_t)(g.count - 1) * D + nk;
    const int threads = (g.count + R - 1) / R;
    hipLaunchKernelGGL((k_decimate_real16<D, L, RINGED>), dim3((threads + DR_NT - 1) / DR_NT), dim3(DR_NT), lds_bytes, s, d_in, pos0, g.count, avail_total,
                       d_taps, nk, gain, apply_gain ? 1 : 0, d_out);
    g_decreal_launches++;
    return true;
}


template <int D, int L>
bool launch_dr_sym(hipStream_t s, const Geom& g, const float* d_half, int nhalf, const float* d_in, float* d_out, float gain, bool apply_gain)
{
    constexpr int R = 16 / D, PM = (R - 1) * D, A = (PM + 3) / 4 * 4 + 16;
    const int span = (DR_NT - 1) * 16 + A + 2 * nhalf + 16;
    const int span4 = (span + 3) / 4;
    if (span4 > DR_NV * DR_NT) return false;
    const size_t lds_bytes = (size_t)(4 * span4 + 4 * (span4 / 4) + 8) * sizeof(float);
    if (lds_bytes > 60 * 1024) return false;
    const int64_t pos0 = g.k_begin * g.D - g.in_base;
    const int64_t avail_total = (int64_t)(g.count - 1) * D + 2 * nhalf;
    const int threads = (g.count + R - 1) / R;
    hipLaunchKernelGGL((k_decimate_real16_sym<D, L>), dim3((threads + DR_NT - 1) / DR_NT), dim3(DR_NT), lds_bytes, s, d_in, pos0, g.count, avail_total,
                       d_half, nhalf, gain, apply_gain ? 1 : 0, d_out);
    g_decreal_launches++;
    return true;
}

}  // namespace

long long decimate_real16_launch_count() { return g_decreal_launches.load(); }

bool launch_decimate_real16_fast(hipStream_t s, const Geom& g, int lanes, const float* d_taps, int nk, const float* d_cross_taps, const float* d_in,
                                 float* d_out, float gain, bool apply_gain, int ncross, bool sym)
{
    if (ncross <= 0) ncross = g.Lp;              // taps the sequential (Cross) outputs walk: a resampler's are the unpadded ones
    if (g.I != 1 || g.seamBI < 0 || g.count < 4096) return false;
    if (!(lanes == 8 || lanes == 4) || nk < 8 || nk > 2048) return false;
    if (sym ? (nk % 8 != 0 || 2 * nk != g.Lp) : (nk % lanes != 0 || nk != g.Lp)) return false;     // sym: nk = half-taps
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    bool took = false;
#define DR(DV, RDEF)                                                                                               \
    if (g.D == DV) {                                                                                               \
        const bool ring = RDEF;                                                                                     \
        if (lanes == 8) took = ring ? launch_dr<DV, 8, true>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain)       \
                                    : launch_dr<DV, 8, false>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain);     \
        else took = ring ? launch_dr<DV, 4, true>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain)                  \
                         : launch_dr<DV, 4, false>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain);                \
    }
#define DRS(DV) if (g.D == DV) took = lanes == 8 ? launch_dr_sym<DV, 8>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain) \
                                                : launch_dr_sym<DV, 4>(s, g, d_taps, nk, d_in, d_out, gain, apply_gain)
    if (sym) { DRS(2); DRS(4); DRS(8); DRS(16); }
    else { DR(2, false); DR(4, false); DR(8, false); DR(16, true); }
#undef DRS
#undef DR
    if (!took) return false;
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            // LDS-staged fix-up (the resamplers' kernel with interpolation 1: one group of 32 / 64 lanes per seam, the straddlers'
            // union of inputs and the taps in LDS) where a seam's straddlers fit, the generic one (global reads) beyond
            const int64_t last_m = g.k_begin + g.count - 1;
            const int64_t in_avail = last_m * g.D - g.in_base + g.Lp;                                // inputs the caller guarantees
            const int ga = apply_gain ? 1 : 0;
            auto uni = [&](int PER) { return g.Lp + PER * g.D + 4; };
            if (per <= 16 && uni(16) <= 416)
                hipLaunchKernelGGL((k_resample_real_crossfix<16, 416, 32>), dim3((nseams + 7) / 8), dim3(256), 0, s, g, d_cross_taps, ncross, d_in, d_out,
                                   first, nseams, in_avail, gain, ga);
            else if (per <= 32 && uni(32) <= 416)
                hipLaunchKernelGGL((k_resample_real_crossfix<32, 416, 32>), dim3((nseams + 7) / 8), dim3(256), 0, s, g, d_cross_taps, ncross, d_in, d_out,
                                   first, nseams, in_avail, gain, ga);
            else if (per <= 64 && uni(64) <= 1152)
                hipLaunchKernelGGL((k_resample_real_crossfix<64, 1152, 64>), dim3((nseams + 3) / 4), dim3(256), 0, s, g, d_cross_taps, ncross, d_in, d_out,
                                   first, nseams, in_avail, gain, ga);
            else if (ncross != g.Lp) {
                // a resampler with interpolation 1 whose seams do not fit the LDS kernel: its own generic fix-up (unpadded taps)
                const int64_t total = (int64_t)nseams * per;
                hipLaunchKernelGGL(k_resample_crossfix<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_cross_taps, ncross, d_in, d_out,
                                   first, nseams, per);
            } else {
                const int64_t total = (int64_t)nseams * per;
                hipLaunchKernelGGL(k_fir_real_crossfix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams,
                                   per, gain, ga);
            }
        }
    }
    return true;
}

}  // namespace sdrhip
