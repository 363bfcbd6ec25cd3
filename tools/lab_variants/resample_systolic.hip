// kernels_resample_systolic.hip -- the FM chain's 3/10 polyphase resampler (resampleAVXRR, c_sources/resample.c:70-87 ->
// avx_dotprod_R common.h:58-72) as a register-resident systolic walk (round 4), the design of kernels_systolic.hip applied to a real
// resampler: no barrier, no LDS in the multiply-add loop, whole waves streaming independently.
//
// Three polyphase groups of 64 (padded) taps, increments {4, 3, 3}: the outputs of polyphase CYCLE c start at inputs 10c + {0, 4, 7};
// output = tree8 of the lane partials a_l = sum_{j = l (mod 8)} c_g[j] * x[start + j], every a_l from +0 in increasing j, separate
// multiply and add.
//
// * A wave owns a STRIP of 2560 inputs; lane l keeps inputs 40l .. 40l + 39 (four cycles) in 40 VGPRs.  They arrive by coalesced
//   16-byte loads (a wave reads 10 KiB contiguously) and are transposed once through a wave-private LDS buffer (rows of 160 + 16 B:
//   conflict-free per 16 lanes) -- the kernel's only LDS traffic, 10 writes + 10 reads per lane and 12 outputs, no barrier.
// * The 96 partial sums of the GROUP q = the 12 outputs of cycles 4q .. 4q + 3 travel: they start in lane q and move one lane up per
//   STAGE of 40 inputs, as the DPP operand of the stage's first addition (v_add_f32_dpp wave_shr:1).  Output (a, g) of a group starts
//   s = 10a + {0, 4, 7}[g] inputs into its home lane; in stage t its tap j meets input s + j - 40t of lane q + t -- the same tap for every
//   lane, so taps are SGPR operands.  Three stages (s + 63 <= 100); outputs with s <= 16 are complete after two.
// * Packed pairs: taps (j, j + 1) with j even for the groups that start at an even input (g = 0, 1), j odd for g = 2 (s odd), so that
//   the two inputs are an aligned register pair; the partials are then paired (0,1)(2,3)(4,5)(6,7) resp. (1,2)(3,4)(5,6)(7,0), and
//   taps 0 and 63 of g = 2 are single operations.  A partial still adds its products in increasing tap order from +0: SAME BITS.
// * Lanes 0, 1 of a wave only warm the pipe up: 2560 inputs -> 62 groups = 744 outputs; strips advance by 2480 inputs.
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "lab.hpp"

namespace sdrhip {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kChunk = 40;                 // inputs per lane
constexpr int kCyc = 4;                    // cycles per lane
constexpr int kOutL = 12;                  // outputs per lane (group)
constexpr int kGroups = 62;                // complete groups per strip
constexpr int kStripCycles = kGroups * kCyc;        // 248
constexpr int kStripStep = kStripCycles * 10;       // 2480 inputs between strips
constexpr int kStripSpan = 64 * kChunk;             // 2560 inputs a strip reads
constexpr int kRow = 44;                   // dwords per LDS row (40 + 4 of padding)
constexpr int kWaveDw = 64 * kRow;
constexpr int kWaves = 4;
constexpr int kNL = 64;                    // taps per group row

__host__ __device__ constexpr int pre_of(int g) { return g == 0 ? 0 : g == 1 ? 4 : 7; }
__host__ __device__ constexpr int start_of(int o) { return 10 * (o / 3) + pre_of(o % 3); }      // output o = 3a + g of a group
__host__ __device__ constexpr bool odd_group(int o) { return (o % 3) == 2; }
// taps of output o that fall into stage t: [lo, hi)
__host__ __device__ constexpr int stage_lo(int o, int t) { return 40 * t - start_of(o) > 0 ? 40 * t - start_of(o) : 0; }
__host__ __device__ constexpr int stage_hi(int o, int t) { return 40 * (t + 1) - start_of(o) < kNL ? 40 * (t + 1) - start_of(o) : kNL; }
__host__ __device__ constexpr int last_stage(int o) { return (start_of(o) + kNL - 1) / 40; }
// pair p of output o holds partials (2p, 2p + 1) [even groups] or (2p + 1, (2p + 2) & 7) [odd group]
__host__ __device__ constexpr int pair_of_partial(int o, int l) { return odd_group(o) ? ((l + 7) & 7) / 2 : l / 2; }
__host__ __device__ constexpr int half_of_partial(int o, int l) { return odd_group(o) ? ((l + 7) & 7) % 2 : l % 2; }
// does partial l of output o receive an addition in stage t?
__host__ __device__ constexpr bool partial_touched(int o, int l, int t)
{
    for (int j = stage_lo(o, t); j < stage_hi(o, t); j++)
        if ((j & 7) == l) return true;
    return false;
}
// the first tap of partial l of output o in stage t (-1: none)
__host__ __device__ constexpr int first_tap(int o, int l, int t)
{
    for (int j = stage_lo(o, t); j < stage_hi(o, t); j++)
        if ((j & 7) == l) return j;
    return -1;
}

// has partial l of output o received an addition before stage t?  (Outputs that start late in their home lane -- s >= 33 -- have
// fewer than 8 taps in stage 0: the other partials start from +0 in stage 1, with nothing to move up.)
__host__ __device__ constexpr bool started_before(int o, int l, int t) { return l < stage_lo(o, t); }

__device__ __forceinline__ float dpp_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

// one addition into partial l of output o at tap j of stage t: the first addition of a stage carries the move from the lane below
template <int O, int L, int J, int T>
__device__ __forceinline__ void add1(f2 (&A)[kOutL][4], float prod)
{
    constexpr int p = pair_of_partial(O, L), h = half_of_partial(O, L);
    float cur = h == 0 ? A[O][p].x : A[O][p].y;
    float nv;
    if constexpr (J == first_tap(O, L, T) && !started_before(O, L, T)) nv = 0.0f + prod;
    else if constexpr (J == first_tap(O, L, T)) nv = dpp_shr1(cur) + prod;
    else nv = cur + prod;
    if (h == 0) A[O][p].x = nv; else A[O][p].y = nv;
}

// taps (J, J + 1) of output O in stage T as one packed pair (both in the stage's range, inputs an aligned register pair).  Either half
// may be the first addition of its partial in the stage (it then starts from +0 resp. carries the move from the lane below); when
// only one of them is -- the pair (7, 0) of the odd group, whose partial 0 was started by the single tap 0 -- the two halves are added
// separately.
template <int O, int J, int T>
__device__ __forceinline__ void add2(f2 (&A)[kOutL][4], const f2 x, const f2 c)
{
    constexpr int l0 = J & 7, l1 = (J + 1) & 7, p = pair_of_partial(O, l0);
    static_assert(half_of_partial(O, l0) == 0 && pair_of_partial(O, l1) == p && half_of_partial(O, l1) == 1, "not a pair");
    constexpr bool f0 = J == first_tap(O, l0, T), f1 = J + 1 == first_tap(O, l1, T);
    constexpr bool n0 = f0 && !started_before(O, l0, T), n1 = f1 && !started_before(O, l1, T);      // the partial's very first addition
    const f2 prod = x * c;
    if constexpr (!f0 && !f1) {
        A[O][p] = A[O][p] + prod;
    } else if constexpr (n0 && n1) {
        A[O][p] = f2{0.0f, 0.0f} + prod;
    } else {
        const float ax = n0 ? 0.0f : (f0 ? dpp_shr1(A[O][p].x) : A[O][p].x);
        const float ay = n1 ? 0.0f : (f1 ? dpp_shr1(A[O][p].y) : A[O][p].y);
        A[O][p] = f2{ax + prod.x, ay + prod.y};
    }
}

template <bool WHOLE>
__device__ __forceinline__ void resample_strip(const float* __restrict__ in, int64_t pos, int strip, int ncycles, int64_t avail_total,
                                               const float* __restrict__ groups, int row_stride, float* __restrict__ out,
                                               float* __restrict__ wbuf, int lane)
{
    const int64_t base = pos + (int64_t)kStripStep * strip;
    const int64_t avail = WHOLE ? kStripSpan : avail_total - (int64_t)kStripStep * strip;
    // coalesced 16-byte loads, all in flight; then the transpose through the wave's LDS rows
    f2 X[kChunk / 2];
    {
        float4 v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int e = 4 * (64 * k + lane);                 // first of the four inputs of this vector, relative to the strip
            const float* p = in + base + e;
            if (WHOLE || e + 4 <= avail) {
                v[k] = *reinterpret_cast<const float4*>(p);
            } else {
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e + 0 < avail) v[k].x = p[0];
                if (e + 1 < avail) v[k].y = p[1];
                if (e + 2 < avail) v[k].z = p[2];
            }
        }
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int e = 4 * (64 * k + lane);
            *reinterpret_cast<float4*>(wbuf + kRow * (e / kChunk) + (e % kChunk)) = v[k];    // 40 = 10 vectors: a vector never straddles rows
        }
#pragma unroll
        for (int q = 0; q < 10; q++) {
            const float4 t4 = *reinterpret_cast<const float4*>(wbuf + kRow * lane + 4 * q);
            X[2 * q] = f2{t4.x, t4.y};
            X[2 * q + 1] = f2{t4.z, t4.w};
        }
    }

    f2 A[kOutL][4];
    static_for(std::make_integer_sequence<int, 3>{}, [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        // outputs still under way whose partials receive no addition in this stage move up on their own
        static_for(std::make_integer_sequence<int, kOutL>{}, [&](auto oc) {
            constexpr int O = decltype(oc)::value;
            if constexpr (T > 0 && T <= last_stage(O)) {
                static_for(std::make_integer_sequence<int, 8>{}, [&](auto lc) {
                    constexpr int L = decltype(lc)::value;
                    if constexpr (!partial_touched(O, L, T) && started_before(O, L, T)) {
                        constexpr int p = pair_of_partial(O, L), h = half_of_partial(O, L);
                        if (h == 0) A[O][p].x = dpp_shr1(A[O][p].x); else A[O][p].y = dpp_shr1(A[O][p].y);
                    }
                });
            }
        });
        static_for(std::make_integer_sequence<int, 3>{}, [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            const float* c = groups + G * row_stride;
            static_for(std::make_integer_sequence<int, kNL>{}, [&](auto jc) {
                constexpr int J = decltype(jc)::value;
                static_for(std::make_integer_sequence<int, kCyc>{}, [&](auto ac) {
                    constexpr int O = 3 * decltype(ac)::value + G;
                    constexpr int s = start_of(O), lo = stage_lo(O, T), hi = stage_hi(O, T);
                    if constexpr (J >= lo && J < hi) {
                        constexpr int u = s + J - 40 * T;                    // the lane's input this tap meets
                        constexpr bool pair_start = (u % 2 == 0) && J + 1 < hi;
                        constexpr bool pair_second = (u % 2 == 1) && J - 1 >= lo;
                        if constexpr (pair_start) {
                            add2<O, J, T>(A, X[u / 2], f2{c[J], c[J + 1]});
                        } else if constexpr (!pair_second) {
                            const float xv = (u % 2 == 0) ? X[u / 2].x : X[u / 2].y;
                            add1<O, (J & 7), J, T>(A, c[J] * xv);
                        }
                    }
                });
            });
        });
        // outputs complete after this stage (and not the last one): fold here, the result moves up below
    });

    // fold: tree8 = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)).  Outputs that were complete after stage 1 sit one lane below
    // the others (their sums did not move in stage 2): their folded result moves up by one more DPP.
    float res[kOutL];
    static_for(std::make_integer_sequence<int, kOutL>{}, [&](auto oc) {
        constexpr int O = decltype(oc)::value;
        float a[8];
#pragma unroll
        for (int l = 0; l < 8; l++) a[l] = half_of_partial(O, l) == 0 ? A[O][pair_of_partial(O, l)].x : A[O][pair_of_partial(O, l)].y;
        const float r = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        if constexpr (last_stage(O) < 2) res[O] = dpp_shr1(r);
        else res[O] = r;
    });
    if (lane >= 2) {
        const int cyc = kStripCycles * strip + kCyc * (lane - 2);            // first cycle of the lane's group
        float* dst = out + 3 * (int64_t)cyc;
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        if (WHOLE || cyc + kCyc <= ncycles) {
#pragma unroll
            for (int q = 0; q < 3; q++) *reinterpret_cast<f4u*>(dst + 4 * q) = f4u{res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]};
        } else {
#pragma unroll
            for (int o = 0; o < kOutL; o++)
                if (cyc + o / 3 < ncycles) dst[o] = res[o];
        }
    }
}

__global__ void __launch_bounds__(64 * kWaves, 3) k_resample3_systolic(const float* __restrict__ in, int64_t pos, int ncycles, int64_t avail_total,
                                                                        const float* __restrict__ groups, int row_stride, float* __restrict__ out,
                                                                        int nwhole, int nstrips)
{
    __shared__ __attribute__((aligned(16))) float tbuf[kWaves * kWaveDw];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * kWaves + wave;
    if (strip >= nstrips) return;
    float* wbuf = tbuf + kWaveDw * wave;
    if (strip < nwhole) resample_strip<true>(in, pos, strip, ncycles, avail_total, groups, row_stride, out, wbuf, lane);
    else resample_strip<false>(in, pos, strip, ncycles, avail_total, groups, row_stride, out, wbuf, lane);
}

std::atomic<long long> g_launches{0};
// OFF by default: measured on MI355X inside the chain (2^26 inputs per launch, alternating A/B in one process, tools/k2k3_fusion_ab.py)
// the resample stage takes 0.0921-0.0931 ms with this kernel against 0.0908-0.0911 with the LDS-tiled k_resample3_fast -- both sit at
// ~4.3 TB/s of algorithmic traffic with the VALU 60 % busy; neither the barrier nor the LDS window is what the tile kernel waits for.
// Kept as an option (same bits, under test): SDRHIP_RESAMP_SYSTOLIC=1 / sdrhip_debug_set_resample_systolic(1).
std::atomic<int> g_on{getenv("SDRHIP_RESAMP_SYSTOLIC") ? atoi(getenv("SDRHIP_RESAMP_SYSTOLIC")) : 0};

}  // namespace

void set_resample_systolic(int on) { g_on.store(on); }

long long resample_systolic_launch_count() { return g_launches.load(); }

// strips of a launch of `ncycles` cycles whose inputs exist up to avail_total: [0, nwhole) are whole (host arithmetic, CPU-testable)
void resample_systolic_plan(int ncycles, int64_t avail_total, int* nstrips, int* nwhole)
{
    *nstrips = (ncycles + kStripCycles - 1) / kStripCycles;
    int w = ncycles / kStripCycles;
    while (w > 0 && (int64_t)kStripStep * (w - 1) + kStripSpan > avail_total) w--;
    *nwhole = w;
}

// The whole cycles of a 3/10 launch in the AVX order: cycle c (outputs 3c .. 3c + 2) starts at d_in[pos + 10c].  false = not taken.
bool launch_resample3_systolic(hipStream_t s, const float* d_in, int64_t pos, int ncycles, int64_t avail_total, const float* d_groups,
                               int row_stride, float* d_out)
{
    if (!g_on.load(std::memory_order_relaxed) || ncycles < 64 * kStripCycles) return false;
    if ((reinterpret_cast<uintptr_t>(d_in + pos) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 3) != 0) return false;
    int nstrips, nwhole;
    resample_systolic_plan(ncycles, avail_total, &nstrips, &nwhole);
    hipLaunchKernelGGL(k_resample3_systolic, dim3((nstrips + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, s, d_in, pos, ncycles, avail_total, d_groups,
                       row_stride, d_out, nwhole, nstrips);
    g_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

}  // namespace sdrhip
