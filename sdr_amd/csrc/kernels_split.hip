// kernels_split.hip -- one LDS-tiled kernel for every FIR family that has no hand-specialised kernel
// (real decimators, complex decimators with any factor / tap count, rational resamplers with any
// I/D, real or complex, the SSE orders, the symmetric variants).
//
// The reference's SIMD kernels keep M partial sums per output -- partial l accumulates taps j = l, l+M, l+2M, ... in
// increasing j from +0 -- and fold them with a fixed tree (common.h:18-29,58-72,82-90,108-155).  Here those M partial
// sums are M LANES of a wavefront: lane (o, l) walks  acc = acc + h[M*i + l] * x[start_o + M*i + l]  for i = 0, 1, ..,
// so a wave works on 64/M outputs at a time and
//   * its LDS reads are runs of M consecutive samples per output: conflict-free for every decimation factor whose
//     runs do not wrap the 32 banks (all D <= 8; two-way at worst for the other usual ones),
//   * each lane's taps h[l], h[M+l], ... sit in registers for as long as the wave stays on one tap row,
//   * the tree is a butterfly over the M lanes (xor 1, 2, 4 for the RR / RC kernels; xor M/2 first for the "RC2" order,
//     whose partials are folded q_l = p_l + p_{l+M/2} before the tree, common.h:142-155).  Lane 0 always adds
//     (its own) + (the other's), so operand order is the reference's as well.
// A resampler is NC = numGroups interleaved decimators: outputs i = n*NC + c share tap row (group0 + c) % NC and start
// at pos0 + n*period + pre[c] (Filter.hs:613-641, resample.c:70-87), so a wave stays on one row for a whole run of n.
// One workgroup stages the input span of NN cycles once, computes its NN*NC outputs, and stores them coalesced from LDS.
// Seam straddlers ("Cross" outputs) are rewritten afterwards by the generic kernels of crossfix.hpp.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "crossfix.hpp"
#include "kernels.hpp"

// Tile shape (measured alternatives: LABNOTES): chunks per guarded block, input-span bytes per tile, outputs per lane.
constexpr int SPLIT_CB = 4;
constexpr int SPLIT_SPAN_BYTES = 32768;
constexpr int SPLIT_U = 2;

namespace sdrhip {

namespace {

struct SplitArgs {
    const void* in;       // element 0 = the launch's in_base
    float* out;           // out[0] = first output of the launch
    const float* taps;    // nrows rows of row_stride floats
    int nrows, row_stride, row0;
    int nch;              // chunks of M taps per output
    int NC;               // interleaved output classes (1: filter / decimator; numGroups: resampler)
    int period;           // inputs consumed per cycle of NC outputs
    int pre[64];          // window start of class c within its cycle
    int64_t pos0;         // window start of output 0, relative to `in`
    int64_t in_len;       // elements of `in` that may be read
    int count;            // outputs
    int NN;               // cycles per workgroup
    int span;             // input elements staged per workgroup
    int reach;            // elements one window touches (SYM: the full length 2n)
    float gain;
    int apply_gain;
};

__device__ __forceinline__ float ld_zero(float) { return 0.0f; }
__device__ __forceinline__ float2 ld_zero(float2) { return make_float2(0.0f, 0.0f); }
__device__ __forceinline__ float mac(float acc, float h, float x) { return acc + h * x; }
__device__ __forceinline__ float2 mac(float2 acc, float h, float2 x) { return make_float2(acc.x + h * x.x, acc.y + h * x.y); }
__device__ __forceinline__ float add2(float a, float b) { return a + b; }
__device__ __forceinline__ float2 add2(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
// Partner exchange inside a group of M lanes as DPP operands of the add (no LDS round trip).  STEP 1 / 2: the xor-1 /
// xor-2 partner (quad_perm).  STEP 4: lane l receives lane l+4 of its row (row_shl:4) -- the xor-4 partner for the lower
// half of every 8-lane group, which is the only half whose result is used afterwards.
template <int STEP>
__device__ __forceinline__ float dpp_partner(float v)
{
    constexpr int ctrl = STEP == 1 ? 0xB1 /* quad_perm:[1,0,3,2] */ : STEP == 2 ? 0x4E /* quad_perm:[2,3,0,1] */ : 0x104 /* row_shl:4 */;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}
template <int STEP> __device__ __forceinline__ float add_partner(float a) { return a + dpp_partner<STEP>(a); }
template <int STEP> __device__ __forceinline__ float2 add_partner(float2 a)
{
    return make_float2(a.x + dpp_partner<STEP>(a.x), a.y + dpp_partner<STEP>(a.y));
}
__device__ __forceinline__ float scale1(float v, float g) { return v * g; }
__device__ __forceinline__ float2 scale1(float2 v, float g) { return make_float2(v.x * g, v.y * g); }

// ORD 0: tree over adjacent lanes (xor 1, 2, .., M/2).  ORD 1: fold the two halves first (xor M/2), then the tree over
// the lower half.  Lane 0 of the group ends up with the reference's value; it always adds (its own) + (the partner's).
template <int M, int ORD, class E>
__device__ __forceinline__ E fold(E a)
{
    if constexpr (ORD == 0) {
        if constexpr (M >= 2) a = add_partner<1>(a);
        if constexpr (M >= 4) a = add_partner<2>(a);
        if constexpr (M >= 8) a = add_partner<4>(a);
    } else {
        if constexpr (M == 8) { a = add_partner<4>(a); a = add_partner<1>(a); a = add_partner<2>(a); }
        else { a = add_partner<2>(a); a = add_partner<1>(a); }
    }
    return a;
}

template <bool SYM, int M, class E>
__device__ __forceinline__ E fetch(const E* x, int i, int mirror)
{
    if constexpr (SYM) return add2(x[M * i], x[mirror - M * i]);
    else return x[M * i];
}

// The arithmetic of one tile.  FULL: every output of the tile exists (all tiles but the last), so no validity logic.
// Window / output addresses advance by constant strides, chunks go in blocks of CB with ONE wave-uniform guard per
// block (the last nch % CB chunks are guarded one by one); no early exit anywhere, so the loops unroll completely and
// tp[] keeps static register indices.  EXACT: nch == NCHMAX, no guards at all (one straight-line block per run).
template <class E, int M, int ORD, bool SYM, int NCHMAX, int U, bool FULL, bool EXACT>
__device__ __forceinline__ void split_compute(const SplitArgs& a, const E* __restrict__ lin, E* __restrict__ lout, int full,
                                              int rem, int w, int l, int og)
{
    constexpr int G = 64 / M;                       // outputs one wave instruction works on
    constexpr int CB = NCHMAX >= SPLIT_CB ? SPLIT_CB : NCHMAX;
    const int QN = a.NN / G;                        // output groups per class (a multiple of U)
    const float g = a.apply_gain ? a.gain : 1.0f;   // r * 1.0f == r bit for bit
    const int mirror = a.reach - 1 - 2 * l;
    const int xstep = G * a.period;                 // elements between the windows of consecutive groups
    const int ostep = G * a.NC;
    float tp[NCHMAX];
    for (int c = 0; c < a.NC; c++) {
        {
            int row = a.row0 + c;
            if (row >= a.nrows) row -= a.nrows;
            const float* rp = a.taps + (size_t)row * a.row_stride + l;
#pragma unroll
            for (int i = 0; i < NCHMAX; i++)
                if (EXACT || i < a.nch) tp[i] = rp[M * i];
        }
        // the four waves take runs of U groups round-robin
        const E* xq = lin + a.pre[c] + l + (w * U * G + og) * a.period;
        E* oq = lout + (w * U * G + og) * a.NC + c;
        for (int q = w * U; q < QN; q += 4 * U, xq += 4 * U * xstep, oq += 4 * U * ostep) {
            const E* x[U];
            E acc[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if constexpr (FULL) {
                    valid[u] = true;
                    x[u] = xq + u * xstep;
                } else {
                    const int n = (q + u) * G + og;
                    valid[u] = n < full || (n == full && c < rem);
                    x[u] = valid[u] ? xq + u * xstep : lin + a.pre[c] + l;   // a window that is certainly staged
                }
                acc[u] = ld_zero(E{});
            }
#pragma unroll
            for (int b = 0; b < NCHMAX; b += CB) {
                if (EXACT || b + CB <= a.nch) {
                    E v[CB][U];
#pragma unroll
                    for (int i = 0; i < CB; i++)
#pragma unroll
                        for (int u = 0; u < U; u++) v[i][u] = fetch<SYM, M>(x[u], b + i, mirror);
#pragma unroll
                    for (int i = 0; i < CB; i++)
#pragma unroll
                        for (int u = 0; u < U; u++) acc[u] = mac(acc[u], tp[b + i], v[i][u]);
                } else {
#pragma unroll
                    for (int i = b; i < b + CB; i++)
                        if (i < a.nch) {
#pragma unroll
                            for (int u = 0; u < U; u++) acc[u] = mac(acc[u], tp[i], fetch<SYM, M>(x[u], i, mirror));
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const E r = scale1(fold<M, ORD>(acc[u]), g);
                if (l == 0 && valid[u]) oq[u * ostep] = r;
            }
        }
    }
}

template <bool CPLX, int M, int ORD, bool SYM, int NCHMAX, int U, bool EXACT>
__global__ void __launch_bounds__(256) k_split(SplitArgs a)
{
    using E = std::conditional_t<CPLX, float2, float>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* lds = reinterpret_cast<E*>(smem_raw);
    E* lout = lds + ((a.span + 7) & ~3);          // the span may be staged up to 3 elements early (16-byte alignment)
    const int tid = threadIdx.x;
    const int64_t cyc0 = (int64_t)blockIdx.x * a.NN;

    // Stage the tile's input span with 16-byte loads, several in flight per thread.  The span is fetched from the 16-byte
    // boundary at or below its first element (`shift` elements early, so LDS index = span index + shift); vectors that
    // poke outside [0, in_len) are assembled element by element, zero-filled (only discarded lanes read those).
    constexpr int VPE = CPLX ? 2 : 4;               // elements per 16-byte vector
    const E* src = reinterpret_cast<const E*>(a.in);
    const int64_t s0 = a.pos0 + cyc0 * a.period;
    const int shift = (int)((reinterpret_cast<uintptr_t>(src + s0) & 15) / sizeof(E));
    {
        const int nvec = (a.span + shift + VPE - 1) / VPE;
        const int64_t first = s0 - shift;            // element index of vector 0's first element
        constexpr int BATCH = 4;
        for (int v0 = 0; v0 < nvec; v0 += 256 * BATCH) {
            uint4 r[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const int v = v0 + k * 256 + tid;
                const int64_t e0 = first + (int64_t)v * VPE;
                r[k] = make_uint4(0u, 0u, 0u, 0u);
                if (v < nvec) {
                    if (e0 >= 0 && e0 + VPE <= a.in_len) {
                        r[k] = *reinterpret_cast<const uint4*>(src + e0);
                    } else {
                        E tmp[VPE];
#pragma unroll
                        for (int j = 0; j < VPE; j++) tmp[j] = (e0 + j >= 0 && e0 + j < a.in_len) ? src[e0 + j] : ld_zero(E{});
                        r[k] = *reinterpret_cast<const uint4*>(tmp);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const int v = v0 + k * 256 + tid;
                if (v < nvec) *reinterpret_cast<uint4*>(lds + (size_t)v * VPE) = r[k];
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    // cycles of this tile that are complete, and how many classes of the next (partial) one exist
    const int64_t whole = a.count / a.NC - cyc0;
    const int full = whole < a.NN ? (int)whole : a.NN;
    const int rem = whole < a.NN ? a.count % a.NC : 0;
    if (full == a.NN) split_compute<E, M, ORD, SYM, NCHMAX, U, true, EXACT>(a, lds + shift, lout, full, rem, tid >> 6, lane % M, lane / M);
    else split_compute<E, M, ORD, SYM, NCHMAX, U, false, EXACT>(a, lds + shift, lout, full, rem, tid >> 6, lane % M, lane / M);
    __syncthreads();
    {
        const int64_t o0 = cyc0 * a.NC;
        const int64_t left = a.count - o0;
        const int nout = left < (int64_t)a.NN * a.NC ? (int)left : a.NN * a.NC;
        E* dst = reinterpret_cast<E*>(a.out) + o0;
        for (int e = tid; e < nout; e += 256) dst[e] = lout[e];
    }
}

// Tile geometry: as many cycles as fit a 32 KiB input span (64 KiB when a large decimation factor needs it) and 1024
// outputs, in whole runs of U groups of G = 64/M outputs (U = 2 independent accumulator chains per lane -- 4 and 8 measured no better -- when the tile is
// big enough, else 1).
bool plan_tile(SplitArgs& a, int M, bool cplx, int& U)
{
    const int G = 64 / M;
    const int esz = cplx ? 8 : 4;
    const int pre_max = a.pre[a.NC - 1];
    for (int bytes = SPLIT_SPAN_BYTES; bytes <= 65536; bytes *= 2) {
        const int64_t room = (int64_t)bytes / esz - pre_max - a.reach;
        if (room < 0) continue;
        int64_t NN = room / a.period + 1;
        if (NN > 1024 / a.NC) NN = 1024 / a.NC;
        if (NN >= 4 * SPLIT_U * G) { U = SPLIT_U; NN = NN / (SPLIT_U * G) * (SPLIT_U * G); }
        else if (bytes == 65536 && NN >= G) { U = 1; NN = NN / G * G; }
        else continue;
        a.NN = (int)NN;
        a.span = (int)((NN - 1) * a.period + pre_max + a.reach);
        return true;
    }
    return false;
}

template <bool CPLX, int M, int ORD, bool SYM, int NCHMAX, int U, bool EXACT>
void launch_kernel(hipStream_t s, const SplitArgs& a, unsigned grid, size_t lds_bytes)
{
    auto kern = k_split<CPLX, M, ORD, SYM, NCHMAX, U, EXACT>;
    // per instantiation and device: tiles of large decimation factors exceed the 64 KiB default cap
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, s, a);
}

template <bool CPLX, int M, int ORD, bool SYM, int U>
void launch_variant_u(hipStream_t s, const SplitArgs& a)
{
    const int64_t ncyc = ((int64_t)a.count + a.NC - 1) / a.NC;
    const unsigned grid = (unsigned)((ncyc + a.NN - 1) / a.NN);
    const size_t esz = CPLX ? 8 : 4;
    const size_t lds_bytes = ((size_t)((a.span + 7) & ~3) + (size_t)a.NN * a.NC) * esz;
    // Real data, exact tap-chunk counts of the usual filter lengths: guard-free code (measured +20 % on the 128-tap real
    // decimator; on complex data the scheduler hoists more reads than there are registers for and it is 2x slower).
    constexpr bool EX = U == SPLIT_U && !CPLX;
    if (EX && a.nch == 8) launch_kernel<CPLX, M, ORD, SYM, 8, U, EX>(s, a, grid, lds_bytes);
    else if (EX && a.nch == 16) launch_kernel<CPLX, M, ORD, SYM, 16, U, EX>(s, a, grid, lds_bytes);
    else if (EX && a.nch == 32) launch_kernel<CPLX, M, ORD, SYM, 32, U, EX>(s, a, grid, lds_bytes);
    else if (a.nch <= 8) launch_kernel<CPLX, M, ORD, SYM, 8, U, false>(s, a, grid, lds_bytes);
    else if (a.nch <= 16) launch_kernel<CPLX, M, ORD, SYM, 16, U, false>(s, a, grid, lds_bytes);
    else if (a.nch <= 32) launch_kernel<CPLX, M, ORD, SYM, 32, U, false>(s, a, grid, lds_bytes);
    else launch_kernel<CPLX, M, ORD, SYM, 64, U, false>(s, a, grid, lds_bytes);
}

template <bool CPLX, int M, int ORD, bool SYM>
void launch_variant(hipStream_t s, const SplitArgs& a, int U)
{
    if (U == SPLIT_U) launch_variant_u<CPLX, M, ORD, SYM, SPLIT_U>(s, a);
    else launch_variant_u<CPLX, M, ORD, SYM, 1>(s, a);
}

// M / ORD of a summation order
struct OrderInfo { int M; int ord; };
bool real_order(int lanes, OrderInfo& o)
{
    if (lanes == 8) { o = {8, 0}; return true; }
    if (lanes == 4) { o = {4, 0}; return true; }
    return false;   // scalar: one sequential chain, nothing to split
}
bool cplx_order(ComplexOrder c, OrderInfo& o)
{
    switch (c) {
        case CO_L4: o = {4, 0}; return true;
        case CO_L2: o = {2, 0}; return true;
        case CO_X4: o = {8, 1}; return true;
        case CO_X2: o = {4, 1}; return true;
        default: return false;
    }
}

template <bool CPLX>
bool dispatch(hipStream_t s, SplitArgs& a, OrderInfo o, bool sym)
{
    if (a.nch < 1 || a.nch > 64) return false;   // the taps of a lane live in registers
    int U = 1;
    if (!plan_tile(a, o.M, CPLX, U)) return false;
    if constexpr (!CPLX) {
        if (o.M == 8) { if (sym) launch_variant<false, 8, 0, true>(s, a, U); else launch_variant<false, 8, 0, false>(s, a, U); }
        else { if (sym) launch_variant<false, 4, 0, true>(s, a, U); else launch_variant<false, 4, 0, false>(s, a, U); }
    } else {
        if (sym) return false;   // no descriptor builds the symmetric complex kernels (drop-in symbols only): generic path
        if (o.ord == 0) {
            if (o.M == 4) launch_variant<true, 4, 0, false>(s, a, U); else launch_variant<true, 2, 0, false>(s, a, U);
        } else if (o.M == 8) {
            launch_variant<true, 8, 1, false>(s, a, U);
        } else {
            launch_variant<true, 4, 1, false>(s, a, U);
        }
    }
    return true;
}

std::atomic<long long> g_split_launches{0};

constexpr int SPLIT_MIN_OUTPUTS = 4096;   // below this a launch is latency-bound either way: leave it to the generic kernel

}  // namespace

long long split_launch_count() { return g_split_launches.load(); }

// Filter / decimator on real or complex data.  d_taps: plain taps (sym: the half taps), ntaps of them.
bool launch_fir_split(hipStream_t s, const Geom& g, bool cplx, int lanes, ComplexOrder corder, bool sym, const float* d_taps,
                      int ntaps, const float* d_cross_taps, const float* d_in, float* d_out, float gain, bool apply_gain)
{
    if (g.I != 1 || g.count < SPLIT_MIN_OUTPUTS || g.seamBI < 0) return false;
    if (g.seamBI != 0 && d_cross_taps == nullptr) return false;
    OrderInfo o;
    if (!(cplx ? cplx_order(corder, o) : real_order(lanes, o))) return false;
    if (ntaps % o.M != 0) return false;
    if (cplx && apply_gain) return false;
    SplitArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.taps = d_taps;
    a.nrows = 1;
    a.row_stride = ntaps;
    a.row0 = 0;
    a.nch = ntaps / o.M;
    a.NC = 1;
    a.period = g.D;
    a.pre[0] = 0;
    a.pos0 = g.k_begin * g.D - g.in_base;
    a.reach = sym ? 2 * ntaps : ntaps;
    if (a.reach != g.Lp) return false;
    a.in_len = a.pos0 + (int64_t)(g.count - 1) * g.D + a.reach;
    a.count = g.count;
    a.gain = gain;
    a.apply_gain = apply_gain ? 1 : 0;
    if (!(cplx ? dispatch<true>(s, a, o, sym) : dispatch<false>(s, a, o, sym))) return false;
    g_split_launches++;
    if (g.seamBI != 0) {
        int64_t first, last;
        seam_range(g, first, last);
        if (last >= first) {
            const int nseams = (int)(last - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            const int64_t total = (int64_t)nseams * per;
            const dim3 grid((unsigned)((total + 255) / 256));
            if (cplx)
                hipLaunchKernelGGL(k_fir_cplx_crossfix<false>, grid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per);
            else
                hipLaunchKernelGGL(k_fir_real_crossfix, grid, dim3(256), 0, s, g, d_cross_taps, d_in, d_out, first, nseams, per, gain,
                                   apply_gain ? 1 : 0);
        }
    }
    return true;
}

// Rational resampler on real or complex data (groups = prepareCoeffs rows, FilterInternal.hs:297-319).
bool launch_resample_split(hipStream_t s, const Geom& g, bool cplx, int lanes, ComplexOrder corder, const ResampTable& t,
                           const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out)
{
    if (g.count < SPLIT_MIN_OUTPUTS || g.seamBI < 0 || t.force_seq || t.ext != nullptr) return false;
    if (g.seamBI != 0 && d_plain_taps == nullptr) return false;
    OrderInfo o;
    if (!(cplx ? cplx_order(corder, o) : real_order(lanes, o))) return false;
    if (t.nloop % o.M != 0 || t.ngroups < 1 || t.ngroups > 64) return false;
    SplitArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.taps = d_groups;
    a.nrows = t.ngroups;
    a.row_stride = t.row_stride;
    a.row0 = t.group0;
    a.nch = t.nloop / o.M;
    a.NC = t.ngroups;
    a.period = t.period;
    for (int c = 0; c < t.ngroups; c++) a.pre[c] = t.pre[c];
    a.pos0 = t.pos0;
    a.reach = t.nloop;
    const int64_t last = g.count - 1;
    a.in_len = t.pos0 + (last / t.ngroups) * t.period + t.pre[last % t.ngroups] + t.nloop;
    a.count = g.count;
    a.gain = 1.0f;
    a.apply_gain = 0;
    if (!(cplx ? dispatch<true>(s, a, o, false) : dispatch<false>(s, a, o, false))) return false;
    g_split_launches++;
    if (g.seamBI != 0) {
        int64_t first, lastb;
        seam_range(g, first, lastb);
        if (lastb >= first) {
            const int nseams = (int)(lastb - first + 1);
            const int per = (g.Lp - 1 + g.D - 1) / g.D;
            const int64_t total = (int64_t)nseams * per;
            const dim3 grid((unsigned)((total + 255) / 256));
            if (cplx)
                hipLaunchKernelGGL(k_resample_crossfix<true>, grid, dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in, d_out, first,
                                   nseams, per);
            else
                hipLaunchKernelGGL(k_resample_crossfix<false>, grid, dim3(256), 0, s, g, d_plain_taps, t.ntaps_plain, d_in, d_out, first,
                                   nseams, per);
        }
    }
    return true;
}

}  // namespace sdrhip
