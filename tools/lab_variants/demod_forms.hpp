// (tools/lab_variants) The full set of fmDemod restatements as of round 5 -- the product keeps three of them (sdr_amd/csrc/demod.hpp).
// demod.hpp -- fmDemod's per-sample arithmetic (Demod.hs:21-46 + GHC base atan2 + fdlibm atanf), shared by the
// stand-alone kernel (kernels_chain.hip) and the fused tail kernel (kernels_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrhip {

// ---------------------------------------------------------------------------
// K3  fmDemod, Demod.hs:21-46 (+ GHC base atan2, glibc/fdlibm atanf).  Same
// operations in the same order as device_atanf/ghc_atan2 of kernels_generic.hip, but
// every data-dependent branch is a select: on noise-like IQ all of fdlibm's five
// argument ranges occur in every wave, so branches would serialise all of them.
// ---------------------------------------------------------------------------
// Every alternative is first computed into a local and then chosen with a ternary whose arms are locals or constants:
// clang lowers exactly that shape to `select` (v_cndmask); an arithmetic expression inside an arm, or a nested ternary,
// becomes control flow instead -- one basic block per range per sample, which serialises the samples a thread works on
// (no instruction-level parallelism across them) and costs exec-mask bookkeeping.  Speculating the arms is free of side
// effects: they are single additions / multiplications.
__device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }

__device__ __forceinline__ float atanf_sel(float x)
{
    const uint32_t hx = __float_as_uint(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    const float ax = __uint_as_float(ix);
    const bool tiny_range = ix < 0x3ee00000u;                 // |x| < 0.4375: no reduction, keeps sign
    const bool r0 = !tiny_range & (ix < 0x3f300000u);
    const bool r1 = !tiny_range & !r0 & (ix < 0x3f980000u);
    const bool r2 = !tiny_range & !r0 & !r1 & (ix < 0x401c0000u);
    // numerator / denominator of the reduction; x/1 is exact so the unreduced range shares the divide
    const float n0 = 2.0f * ax - 1.0f, n1 = ax - 1.0f, n2 = ax - 1.5f;
    const float d0 = 2.0f + ax, d1 = ax + 1.0f, d2 = 1.0f + 1.5f * ax;
    const float num = sel(tiny_range, x, sel(r0, n0, sel(r1, n1, sel(r2, n2, -1.0f))));
    const float den = sel(tiny_range, 1.0f, sel(r0, d0, sel(r1, d1, sel(r2, d2, ax))));
    const float xr = num / den;
    const float hv = sel(r0, 4.6364760399e-01f, sel(r1, 7.8539812565e-01f, sel(r2, 9.8279368877e-01f, 1.5707962513e+00f)));
    const float lv = sel(r0, 5.0121582440e-09f, sel(r1, 3.7748947079e-08f, sel(r2, 3.4473217170e-08f, 7.5497894159e-08f)));
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float small = xr - xr * (s1 + s2);
    const float zz = hv - ((xr * (s1 + s2) - lv) - xr);
    const float nzz = -zz;
    float res = sel(tiny_range, small, sel(neg, nzz, zz));
    res = sel(ix < 0x31000000u, x, res);                       // |x| < 2^-29
    const float big = 1.5707962513e+00f + 7.5497894159e-08f;
    const float xx = x + x;
    const float huge_res = sel(ix > 0x7f800000u, xx, sel(neg, -big, big));
    res = sel(ix >= 0x4c000000u, huge_res, res);               // |x| >= 2^25, inf, nan
    return res;
}

__device__ __forceinline__ bool negzero(float v) { return __float_as_uint(v) == 0x80000000u; }

// GHC RealFloat default atan2, evaluated once on (|case|-folded) operands
__device__ __forceinline__ float ghc_atan2_sel(float y, float x)
{
    const float pi = 3.14159274101257324f;
    // clause 4 (negate (atan2 (negate y) x)) folds the lower half-plane onto the upper one
    const bool nzx = negzero(x), nzy = negzero(y);
    const bool fold = ((x <= 0.0f) & (y < 0.0f)) | ((x < 0.0f) & nzy) | (nzx & nzy);
    const bool c1 = x > 0.0f;
    const bool flip = !c1 & fold;
    const float ny = -y;
    const float yy = sel(flip, ny, y);
    const float a = atanf_sel(yy / x);
    const bool xz = x == 0.0f, xn = x < 0.0f, yp = yy > 0.0f, yz = yy == 0.0f;
    const float pa = pi + a, xy = x + yy;
    float r = xy;                                             // applied last to first: the first matching clause wins
    r = sel(xz & yz, yy, r);
    r = sel(yz & (xn | nzx), pi, r);
    r = sel(xn & yp, pa, r);
    r = sel(xz & yp, pi / 2.0f, r);
    r = sel(c1, a, r);                                        // clause 1 (never folded: x > 0)
    const float nr = -r;
    return sel(flip, nr, r);
}

__device__ __forceinline__ float fm_phase_sel(float2 cur, float2 prev)
{
    const float nd = -prev.y;
    const float re = cur.x * prev.x - cur.y * nd;
    const float im = cur.x * nd + cur.y * prev.x;
    const float p = ghc_atan2_sel(im, re);
    return sel((re == 0.0f) & (im == 0.0f), 0.0f, p);
}

// The same arithmetic written with nested ternaries: clang turns those into control flow (a basic block per argument range).
// On its own that is the faster form -- the stand-alone fmDemod kernel runs eight waves per SIMD and skips the ranges a
// wave does not meet (0.168 ms against 0.183 ms per 2^26 samples for the select form) -- while inside the fused tail kernel,
// at four waves per SIMD, the select form's instruction-level parallelism across a thread's four samples wins.
__device__ __forceinline__ float atanf_tern(float x)
{
    const uint32_t hx = __float_as_uint(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    const float ax = __uint_as_float(ix);
    const bool tiny_range = ix < 0x3ee00000u;                 // |x| < 0.4375: no reduction, keeps sign
    const bool r0 = !tiny_range && ix < 0x3f300000u;
    const bool r1 = !tiny_range && !r0 && ix < 0x3f980000u;
    const bool r2 = !tiny_range && !r0 && !r1 && ix < 0x401c0000u;
    // numerator / denominator of the reduction; x/1 is exact so the unreduced range shares the divide
    const float num = tiny_range ? x : r0 ? (2.0f * ax - 1.0f) : r1 ? (ax - 1.0f) : r2 ? (ax - 1.5f) : -1.0f;
    const float den = tiny_range ? 1.0f : r0 ? (2.0f + ax) : r1 ? (ax + 1.0f) : r2 ? (1.0f + 1.5f * ax) : ax;
    const float xr = num / den;
    const float hv = r0 ? 4.6364760399e-01f : r1 ? 7.8539812565e-01f : r2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lv = r0 ? 5.0121582440e-09f : r1 ? 3.7748947079e-08f : r2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float small = xr - xr * (s1 + s2);
    const float zz = hv - ((xr * (s1 + s2) - lv) - xr);
    float res = tiny_range ? small : (neg ? -zz : zz);
    res = ix < 0x31000000u ? x : res;                          // |x| < 2^-29
    const float big = 1.5707962513e+00f + 7.5497894159e-08f;
    const float huge_res = ix > 0x7f800000u ? x + x : (neg ? -big : big);
    res = ix >= 0x4c000000u ? huge_res : res;                  // |x| >= 2^25, inf, nan
    return res;
}


// GHC RealFloat default atan2, evaluated once on (|case|-folded) operands
__device__ __forceinline__ float ghc_atan2_tern(float y, float x)
{
    const float pi = 3.14159274101257324f;
    // clause 4 (negate (atan2 (negate y) x)) folds the lower half-plane onto the upper one
    const bool fold = (x <= 0.0f && y < 0.0f) || (x < 0.0f && negzero(y)) || (negzero(x) && negzero(y));
    const bool c1 = x > 0.0f;
    const float yy = (!c1 && fold) ? -y : y;
    const float a = atanf_tern(yy / x);
    const bool xz = x == 0.0f, xn = x < 0.0f, yp = yy > 0.0f, yz = yy == 0.0f;
    const float r = c1 ? a                                    // clause 1 (never folded: x > 0)
                  : (xz && yp) ? pi / 2.0f
                  : (xn && yp) ? pi + a
                  : (yz && (xn || negzero(x))) ? pi
                  : (xz && yz) ? yy
                  : x + yy;
    return (!c1 && fold) ? -r : r;
}

__device__ __forceinline__ float fm_phase_tern(float2 cur, float2 prev)
{
    const float nd = -prev.y;
    const float re = cur.x * prev.x - cur.y * nd;
    const float im = cur.x * nd + cur.y * prev.x;
    const float p = ghc_atan2_tern(im, re);
    return (re == 0.0f && im == 0.0f) ? 0.0f : p;
}


// ---------------------------------------------------------------------------
// Round 4: the COMMON case on its own.  Of the clauses above only two ever apply to a sample whose product re + i*im is finite
// and off both axes -- atan2's clause 1 (re > 0) and clause 3 (re < 0, folded im > 0) -- and of atanf's argument ranges the two
// outermost (|q| < 2^-29 and |q| >= 2^25, inf, NaN) need a ratio no FM signal produces.  One test on the RATIO q = yy / re
// identifies everything else: q is +-0 or denormal when im is (or underflows against re), infinite when re is zero or im
// infinite, NaN when either is NaN or 0/0 or inf/inf -- so "2^-29 <= |q| < 2^25" implies re and im finite, non-zero, non-NaN.
// For such a sample this function performs exactly the operations ghc_atan2_sel / atanf_sel select (same operands, same order);
// for any other it sets `rare` and its value is to be discarded -- the caller votes across the wave and re-evaluates with
// fm_phase_sel (kernels_chain.hip's tile loader: the vote fails once in a blue moon; all-zero IQ takes the full form everywhere).
// 94 VALU instructions per sample against 119: 20 of the 28 selects, the clause compares and the huge / tiny argument arms go.
// The sign of atanf's reduced ranges is copysign(zz, q) (one v_bfi_b32): zz = hi - ((t - lo) - xr) lies in [0.41, 1.58].
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fm_phase_common(float2 cur, float2 prev, bool& rare)
{
    const float pi = 3.14159274101257324f;
    const float nd = -prev.y;
    const float re = cur.x * prev.x - cur.y * nd;
    const float im = cur.x * nd + cur.y * prev.x;
    const bool c1 = re > 0.0f;
    const bool flip = !c1 & (im < 0.0f);                      // clause 4 (re < 0 here), undone at the end
    const float nim = -im;
    const float yy = sel(flip, nim, im);
    const float q = yy / re;
    const uint32_t ix = __float_as_uint(q) & 0x7fffffffu;
    const float ax = __uint_as_float(ix);
    rare = (ix - 0x31000000u) >= (0x4c000000u - 0x31000000u);
    // atanf's argument reduction, innermost range last so that four plain thresholds serve all four select chains
    const bool t0 = ix < 0x3ee00000u, t1 = ix < 0x3f300000u, t2 = ix < 0x3f980000u, t3 = ix < 0x401c0000u;
    const float n0 = 2.0f * ax - 1.0f, n1 = ax - 1.0f, n2 = ax - 1.5f;
    const float d0 = 2.0f + ax, d1 = ax + 1.0f, d2 = 1.0f + 1.5f * ax;
    const float num = sel(t0, q, sel(t1, n0, sel(t2, n1, sel(t3, n2, -1.0f))));
    const float den = sel(t0, 1.0f, sel(t1, d0, sel(t2, d1, sel(t3, d2, ax))));
    const float xr = num / den;
    const float hv = sel(t1, 4.6364760399e-01f, sel(t2, 7.8539812565e-01f, sel(t3, 9.8279368877e-01f, 1.5707962513e+00f)));
    const float lv = sel(t1, 5.0121582440e-09f, sel(t2, 3.7748947079e-08f, sel(t3, 3.4473217170e-08f, 7.5497894159e-08f)));
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float t = xr * (s1 + s2);
    const float small = xr - t;
    const float zz = hv - ((t - lv) - xr);
    const float a = sel(t0, small, __builtin_copysignf(zz, q));
    const float pa = pi + a;
    const float r = sel(c1, a, pa);
    const float nr = -r;
    return sel(flip, nr, r);
}

// ---------------------------------------------------------------------------
// The common case with atanf's argument reduction looked up instead of selected (round 4, the resampler's fused loader).
// fdlibm's four reductions and the unreduced range are all of the shape  xr = (A*ax + B) / (C*ax + D),  result = hi - ((t - lo) - xr):
//     |x| < 7/16      (1*ax + -0) / (0*ax + 1)        hi = lo = 0      [x/1; 0 - ((t - 0) - xr) = xr - t: the unreduced polynomial]
//     < 11/16         (2*ax + -1) / (1*ax + 2)        atan(0.5)
//     < 19/16         (1*ax + -1) / (1*ax + 1)        atan(1)
//     < 39/16         (1*ax + -1.5) / (1.5*ax + 1)    atan(1.5)
//     otherwise       (0*ax + -1) / (1*ax + -0)       atan(inf)
// with every product by 0, 1, 2 exact and every sum the one fdlibm writes (a + b = b + a; x + -0 = x), so each row yields the bits of
// the expression it replaces for finite ax > 0; the sign goes on last by copysign (the polynomial is odd: evaluating it on |x| and
// negating is exact).  The range thresholds are multiples of 2^18 in the float's bit pattern, so the row is a direct index:
// clamp(bits >> 18, 0xfb7, 0x1007) - 0xfb7, 81 rows of (A, B, C, D) and 81 of (hi, lo) in LDS, filled by the first 81 threads of a workgroup.
// 9 VALU instructions + two LDS reads where the select chains take 26; 73 per sample all told.
// ---------------------------------------------------------------------------

// The compiler's IEEE f32 division is v_div_scale x 2, v_rcp, two Newton steps on the reciprocal, three fma steps on the quotient,
// v_div_fmas, v_div_fixup.  For a denominator in [1, 2^26] and a numerator that is 0 or of magnitude in [2^-29, 2^25] -- atanf's
// reduced fraction for an argument inside the common case -- neither v_div_scale rescales (no operand or quotient near the denormal
// or overflow range: both return their input and clear VCC), v_div_fmas is then a plain fma and v_div_fixup returns the quotient
// it is given (finite non-zero operands; a zero numerator gives the same +0 through the arithmetic): the same eight operations
// without the three that do nothing.  Only called for lanes whose result is used when the wave's vote says "common".
#ifndef SDRHIP_DEMOD_DIV2_PLAIN
#define SDRHIP_DEMOD_DIV2_PLAIN 1
#endif
__device__ __forceinline__ float div_unscaled(float num, float den)
{
    const float r0 = __builtin_amdgcn_rcpf(den);
    const float e0 = __builtin_fmaf(-den, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = num * r1;
    const float e1 = __builtin_fmaf(-den, q0, num);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-den, q1, num);
    return __builtin_fmaf(e2, r1, q1);
}

constexpr int kAtanRows = 0x1007 - 0xfb7 + 1;      // 81
constexpr int kAtanRowFloats = 6;                  // per row: a float4 (A, B, C, D) in the first part of the table, a float2 (hi, lo) in the second --
                                                   // 16-byte rows use all 64 banks (16 bank classes), 32-byte rows only half of them (8)

__device__ __forceinline__ void atan_table_fill(float* tbl, int row)
{
    if (row >= kAtanRows) return;
    const uint32_t t = 0xfb7u + (uint32_t)row;
    const int k = row == 0 ? 0 : t < 0xfccu ? 1 : t < 0xfe6u ? 2 : t < 0x1007u ? 3 : 4;
    const float A[5] = {1.0f, 2.0f, 1.0f, 1.0f, 0.0f};
    const float Bc[5] = {-0.0f, -1.0f, -1.0f, -1.5f, -1.0f};
    const float C[5] = {0.0f, 1.0f, 1.0f, 1.5f, 1.0f};
    const float D[5] = {1.0f, 2.0f, 1.0f, 1.0f, -0.0f};
    const float H[5] = {0.0f, 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float Lo[5] = {0.0f, 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    float a = A[0], b = Bc[0], c = C[0], d = D[0], h = H[0], l = Lo[0];
#pragma unroll
    for (int i = 1; i < 5; i++)
        if (k == i) { a = A[i]; b = Bc[i]; c = C[i]; d = D[i]; h = H[i]; l = Lo[i]; }
    reinterpret_cast<float4*>(tbl)[row] = make_float4(a, b, c, d);
    reinterpret_cast<float2*>(tbl + 4 * kAtanRows)[row] = make_float2(h, l);
}

__device__ __forceinline__ float fm_phase_common_tbl(float2 cur, float2 prev, bool& rare, const float* tbl)
{
    const float pi = 3.14159274101257324f;
    const float nd = -prev.y;
    const float re = cur.x * prev.x - cur.y * nd;
    const float im = cur.x * nd + cur.y * prev.x;
    const bool c1 = re > 0.0f;
    const bool flip = !c1 & (im < 0.0f);
    const float nim = -im;
    const float yy = sel(flip, nim, im);
    const float q = yy / re;
    const uint32_t ix = __float_as_uint(q) & 0x7fffffffu;
    const float ax = __uint_as_float(ix);
    rare = (ix - 0x31000000u) >= (0x4c000000u - 0x31000000u);
    const uint32_t tq = ix >> 18;
    const uint32_t tc = tq < 0xfb7u ? 0xfb7u : tq > 0x1007u ? 0x1007u : tq;      // v_med3_u32
    const float4 abcd = reinterpret_cast<const float4*>(tbl)[tc - 0xfb7u];
    const float2 hl = reinterpret_cast<const float2*>(tbl + 4 * kAtanRows)[tc - 0xfb7u];
    const float num = abcd.x * ax + abcd.y;
    const float den = abcd.z * ax + abcd.w;
#if SDRHIP_DEMOD_DIV2_PLAIN
    const float xr = div_unscaled(num, den);
#else
    const float xr = num / den;
#endif
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float t = xr * (s1 + s2);
    const float zz = hl.x - ((t - hl.y) - xr);
    const float a = __builtin_copysignf(zz, q);
    const float pa = pi + a;
    const float r = sel(c1, a, pa);
    const float nr = -r;
    return sel(flip, nr, r);
}

// ---------------------------------------------------------------------------
// Round 5: TWO samples per call, packed.  tools/k4lab/issue_bench.hip: a wave issues at most one VALU instruction every four cycles;
// a scalar f32 operation occupies the SIMD for two, so a SIMD needs two READY waves to run scalar arithmetic at full rate (one
// wave: 0.43-0.52 G instructions/s, two: 0.83-0.97), while a packed operation (v_pk_mul / add / fma_f32: two lanes' worth in four
// cycles) keeps it busy from a single wave (0.44-0.48 of 0.55-0.57).  The fused loaders run few waves per SIMD and those wait on LDS
// lookups and barriers, so fmDemod is issue-bound, not execution-bound: this form issues ~97 instructions per two samples
// instead of 140 -- the complex product as three packed operations per sample ((cx,cx)*(px,-py), (cy,cy)*(-py,px), one add with the
// low half negated: the same four products and the same two sums as fm_phase_common), the fma steps of both divisions and the whole
// polynomial across the two samples.  The selects on sign conditions become bit operations, exact for every sample that does not
// set `rare` (finite, non-zero re and im -- see fm_phase_common):
//     yy = flip ? -im : im   =  im with its sign cleared when re is negative          (flip = re < 0 && im < 0)
//     r  = re > 0 ? a : pi + a  =  a + (re < 0 ? pi : +0)                             (x + 0 = x exactly, a != 0; pi + a = a + pi)
//     flip ? -r : r          =  r with its sign flipped when re and im are both negative
// The IEEE division is the compiler's own sequence (v_div_scale x 2, v_rcp, fma x 2, mul, fma x 3, v_div_fmas, v_div_fixup) written
// out so that its fma steps pack.  Same operations on the same operands in the same order: same bits
// (tests/test_gpu_stream.py: every form on arbitrary bit patterns and dense argument ranges).
// ---------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f2v div_ieee_pair(f2v num, f2v den)
{
    bool fdx, fdy, fnx, fny;
    const f2v ds = f2v{__builtin_amdgcn_div_scalef(num.x, den.x, false, &fdx), __builtin_amdgcn_div_scalef(num.y, den.y, false, &fdy)};
    const f2v ns = f2v{__builtin_amdgcn_div_scalef(num.x, den.x, true, &fnx), __builtin_amdgcn_div_scalef(num.y, den.y, true, &fny)};
    const f2v r0 = f2v{__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
    const f2v one = f2v{1.0f, 1.0f};
    const f2v e0 = pk_fma(-ds, r0, one);
    const f2v r1 = pk_fma(e0, r0, r0);
    const f2v q0 = ns * r1;
    const f2v e1 = pk_fma(-ds, q0, ns);
    const f2v q1 = pk_fma(e1, r1, q0);
    const f2v e2 = pk_fma(-ds, q1, ns);
    const float qx = __builtin_amdgcn_div_fmasf(e2.x, r1.x, q1.x, fnx), qy = __builtin_amdgcn_div_fmasf(e2.y, r1.y, q1.y, fny);
    return f2v{__builtin_amdgcn_div_fixupf(qx, den.x, num.x), __builtin_amdgcn_div_fixupf(qy, den.y, num.y)};
}

__device__ __forceinline__ f2v div_unscaled_pair(f2v num, f2v den)
{
    const f2v r0 = f2v{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f2v e0 = pk_fma(-den, r0, f2v{1.0f, 1.0f});
    const f2v r1 = pk_fma(e0, r0, r0);
    const f2v q0 = num * r1;
    const f2v e1 = pk_fma(-den, q0, num);
    const f2v q1 = pk_fma(e1, r1, q0);
    const f2v e2 = pk_fma(-den, q1, num);
    return pk_fma(e2, r1, q1);
}

// (re, im) of cur * conj(prev) with Data.Complex's operations (Demod.hs:28): re = cx*px - cy*(-py), im = cx*(-py) + cy*px
__device__ __forceinline__ f2v conj_product(float2 cur, float2 prev)
{
    const f2v C = f2v{cur.x, cur.y}, P = f2v{prev.x, prev.y};
    const f2v t1 = f2v{C.x, C.x} * f2v{P.x, -P.y};          // (cx*px, cx*nd)
    const f2v t2 = f2v{C.y, C.y} * f2v{-P.y, P.x};          // (cy*nd, cy*px)
    return t1 + f2v{-t2.x, t2.y};                           // (cx*px - cy*nd, cx*nd + cy*px)
}

__device__ __forceinline__ float2 fm_phase_common_tbl2(float2 cur0, float2 prev0, float2 cur1, float2 prev1, bool& rare, const float* tbl)
{
    const uint32_t pi_bits = 0x40490fdbu;                    // 3.14159274101257324f
    const f2v z0 = conj_product(cur0, prev0), z1 = conj_product(cur1, prev1);
    const uint32_t re0 = __float_as_uint(z0.x), im0 = __float_as_uint(z0.y), re1 = __float_as_uint(z1.x), im1 = __float_as_uint(z1.y);
    const uint32_t s0 = re0 & 0x80000000u, s1 = re1 & 0x80000000u;                 // re negative
    const f2v yy = f2v{__uint_as_float(im0 & ~s0), __uint_as_float(im1 & ~s1)};
    const f2v q = div_ieee_pair(yy, f2v{z0.x, z1.x});
    const uint32_t ix0 = __float_as_uint(q.x) & 0x7fffffffu, ix1 = __float_as_uint(q.y) & 0x7fffffffu;
    rare = ((ix0 - 0x31000000u) >= (0x4c000000u - 0x31000000u)) | ((ix1 - 0x31000000u) >= (0x4c000000u - 0x31000000u));
    const uint32_t tq0 = ix0 >> 18, tq1 = ix1 >> 18;
    const uint32_t tc0 = tq0 < 0xfb7u ? 0xfb7u : tq0 > 0x1007u ? 0x1007u : tq0;    // v_med3_u32
    const uint32_t tc1 = tq1 < 0xfb7u ? 0xfb7u : tq1 > 0x1007u ? 0x1007u : tq1;
    const float4 abcd0 = reinterpret_cast<const float4*>(tbl)[tc0 - 0xfb7u], abcd1 = reinterpret_cast<const float4*>(tbl)[tc1 - 0xfb7u];
    const float2 hl0 = reinterpret_cast<const float2*>(tbl + 4 * kAtanRows)[tc0 - 0xfb7u];
    const float2 hl1 = reinterpret_cast<const float2*>(tbl + 4 * kAtanRows)[tc1 - 0xfb7u];
    const float ax0 = __uint_as_float(ix0), ax1 = __uint_as_float(ix1);
    // (the table values arrive in the registers their LDS reads name, so these stay scalar: a packed operand would need moves)
    const f2v num = f2v{abcd0.x * ax0 + abcd0.y, abcd1.x * ax1 + abcd1.y};
    const f2v den = f2v{abcd0.z * ax0 + abcd0.w, abcd1.z * ax1 + abcd1.w};
    const f2v xr = div_unscaled_pair(num, den);
    const f2v z = xr * xr;
    const f2v w = z * z;
    auto K = [](float c) { return f2v{c, c}; };
    const f2v s1v = z * (K(3.3333334327e-01f) + w * (K(1.4285714924e-01f) + w * (K(9.0908870101e-02f) + w * (K(6.6610731184e-02f) + w * (K(4.9768779427e-02f) + w * K(1.6285819933e-02f))))));
    const f2v s2v = w * (K(-2.0000000298e-01f) + w * (K(-1.1111110449e-01f) + w * (K(-7.6918758452e-02f) + w * (K(-5.8335702866e-02f) + w * K(-3.6531571299e-02f)))));
    const f2v t = xr * (s1v + s2v);
    const float zz0 = hl0.x - ((t.x - hl0.y) - xr.x), zz1 = hl1.x - ((t.y - hl1.y) - xr.y);
    const f2v a = f2v{__builtin_copysignf(zz0, q.x), __builtin_copysignf(zz1, q.y)};
    // pi where re is negative, +0 elsewhere: an arithmetic shift spreads re's sign over the word
    const f2v piz = f2v{__uint_as_float((uint32_t)((int32_t)re0 >> 31) & pi_bits), __uint_as_float((uint32_t)((int32_t)re1 >> 31) & pi_bits)};
    const f2v r = a + piz;
    return make_float2(__uint_as_float(__float_as_uint(r.x) ^ (s0 & im0)), __uint_as_float(__float_as_uint(r.y) ^ (s1 & im1)));
}

// N consecutive phases y[e] = phase(v[e + 1] * conj v[e]) the voted way: the common case for every lane, the full select form for
// the wave if any lane holds anything else (the vote is over the lanes active at the call, so it may sit inside divergent code).
template <int N>
__device__ __forceinline__ void fm_phase_voted(const float2 (&v)[N + 1], float (&y)[N])
{
    bool rare = false;
#pragma unroll
    for (int e = 0; e < N; e++) {
        bool q;
        y[e] = fm_phase_common(v[e + 1], v[e], q);
        rare |= q;
    }
    if (__any(rare)) {
#pragma unroll
        for (int e = 0; e < N; e++) y[e] = fm_phase_sel(v[e + 1], v[e]);
    }
}

}  // namespace sdrhip
