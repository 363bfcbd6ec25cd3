/*
 * sdr_oracle.c -- CPU restatement of the reference's FIR / decimate / resample /
 * convert / FM-demod arithmetic.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker / the
 * reported CPU baseline.  Nothing under sdr_amd/ links, imports or calls it.
 *
 * The reference (adamwalker/sdr, /root/reference) evaluates every FIR output
 * as an f32 dot product in a fixed SIMD lane order with separate multiply and
 * add (no FMA).  This file restates that order in plain C: `L` strided partial
 * sums accumulated in increasing tap order from +0.0f, reduced by the same
 * pairwise tree the SSE/AVX horizontal adds perform.  It must be compiled with
 * -ffp-contract=off (see oracle/Makefile); the loops are written so gcc can
 * vectorise them without changing any rounding.
 *
 * Parity pin: oracle/_ref/libsdr_ref.so (the reference's own c_sources built
 * unmodified by oracle/Makefile) -- tests/test_oracle_vs_ref.py checks every
 * C-backed function here bit-for-bit against it, and tests/golden/ holds
 * outputs generated from it.  The Haskell-only pieces (cross-buffer kernels,
 * prepareCoeffs, fmDemod) cannot be executed in this image (no GHC):
 * they are restated from the cited lines; fmDemod additionally depends on GHC
 * `base` + host libm atanf => "parity unpinned" for orc_fm_demod.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <gnu/libc-version.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------------- *
 * Horizontal reductions.
 *   L=1 : scalar                     (common.h:34-41 dotprod_R)
 *   L=4 : (a0+a1)+(a2+a3)            (common.h:12-16 sse_hadd_R)
 *   L=8 : ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7))   (common.h:18-29 avx_hadd_R)
 * ------------------------------------------------------------------------- */
static inline float hadd_r(const float *a, int L)
{
    if (L == 1) return a[0];
    if (L == 4) return (a[0] + a[1]) + (a[2] + a[3]);
    /* L == 8 */
    float lo = (a[0] + a[1]) + (a[2] + a[3]);
    float hi = (a[4] + a[5]) + (a[6] + a[7]);
    return lo + hi;
}

/* Complex data viewed as interleaved floats with duplicated taps: the real
 * dot product runs over 2P floats, float lane f = 2*k + comp, complex lane k.
 *   CL=1 : scalar                                  (common.h:95-106 dotprod_C)
 *   CL=2 : l0 + l1 per component                   (common.h:77-80 sse_hadd_C)
 *   CL=4 : (l0+l1)+(l2+l3) per component           (common.h:82-90 avx_hadd_C)
 * avx_hadd_C: permute -> [a0 a2 a1 a3 | a4 a6 a5 a7]; hadd(lo,hi) ->
 * [a0+a2, a1+a3, a4+a6, a5+a7]; permute -> [a0+a2, a4+a6, a1+a3, a5+a7];
 * hadd -> [(a0+a2)+(a4+a6), (a1+a3)+(a5+a7)] where a_{2k+c} is complex lane k.
 */
static inline void hadd_c(const float *a /* 2*CL floats, interleaved */, int CL, float *out)
{
    if (CL == 1) { out[0] = a[0]; out[1] = a[1]; return; }
    if (CL == 2) { out[0] = a[0] + a[2]; out[1] = a[1] + a[3]; return; }
    out[0] = (a[0] + a[2]) + (a[4] + a[6]);
    out[1] = (a[1] + a[3]) + (a[5] + a[7]);
}

/* Strided partial sums: acc[l] += c[i+l]*x[i+l], i += L.  n must be a multiple
 * of L (the reference pads the taps so that it is).  common.h:43-72. */
static inline void dot_lanes(int n, const float *c, const float *x, int L, float *acc)
{
    for (int l = 0; l < L; l++) acc[l] = 0.0f;
    if (L == 8) {
        for (int i = 0; i < n; i += 8)
            for (int l = 0; l < 8; l++) acc[l] = acc[l] + c[i + l] * x[i + l];
    } else if (L == 4) {
        for (int i = 0; i < n; i += 4)
            for (int l = 0; l < 4; l++) acc[l] = acc[l] + c[i + l] * x[i + l];
    } else {
        for (int i = 0; i < n; i++) acc[0] = acc[0] + c[i] * x[i];
    }
}

/* Symmetric: acc[l] += c[i+l] * (x[i+l] + x[2n-1-i-l]).  common.h:161-201. */
static inline void sym_dot_lanes(int n, const float *c, const float *x, int L, float *acc)
{
    const float *e = x + 2 * n - 1;
    for (int l = 0; l < L; l++) acc[l] = 0.0f;
    if (L == 8) {
        for (int i = 0; i < n; i += 8)
            for (int l = 0; l < 8; l++) acc[l] = acc[l] + c[i + l] * (x[i + l] + e[-(i + l)]);
    } else {
        for (int i = 0; i < n; i += 4)
            for (int l = 0; l < 4; l++) acc[l] = acc[l] + c[i + l] * (x[i + l] + e[-(i + l)]);
    }
}

/* ------------------------------------------------------------------------- *
 * A1  u8 -> f32.  convert.c:15-50 (convertC / convertCSSE / convertCAVX are
 * bit-identical: every result is exactly representable).  `num` = bytes.
 * ------------------------------------------------------------------------- */
void orc_convert_u8(int num, const uint8_t *in, float *out)
{
    for (int i = 0; i < num; i++) out[i] = ((float)in[i] - 128.0f) * (1.0f / 128.0f);
}

/* convert.c:52-85 BladeRF i16 -> f32 (x * 1/2048). */
void orc_convert_i16(int num, const int16_t *in, float *out)
{
    for (int i = 0; i < num; i++) out[i] = (float)in[i] * (1.0f / 2048.0f);
}

/* scale.c:15-36 */
void orc_scale(int num, float factor, const float *in, float *out)
{
    for (int i = 0; i < num; i++) out[i] = in[i] * factor;
}

/* ------------------------------------------------------------------------- *
 * Real FIR.  filter.c:16-46 (filterRR L=1, filterSSERR L=4, filterAVXRR L=8)
 * ------------------------------------------------------------------------- */
void orc_filter_rr(int L, int num, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    float acc[8];
    for (int o = 0; o < num; o++) {
        dot_lanes(numCoeffs, coeffs, in + o, L, acc);
        out[o] = hadd_r(acc, L);
    }
}

/* filter.c:50-68 (filterSSESymmetricRR L=4, filterAVXSymmetricRR L=8).
 * numCoeffs = HALF length n; the filter has 2n taps. */
void orc_filter_sym_rr(int L, int num, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    float acc[8];
    for (int o = 0; o < num; o++) {
        sym_dot_lanes(numCoeffs, coeffs, in + o, L, acc);
        out[o] = hadd_r(acc, L);
    }
}

/* decimate.c:16-46 (decimateRR / SSERR / AVXRR) */
void orc_decimate_rr(int L, int num, int factor, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    float acc[8];
    for (int o = 0; o < num; o++) {
        dot_lanes(numCoeffs, coeffs, in + (size_t)o * factor, L, acc);
        out[o] = hadd_r(acc, L);
    }
}

/* decimate.c:51-68 (decimateSSESymmetricRR / AVXSymmetricRR) */
void orc_decimate_sym_rr(int L, int num, int factor, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    float acc[8];
    for (int o = 0; o < num; o++) {
        sym_dot_lanes(numCoeffs, coeffs, in + (size_t)o * factor, L, acc);
        out[o] = hadd_r(acc, L);
    }
}

/* ------------------------------------------------------------------------- *
 * Real taps, complex data.
 *  CL=1: filterRC / decimateRC (filter.c:73-79, decimate.c:73-79): plain taps,
 *        numCoeffs = P taps, sequential.
 *  CL=2: filterSSERC / decimateSSERC (filter.c:85-92, decimate.c:84-92),
 *  CL=4: filterAVXRC / decimateAVXRC (filter.c:106-114, decimate.c:105-113):
 *        DUPLICATED taps [h0,h0,h1,h1,..], numCoeffs = 2P floats.
 * `factor` = 1 gives the filter.
 * ------------------------------------------------------------------------- */
void orc_decimate_rc(int CL, int num, int factor, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    float acc[8];
    for (int o = 0; o < num; o++) {
        const float *x = in + (size_t)2 * o * factor;
        if (CL == 1) {
            float re = 0.0f, im = 0.0f;
            for (int i = 0; i < numCoeffs; i++) {
                re = re + x[2 * i] * coeffs[i];
                im = im + x[2 * i + 1] * coeffs[i];
            }
            out[2 * o] = re;
            out[2 * o + 1] = im;
        } else {
            dot_lanes(numCoeffs, coeffs, x, 2 * CL, acc);
            hadd_c(acc, CL, out + 2 * o);
        }
    }
}

void orc_filter_rc(int CL, int num, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    orc_decimate_rc(CL, num, 1, numCoeffs, coeffs, in, out);
}

/* "RC2" order (plain taps splatted in-register, two accumulators):
 *  common.h:108-127 sse_dotprod_C, :129-155 avx_dotprod_C.
 *  AVX: per 8-tap iteration accum1 gets taps i..i+3 (complex lanes 0..3),
 *  accum2 gets taps i+4..i+7; so there are 8 complex partials p_m over taps
 *  m, m+8, ...; q_k = p_k + p_{k+4}; result (q0+q1)+(q2+q3) via avx_hadd_C.
 *  SSE: 4 complex partials p_m over taps m, m+4,..; q_k = p_k + p_{k+2};
 *  result q0+q1.
 *  Used by filter{SSE,AVX}RC2, decimate{SSE,AVX}RC2, resample{SSE,AVX}RC.
 *  numCoeffs = P plain taps (multiple of 2*CL). */
static inline void dot_c2(int CL, int n, const float *c, const float *x, float *out)
{
    float p[16];
    int M = 2 * CL; /* complex partials */
    for (int k = 0; k < 2 * M; k++) p[k] = 0.0f;
    for (int i = 0; i < n; i += M)
        for (int m = 0; m < M; m++) {
            p[2 * m]     = p[2 * m]     + c[i + m] * x[2 * (i + m)];
            p[2 * m + 1] = p[2 * m + 1] + c[i + m] * x[2 * (i + m) + 1];
        }
    float q[8];
    for (int k = 0; k < CL; k++) {
        q[2 * k]     = p[2 * k]     + p[2 * (k + CL)];
        q[2 * k + 1] = p[2 * k + 1] + p[2 * (k + CL) + 1];
    }
    hadd_c(q, CL, out);
}

void orc_decimate_rc2(int CL, int num, int factor, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    for (int o = 0; o < num; o++)
        dot_c2(CL, numCoeffs, coeffs, in + (size_t)2 * o * factor, out + 2 * o);
}

/* Symmetric complex: common.h:206-268.  Half taps c[0..n); element i pairs
 * x[i] with x[2n-1-i] (complex add first), then the RC2 partial structure. */
void orc_decimate_sym_rc(int CL, int num, int factor, int numCoeffs, const float *coeffs, const float *in, float *out)
{
    int n = numCoeffs, M = 2 * CL;
    for (int o = 0; o < num; o++) {
        const float *x = in + (size_t)2 * o * factor;
        float p[16];
        for (int k = 0; k < 2 * M; k++) p[k] = 0.0f;
        for (int i = 0; i < n; i += M)
            for (int m = 0; m < M; m++) {
                int a = i + m, b = 2 * n - 1 - a;
                p[2 * m]     = p[2 * m]     + coeffs[a] * (x[2 * a]     + x[2 * b]);
                p[2 * m + 1] = p[2 * m + 1] + coeffs[a] * (x[2 * a + 1] + x[2 * b + 1]);
            }
        float q[8];
        for (int k = 0; k < CL; k++) {
            q[2 * k]     = p[2 * k]     + p[2 * (k + CL)];
            q[2 * k + 1] = p[2 * k + 1] + p[2 * (k + CL) + 1];
        }
        hadd_c(q, CL, out + 2 * o);
    }
}

/* ------------------------------------------------------------------------- *
 * A3  polyphase resampler.  resample.c:34-87 (resample2RR L=1, resampleSSERR
 * L=4, resampleAVXRR L=8).  num_coeffs is the UNPADDED max group length; the
 * SIMD loops step by L and so read the zero padding (resample.c:58,76 via
 * common.h:47,62).  Returns the end group.
 * ------------------------------------------------------------------------- */
int orc_resample_rr(int L, int buf_size, int num_coeffs, int starting_group, int num_groups,
                    const int *increments, float *const *coeffs, const float *in, float *out)
{
    float acc[8];
    int group = starting_group;
    const float *p = in;
    int n = ((num_coeffs + L - 1) / L) * L;
    for (int i = 0; i < buf_size; i++) {
        dot_lanes(n, coeffs[group], p, L, acc);
        out[i] = hadd_r(acc, L);
        p += increments[group];
        group++;
        if (group == num_groups) group = 0;
    }
    return group;
}

/* resample.c:89-142 (resample2RC CL=1 sequential; resampleSSERC CL=2 and
 * resampleAVXRC CL=4 use the sse/avx_dotprod_C "RC2" order). */
int orc_resample_rc(int CL, int buf_size, int num_coeffs, int starting_group, int num_groups,
                    const int *increments, float *const *coeffs, const float *in, float *out)
{
    int group = starting_group;
    const float *p = in;
    int M = 2 * CL;
    int n = (CL == 1) ? num_coeffs : ((num_coeffs + M - 1) / M) * M;
    for (int i = 0; i < buf_size; i++) {
        if (CL == 1) {
            float re = 0.0f, im = 0.0f;
            for (int j = 0; j < n; j++) {
                re = re + p[2 * j] * coeffs[group][j];
                im = im + p[2 * j + 1] * coeffs[group][j];
            }
            out[2 * i] = re;
            out[2 * i + 1] = im;
        } else {
            dot_c2(CL, n, coeffs[group], p, out + 2 * i);
        }
        p += 2 * increments[group];
        group++;
        if (group == num_groups) group = 0;
    }
    return group;
}

/* Legacy single-array resampler, resample.c:16-32 (sequential order). */
void orc_resample_legacy_rr(int buf_size, int coeff_size, int interpolation, int decimation,
                            int filter_offset, const float *coeffs, const float *in, float *out)
{
    int input_offset = 0;
    for (int k = 0; k < buf_size; k++) {
        float accum = 0.0f;
        for (int l = 0, j = filter_offset; j < coeff_size; l++, j += interpolation)
            accum = accum + in[input_offset + l] * coeffs[j];
        int fo = interpolation - 1 - (decimation - filter_offset - 1) % interpolation;
        input_offset += (decimation - filter_offset - 1) / interpolation + 1;
        filter_offset = fo;
        out[k] = accum;
    }
}

/* ------------------------------------------------------------------------- *
 * A10  prepareCoeffs.  FilterInternal.hs:277-319.
 * Walk filter offsets off0=0, off' = I-1-((D-off-1) mod I) until 0 recurs.
 * group g = strideList I (drop off_g coeffs), zero-padded to roundUp(maxLen,n);
 * increments[g] = (D-off_g-1) div I + 1.
 * Outputs: *num_coeffs (unpadded max len), *num_groups, increments[<=I],
 * offsets[<=I], groups (caller buffer of I*roundUp(ceil(ncoeffs/I),n) floats,
 * row stride = *padded_len).  Returns 0.
 * ------------------------------------------------------------------------- */
int orc_prepare_coeffs(int n, int interpolation, int decimation, const float *coeffs, int ncoeffs,
                       int *num_coeffs, int *num_groups, int *padded_len,
                       int *increments, int *offsets, float *groups)
{
    /* the walk visits at most `interpolation` offsets; they go straight into the caller's offsets[] (<= I entries) */
    int *offs = offsets, ng = 0, off = 0, maxlen = 0;
    do {
        int len = (ncoeffs - off + interpolation - 1) / interpolation;
        if (ncoeffs - off <= 0) len = 0;
        if (len > maxlen) maxlen = len;
        offs[ng] = off;
        increments[ng] = (decimation - off - 1) / interpolation + 1;
        int r = (decimation - off - 1) % interpolation;
        off = interpolation - 1 - r;
        ng++;
    } while (off != 0 && ng < interpolation);
    int pl = ((maxlen + n - 1) / n) * n;
    for (int g = 0; g < ng; g++) {
        float *row = groups + (size_t)g * pl;
        int j = 0;
        for (int i = offs[g]; i < ncoeffs; i += interpolation) row[j++] = coeffs[i];
        for (; j < pl; j++) row[j] = 0.0f;
    }
    *num_coeffs = maxlen;
    *num_groups = ng;
    *padded_len = pl;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * A6-A8  cross-buffer kernels (pure Haskell in the reference): sequential
 * left fold from 0 (`VG.sum`) of data*coeff over (drop i last ++ next).
 * FilterInternal.hs:397-423.  `last`/`next` lengths in elements.
 * ------------------------------------------------------------------------- */
static inline float xr(const float *last, int nlast, const float *next, int idx)
{
    return idx < nlast ? last[idx] : next[idx - nlast];
}

/* FilterInternal.hs:397-402 (real).  filterCrossHighLevel (:404-408) is factor=1. */
void orc_decimate_cross_r(int factor, int ncoeffs, const float *coeffs, int num,
                          const float *last, int nlast, const float *next, float *out)
{
    for (int o = 0; o < num; o++) {
        int i = o * factor;
        float s = 0.0f;
        for (int j = 0; j < ncoeffs; j++) s = s + xr(last, nlast, next, i + j) * coeffs[j];
        out[o] = s;
    }
}

/* complex data via Mult (Util.hs:87-88): (x:+y) `mult` z = (x*z):+(y*z) */
void orc_decimate_cross_c(int factor, int ncoeffs, const float *coeffs, int num,
                          const float *last, int nlast, const float *next, float *out)
{
    for (int o = 0; o < num; o++) {
        int i = o * factor;
        float re = 0.0f, im = 0.0f;
        for (int j = 0; j < ncoeffs; j++) {
            re = re + xr(last, 2 * nlast, next, 2 * (i + j)) * coeffs[j];
            im = im + xr(last, 2 * nlast, next, 2 * (i + j) + 1) * coeffs[j];
        }
        out[2 * o] = re;
        out[2 * o + 1] = im;
    }
}

/* FilterInternal.hs:410-423: taps = stride I (drop filterOffset coeffs) over the
 * UNPADDED coefficient list; phase recurrence :418-420.  Returns end offset. */
int orc_resample_cross_r(int interpolation, int decimation, int ncoeffs, const float *coeffs,
                         int filter_offset, int count,
                         const float *last, int nlast, const float *next, float *out)
{
    int input_offset = 0;
    for (int i = 0; i < count; i++) {
        float s = 0.0f;
        for (int l = 0, j = filter_offset; j < ncoeffs; l++, j += interpolation)
            s = s + xr(last, nlast, next, input_offset + l) * coeffs[j];
        out[i] = s;
        int q = (decimation - filter_offset - 1) / interpolation;
        int r = (decimation - filter_offset - 1) % interpolation;
        input_offset += q + 1;
        filter_offset = interpolation - 1 - r;
    }
    return filter_offset;
}

int orc_resample_cross_c(int interpolation, int decimation, int ncoeffs, const float *coeffs,
                         int filter_offset, int count,
                         const float *last, int nlast, const float *next, float *out)
{
    int input_offset = 0;
    for (int i = 0; i < count; i++) {
        float re = 0.0f, im = 0.0f;
        for (int l = 0, j = filter_offset; j < ncoeffs; l++, j += interpolation) {
            re = re + xr(last, 2 * nlast, next, 2 * (input_offset + l)) * coeffs[j];
            im = im + xr(last, 2 * nlast, next, 2 * (input_offset + l) + 1) * coeffs[j];
        }
        out[2 * i] = re;
        out[2 * i + 1] = im;
        int q = (decimation - filter_offset - 1) / interpolation;
        int r = (decimation - filter_offset - 1) % interpolation;
        input_offset += q + 1;
        filter_offset = interpolation - 1 - r;
    }
    return filter_offset;
}

/* ------------------------------------------------------------------------- *
 * A5  fmDemod.  Demod.hs:21-46: y[n] = phase(x[n] * conjugate x[n-1]).
 * Arithmetic lives in GHC base (Data.Complex, RealFloat atan2 default) and the
 * host libm atanf; restated from SURVEY.md Appendix C.  PARITY UNPINNED: the
 * reference has no test or vector for it and GHC cannot be run here.
 *
 * THE SPEC IS THE MODEL (round 5): `atan` for Float is evaluated by orc_atanf_model below -- the fdlibm / glibc s_atanf.c
 * algorithm in plain f32 arithmetic -- NOT by whatever atanf the machine running the tests links.  glibc 2.35's atanf is that
 * algorithm and agrees with the model on all 2^32 inputs (swept here; tests/test_oracle_demod.py re-checks it, exactly when
 * gnu_get_libc_version() says 2.35 and to 1 ULP otherwise: a correctly-rounded atanf differs from fdlibm's in ~5 % of arguments).
 * So "the reference" for fmDemod means GHC's formulas over fdlibm's atanf, on every box.
 * ------------------------------------------------------------------------- */
static const float ORC_PI = 3.14159274101257324f; /* f32 pi 0x40490FDB */

/* fdlibm-style f32 atan restated in plain f32 arithmetic (sysdeps/ieee754/flt-32/s_atanf.c; what the device kernels evaluate) */
static const float atanhi_[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float atanlo_[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float aT_[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                              9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                              4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};

float orc_atanf_model(float x)
{
    uint32_t hx;
    memcpy(&hx, &x, 4);
    uint32_t ix = hx & 0x7fffffffu;
    int neg = (hx >> 31) != 0;
    int id;
    if (ix >= 0x4c000000u) { /* |x| >= 2^25 */
        if (ix > 0x7f800000u) return x + x;
        float r = atanhi_[3] + atanlo_[3];
        return neg ? -r : r;
    }
    if (ix < 0x3ee00000u) { /* |x| < 0.4375 */
        if (ix < 0x31000000u) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                  { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                  { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT_[0] + w * (aT_[2] + w * (aT_[4] + w * (aT_[6] + w * (aT_[8] + w * aT_[10])))));
    float s2 = w * (aT_[1] + w * (aT_[3] + w * (aT_[5] + w * (aT_[7] + w * aT_[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi_[id] - ((x * (s1 + s2) - atanlo_[id]) - x);
    return neg ? -z : z;
}

static float ghc_atan2f(float y, float x)
{
    if (x > 0.0f) return orc_atanf_model(y / x);
    if (x == 0.0f && y > 0.0f) return ORC_PI / 2.0f;
    if (x < 0.0f && y > 0.0f) return ORC_PI + orc_atanf_model(y / x);
    if ((x <= 0.0f && y < 0.0f) || (x < 0.0f && y == 0.0f && signbit(y)) ||
        (x == 0.0f && signbit(x) && y == 0.0f && signbit(y)))
        return -ghc_atan2f(-y, x);
    if (y == 0.0f && (x < 0.0f || (x == 0.0f && signbit(x)))) return ORC_PI;
    if (x == 0.0f && y == 0.0f) return y;
    return x + y;
}

static inline float ghc_phase(float re, float im)
{
    if (re == 0.0f && im == 0.0f) return 0.0f; /* phase (0:+0) = 0, matches -0 too */
    return ghc_atan2f(im, re);
}

/* one step of Demod.hs:28: sample * conjugate last, with conjugate (c:+d) = c:+(-d) and
 * (a:+b)*(c:+d') = (a*c - b*d') :+ (a*d' + b*c) */
static inline float fm_step(float a, float b, float c, float d)
{
    float nd = -d;
    float re = a * c - b * nd;
    float im = a * nd + b * c;
    return ghc_phase(re, im);
}

/* in: num complex samples; (last_re,last_im) = sample preceding in[0]. */
void orc_fm_demod(int num, float last_re, float last_im, const float *in, float *out)
{
    float c = last_re, d = last_im;
    for (int i = 0; i < num; i++) {
        float a = in[2 * i], b = in[2 * i + 1];
        out[i] = fm_step(a, b, c, d);
        c = a;
        d = b;
    }
}

/* exposed for the atanf cross-check (tests) */
float orc_libm_atanf(float x) { return atanf(x); }
float orc_ghc_atan2f(float y, float x) { return ghc_atan2f(y, x); }
const char *orc_libc_version(void) { return gnu_get_libc_version(); }

static inline uint32_t ulp_distance(float a, float b)
{
    /* distance in representable floats (sign-magnitude -> monotone integer); NaN vs NaN = 0 */
    if (a != a && b != b) return 0;
    if (a != a || b != b) return 0xffffffffu;
    int32_t ia, ib;
    memcpy(&ia, &a, 4);
    memcpy(&ib, &b, 4);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    int64_t d = (int64_t)ia - (int64_t)ib;
    return (uint32_t)(d < 0 ? -d : d);
}

/* Sweep helper: the model against THIS machine's libm atanf over a range of bit patterns: number of arguments where the two differ,
 * the first such argument and the largest distance in ULP.  (A separate, version-gated statement since round 5; threaded.) */
uint64_t orc_atanf_sweep(uint32_t lo, uint32_t hi, uint32_t step, uint32_t *first_bad, uint32_t *max_ulp)
{
    uint64_t bad = 0;
    uint32_t worst = 0, first = 0xffffffffu;
    const int64_t n = ((int64_t)hi - (int64_t)lo) / step + 1;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(max : worst) reduction(min : first)
    for (int64_t k = 0; k < n; k++) {
        uint32_t b = (uint32_t)((uint64_t)lo + (uint64_t)k * step);
        float x, a, m;
        memcpy(&x, &b, 4);
        a = atanf(x);
        m = orc_atanf_model(x);
        uint32_t u = ulp_distance(a, m);
        if (u) {
            bad++;
            if (b < first) first = b;
            if (u > worst) worst = u;
        }
    }
    if (first_bad) *first_bad = first;
    if (max_ulp) *max_ulp = worst;
    return bad;
}

/* ---- exhaustive checks of the DEVICE's fmDemod (tests/test_gpu_demod_exhaustive.py) --------------------------------------------
 * The device demodulates a synthetic stream, the checker recomputes every output from the same closed-form input with the functions
 * above (threaded) and counts the outputs whose bits differ (NaN against NaN counts as equal: payloads are not part of the contract).
 * kind 0, "every atanf argument": sample 2i = 1 + 0i, sample 2i+1 = 1 + q_i i with q_i the float whose bit pattern is lo + i -- so
 *         y[2i+1] = atan2(q_i, 1) = atanf(q_i) for finite q_i and y[2i] = atan2(-q_{i-1}, 1): all 2^32 arguments of atanf in both signs.
 * kind 1, "pseudo-random pairs": sample 2i = 1 + 0i, sample 2i+1 = x_i + y_i i with (x_i, y_i) two different multiplicative hashes
 *         of the index lo + i (each a permutation of all 2^32 bit patterns); odd indices get moderate exponents so that not every product
 *         overflows.  y[2i+1] = atan2(y_i (+-0), x_i (+-0)), y[2i] the phase of the conjugate.
 * `got` holds the device's y for samples [2 * (lo - base) ...): n pairs = 2n outputs; the sample before the first is pair lo - 1's second
 * sample (or 0 + 0i when lo == first_index: the stream start, Demod.hs:41). */
static inline void sweep_pair(int kind, uint64_t idx, float *re, float *im)
{
    uint32_t xb, yb;
    if (kind == 0) {
        xb = 0x3f800000u;
        yb = (uint32_t)idx;
    } else {
        xb = (uint32_t)(idx * 0x9E3779B1u + 0x7F4A7C15u);
        yb = (uint32_t)(idx * 0x85EBCA77u + 0xC2B2AE3Du);
        if (idx & 1) {
            xb = (xb & 0x807fffffu) | ((100u + ((xb >> 23) & 0xffu) % 50u) << 23);
            yb = (yb & 0x807fffffu) | ((100u + ((yb >> 23) & 0xffu) % 50u) << 23);
        }
    }
    memcpy(re, &xb, 4);
    memcpy(im, &yb, 4);
}

/* fills iq[4k .. 4k+3] = (1, 0, re_k, im_k) for k = idx0 .. idx0 + n - 1 (the generator's twin of the device-side construction) */
void orc_demod_sweep_fill(int kind, uint64_t idx0, int64_t n, float *iq)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; k++) {
        float re, im;
        sweep_pair(kind, idx0 + (uint64_t)k, &re, &im);
        iq[4 * k] = 1.0f; iq[4 * k + 1] = 0.0f; iq[4 * k + 2] = re; iq[4 * k + 3] = im;
    }
}

uint64_t orc_demod_sweep_check(int kind, uint64_t idx0, int64_t n, int at_stream_start, const float *got, uint64_t *first_bad)
{
    uint64_t bad = 0, first = ~(uint64_t)0;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : first)
    for (int64_t k = 0; k < n; k++) {
        float re, im, pre = 0.0f, pim = 0.0f;
        sweep_pair(kind, idx0 + (uint64_t)k, &re, &im);
        if (k > 0 || !at_stream_start) sweep_pair(kind, idx0 + (uint64_t)k - 1, &pre, &pim);
        const float e0 = fm_step(1.0f, 0.0f, pre, pim), e1 = fm_step(re, im, 1.0f, 0.0f);
        const float g0 = got[2 * k], g1 = got[2 * k + 1];
        uint32_t be0, be1, bg0, bg1;
        memcpy(&be0, &e0, 4); memcpy(&be1, &e1, 4); memcpy(&bg0, &g0, 4); memcpy(&bg1, &g1, 4);
        const int ok0 = be0 == bg0 || (e0 != e0 && g0 != g0), ok1 = be1 == bg1 || (e1 != e1 && g1 != g1);
        if (!ok0 || !ok1) {
            bad += !ok0 + !ok1;
            const uint64_t where = 2 * (idx0 + (uint64_t)k) + (ok0 ? 1 : 0);
            if (where < first) first = where;
        }
    }
    if (first_bad) *first_bad = first;
    return bad;
}

/* dcBlocker, c_sources/filter.c:152-161.  `0.997` is a double constant: the f32 difference and the f32 state are
 * promoted, multiplied/added in f64 and rounded back to f32 by the assignment (x86-64 SSE2, FLT_EVAL_METHOD 0). */
void orc_dc_blocker(int64_t num, float last_sample, float last_output, float *final_sample, float *final_output,
                    const float *in, float *out)
{
    for (int64_t i = 0; i < num; i++) {
        const float d = in[i] - last_sample;
        last_output = (float)((double)d + 0.997 * (double)last_output);
        out[i] = last_output;
        last_sample = in[i];
    }
    *final_sample = last_sample;
    *final_output = last_output;
}
