// demod_forms.hip -- the stand-alone fmDemod kernel in each of the five restatements of its arithmetic measured in rounds 1-5
// (Demod.hs:21-46 + GHC base atan2 + fdlibm atanf; demod_forms.hpp).  The product keeps form 3.  Per 2^26 samples on MI355X:
// 0 nested ternaries 0.172 ms, 1 selects 0.183, 2 common case + wave vote 0.161-0.164, 3 the same with atanf's reduction as an LDS
// table 0.141, 4 packed pairs (slower than 3 inside the fused loader: 0.207 against 0.198 ms).  All: the same bits (check_lab.py).
#include "lab.hpp"
#include "demod_forms.hpp"

namespace sdrhip {

namespace {

// FORM 3 (the default): form 2 with atanf's argument reduction looked up in an LDS table (demod.hpp: fm_phase_common_tbl; the fused
// loader's form): 0.141 ms per 2^26 samples against 0.164 for form 2.
// FORM 0: nested ternaries (control flow per argument range; rounds 1-3's stand-alone form); 1: selects; 2: the common-case form
// with a wave vote and the select form behind it (round 4: per 2^26
// samples 0.161 ms against 0.172 for the ternaries and 0.183 for the selects).  All three: same bits
// (tests/test_gpu_stream.py::test_fm_demod_random_bit_patterns runs every form over arbitrary bit patterns).
template <int FORM>
__device__ __forceinline__ float4 fm_phase_quad(float2 prev, float2 s0, float2 s1, float2 s2, float2 s3, const float* atbl)
{
    float4 r;
    if constexpr (FORM == 4) {
        // the packed pair form (demod.hpp: fm_phase_common_tbl2; round 5, measured slower, kept under the same tests)
        bool q0, q1;
        const float2 a = fm_phase_common_tbl2(s0, prev, s1, s0, q0, atbl), b = fm_phase_common_tbl2(s2, s1, s3, s2, q1, atbl);
        r = make_float4(a.x, a.y, b.x, b.y);
        if (__any(q0 | q1)) r = make_float4(fm_phase_sel(s0, prev), fm_phase_sel(s1, s0), fm_phase_sel(s2, s1), fm_phase_sel(s3, s2));
    } else if constexpr (FORM == 3) {
        const float2 v[5] = {prev, s0, s1, s2, s3};
        float y[4];
        bool rare = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            bool q;
            y[e] = fm_phase_common_tbl(v[e + 1], v[e], q, atbl);
            rare |= q;
        }
        if (__any(rare)) {
#pragma unroll
            for (int e = 0; e < 4; e++) y[e] = fm_phase_sel(v[e + 1], v[e]);
        }
        r = make_float4(y[0], y[1], y[2], y[3]);
    } else if constexpr (FORM == 0) {
        r.x = fm_phase_tern(s0, prev); r.y = fm_phase_tern(s1, s0); r.z = fm_phase_tern(s2, s1); r.w = fm_phase_tern(s3, s2);
    } else if constexpr (FORM == 1) {
        r.x = fm_phase_sel(s0, prev); r.y = fm_phase_sel(s1, s0); r.z = fm_phase_sel(s2, s1); r.w = fm_phase_sel(s3, s2);
    } else {
        const float2 v[5] = {prev, s0, s1, s2, s3};
        float y[4];
        fm_phase_voted<4>(v, y);
        r = make_float4(y[0], y[1], y[2], y[3]);
    }
    return r;
}

template <int FORM>
__global__ void __launch_bounds__(256) k_fm_demod_fast(const float* __restrict__ in, float* __restrict__ out, int64_t count,
                                                        int has_prev, float last_re, float last_im, int out_vec)
{
    // 4 samples per thread: five 8-byte loads (the IQ stream is only 8-byte aligned in
    // general: it usually starts one sample into a buffer), one 16-byte store
    const float2* in2 = reinterpret_cast<const float2*>(in);
    const int64_t nquad = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    __shared__ __attribute__((aligned(16))) float atbl[FORM >= 3 ? kAtanRows * kAtanRowFloats : 4];
    if constexpr (FORM >= 3) {
        atan_table_fill(atbl, threadIdx.x);
        __syncthreads();
    }
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        // (round 4, measured and not kept: 16-byte loads for interior quads 0.164 ms against 0.159 for these five 8-byte loads per
        // 2^26 samples, non-temporal 0.185 -- the decimator's output is still partly in the last-level cache when this kernel reads it)
        const float2 s0 = in2[4 * q], s1 = in2[4 * q + 1], s2 = in2[4 * q + 2], s3 = in2[4 * q + 3];
        float2 prev;
        if (q > 0 || has_prev) prev = in2[4 * q - 1];
        else prev = make_float2(last_re, last_im);
        const float4 r = fm_phase_quad<FORM>(prev, s0, s1, s2, s3, atbl);
        if (out_vec) {
            reinterpret_cast<float4*>(out)[q] = r;
        } else {
            out[4 * q] = r.x; out[4 * q + 1] = r.y; out[4 * q + 2] = r.z; out[4 * q + 3] = r.w;
        }
    }
    // tail (< 4 samples)
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t i = (nquad << 2) + threadIdx.x;
        const float2 cur = in2[i];
        float2 prev;
        if (i > 0 || has_prev) prev = in2[i - 1];
        else prev = make_float2(last_re, last_im);
        out[i] = fm_phase_tern(cur, prev);
    }
}


}  // namespace

void launch_fm_demod_form(hipStream_t s, int form, const float* d_in_iq, float* d_out, int64_t count, bool has_prev, float last_re, float last_im)
{
    if (count <= 0) return;
    const int out_vec = ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0) ? 1 : 0;
    int64_t blocks = ((count >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 64) blocks = 256 * 64;
    auto k = form == 4 ? k_fm_demod_fast<4> : form == 3 ? k_fm_demod_fast<3> : form == 2 ? k_fm_demod_fast<2> : form == 1 ? k_fm_demod_fast<1> : k_fm_demod_fast<0>;
    hipLaunchKernelGGL(k, dim3((int)blocks), dim3(256), 0, s, d_in_iq, d_out, count, has_prev ? 1 : 0, last_re, last_im, out_vec);
}

}  // namespace sdrhip
