// kernels_tail.hip -- the low-rate tail of the FM chain as ONE kernel:
//     d[k] (decimator output, cfloat)  --fmDemod-->  y[k]  --firResampler 3/10-->  z[m]  --firFilter(sym) * gain-->  audio[q]
//     Demod.hs:21-46                      resample.c:70-87 (AVX order)             filter.c:60-68 (AVX order), fm.hs:40
// d[k] is read from HBM once, y and z live only in LDS, audio is written once: 8 B read + 1.2 B written per decimator
// output instead of 8+4, 4+1.2, 1.2+1.2 through three kernels and three seam fix-up launches.
//
// One workgroup = a tile of A = 2046 audio outputs [qa, qa + A), qa a multiple of 3 (= one polyphase cycle of the 3/10
// resampler), computed back to front:
//   phase 1  y[k] for every k the tile's resampler windows touch (10*NC + 61 values, NC = 725 cycles)  -> LDS
//   phase 2  z[m] for the NC cycles [qa/3, qa/3 + NC) (3 outputs each: groups 0,1,2, taps wave-uniform)   -> LDS
//   phase 3  audio[q] = gain * sym_fir(z[q .. q+127]), 4 consecutive outputs per thread
// Neighbouring tiles recompute the 127-output overlap of z (and its y): 6-7 % redundant work for no HBM round trip.
//
// Seams.  The reference's Pipes compute the outputs whose window straddles two input buffers in sequential order with
// the un-grouped taps (resampleCrossHighLevel / filterCrossHighLevel, FilterInternal.hs:404-423).  Here that is decided
// per output inside the kernel (the same predicates as the fix-up kernels of kernels_chain.hip): a lane whose output is a
// Cross one recomputes it sequentially from the same LDS tile.  ~0.8 % of z and 1.5 % of audio outputs, clustered, so few
// waves diverge.
#include "demod.hpp"
#include "kernels.hpp"

namespace sdrhip {

namespace {

constexpr int TAIL_NT = 256;
constexpr int TAIL_A = kTailTileOutputs;                              // audio outputs per tile (multiple of 3)
constexpr int TAIL_LF = 128;                              // audio filter length (64 half-taps)
constexpr int TAIL_NC = (TAIL_A + TAIL_LF - 1 + 2) / 3;   // resampler cycles per tile: 725
constexpr int TAIL_NZ = 3 * TAIL_NC;                      // z values per tile
constexpr int TAIL_NL = 64;                               // resampler group length (padded)
constexpr int TAIL_NY = 10 * (TAIL_NC - 1) + 7 + TAIL_NL; // y values per tile: 7311
constexpr int TAIL_NY_PAD = (TAIL_NY + 3 + 8) & ~3;


struct TailParams {
    int64_t kd0, kd1;        // d holds [kd0, kd1)
    int64_t ky0, ky1;        // y values the run may compute: [ky0, ky1)
    int64_t q0, q1;          // audio outputs of this launch
    int row_stride;          // floats between the resampler's group rows
    int ntaps;               // resampler taps (unpadded)
    int rLp;                 // numCoeffsR (192)
    float gain;
    int64_t seam;            // block size B of every Pipe (0 = contiguous stream)
};

// eight wave-uniform taps by ONE scalar load (constant address space + an SGPR address the compiler cannot see through:
// no hoisting out of the surrounding loops, no vector loads)
typedef float tail_f8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ tail_f8 tail_taps8(const float* base, int chunk)
{
    typedef const __attribute__((address_space(4))) tail_f8* cp;
    uint64_t a = reinterpret_cast<uint64_t>(base) + 32u * (uint32_t)chunk;
    asm volatile("" : "+s"(a));
    return *reinterpret_cast<cp>(a);
}

// d: decimator output (d[0] = global index kd0); audio[0] = output q0; groups: 3 rows of row_stride floats (group g =
// outputs m = 3c + g); rplain: the resampler's plain taps; fhalf / fplain: the audio filter's 64 half-taps / 128 plain taps
__global__ void __launch_bounds__(TAIL_NT, 4) k_fm_tail(const float* __restrict__ d_in, float* __restrict__ audio,
                                                        const float* __restrict__ groups, const float* __restrict__ rplain,
                                                        const float* __restrict__ fhalf, const float* __restrict__ fplain, TailParams p
)
{
    __shared__ __attribute__((aligned(16))) float ys[TAIL_NY_PAD];
    __shared__ __attribute__((aligned(16))) float zs[TAIL_NZ + 16];
    // the un-grouped taps of the two sequential (Cross) kernels: read per lane at data-dependent offsets, and a dependent
    // chain of global loads on one wave would hold the whole workgroup at the next barrier
    __shared__ float rpl[3 * TAIL_NL];
    __shared__ float fpl[TAIL_LF];
    if (threadIdx.x < 3 * TAIL_NL) rpl[threadIdx.x] = threadIdx.x < p.ntaps ? rplain[threadIdx.x] : 0.0f;
    if (threadIdx.x < TAIL_LF) fpl[threadIdx.x] = fplain[threadIdx.x];
    const int tid = threadIdx.x;
    const int64_t qa = (p.q0 / 3) * 3 + (int64_t)blockIdx.x * TAIL_A;   // first audio output of the tile (cycle aligned)
    const int64_t c0 = qa / 3;                                           // first resampler cycle
    const int64_t y0 = 10 * c0;                                          // first y of the tile

    // ---- phase 1: fmDemod.  y[k] = phase(d[k] * conj d[k-1]); d[-1] = 0 (Demod.hs:41).
    // Four consecutive samples per thread and round (five 8-byte loads: the stream is only 8-byte aligned in general), the
    // next round's loads in flight while this round's ~400 VALU instructions run.
    const float2* d2 = reinterpret_cast<const float2*>(d_in);
    constexpr int NQ = (TAIL_NY + 3) / 4;                      // quads of y per tile
    constexpr int ROUNDS = (NQ + TAIL_NT - 1) / TAIL_NT;
    const int64_t klo = p.ky0 > y0 ? p.ky0 : y0;               // first y this tile has to produce
    auto load5 = [&](int qd, float2 (&v)[5]) {
        const int64_t k = y0 + 4 * (int64_t)qd;                // v[e] = d[k - 1 + e]
#pragma unroll
        for (int e = 0; e < 5; e++) {
            const int64_t ke = k - 1 + e;
            v[e] = (ke >= p.kd0 && ke < p.kd1 && qd < NQ) ? d2[ke - p.kd0] : make_float2(0.0f, 0.0f);
        }
    };
    float2 cur[5], nxt[5];
    load5(tid, cur);
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int qd = tid + rd * TAIL_NT;
        if (rd + 1 < ROUNDS) load5(qd + TAIL_NT, nxt);
        const int64_t kq = y0 + 4 * (int64_t)qd;
        if (qd < NQ && kq + 3 >= klo && kq < p.ky1) {            // quads outside the launch's range feed no output: skipped
            const int64_t k = kq;
            // all four unconditionally (four independent dependency chains side by side; out-of-range inputs were loaded as
            // zeros), the range test is a select afterwards
            float a[4];
            fm_phase_voted<4>(cur, a);                            // common case + vote among the lanes in here (demod.hpp)
            const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
            float4 r;
            r.x = (k + 0 >= klo && k + 0 < p.ky1) ? a0 : 0.0f;
            r.y = (k + 1 >= klo && k + 1 < p.ky1) ? a1 : 0.0f;
            r.z = (k + 2 >= klo && k + 2 < p.ky1) ? a2 : 0.0f;
            r.w = (k + 3 >= klo && k + 3 < p.ky1) ? a3 : 0.0f;
            *reinterpret_cast<float4*>(&ys[4 * qd]) = r;
        }
#pragma unroll
        for (int e = 0; e < 5; e++) cur[e] = nxt[e];
    }
    __syncthreads();

    // ---- phase 2: polyphase resampler, one cycle (3 outputs) per thread and round.
    // Seam classification in 32-bit arithmetic relative to the tile (the launcher guarantees a tile spans less than one
    // buffer of the reference's Pipes, so at most one boundary crosses it): rz0 / rq0 = position of the tile's first z
    // window / first audio window inside its buffer, in upsampled units / z samples.
    const int sBI = (int)(p.seam * 3);
    const int rz0 = sBI > 0 ? (int)((30 * c0) % sBI) : 0;
    const int rq0 = p.seam > 0 ? (int)(qa % p.seam) : 0;
    // z values this launch needs from the tile: m in [max(q0, qa), min(q1, qa + A) + 127) -- a short launch (one source
    // block per push) fills only the front of its single tile, and the rest is not computed
    const int64_t m_need_lo = p.q0 > qa ? p.q0 : qa;
    const int64_t m_need_hi = (p.q1 < qa + TAIL_A ? p.q1 : qa + TAIL_A) + TAIL_LF - 1;
    const int cl_lo = (int)((m_need_lo - qa) / 3);
    const int cl_hi = (int)((m_need_hi - qa + 2) / 3);                 // one past the last cycle needed (<= NC)
#pragma unroll 1
    for (int cl = cl_lo + tid; cl < cl_hi && cl < TAIL_NC; cl += TAIL_NT) {
        constexpr int PRE[3] = {0, 4, 7};
        constexpr int WIN = 7 + TAIL_NL;                               // 71 floats
        float w[WIN + 1];
        const float* win = ys + 10 * cl;                               // 8-byte aligned; 10-dword lane stride: conflict-free
#pragma unroll
        for (int i = 0; i < (WIN + 1) / 2; i++) {
            const float2 t = *reinterpret_cast<const float2*>(win + 2 * i);
            w[2 * i] = t.x;
            w[2 * i + 1] = t.y;
        }
        float res[3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            // the 64 taps of the group are wave-uniform: scalar loads (constant address space) from an address the compiler
            // cannot see through -- loop-invariant code motion would otherwise park all 192 taps in SGPRs across the `cl`
            // loop and spill most of them; within one group the loads are free to go out together
            uint64_t ca = reinterpret_cast<uint64_t>(groups + g * p.row_stride);
            asm volatile("" : "+s"(ca));
            const __attribute__((address_space(4))) float* c = reinterpret_cast<const __attribute__((address_space(4))) float*>(ca);
            float acc[8];
#pragma unroll
            for (int l = 0; l < 8; l++) acc[l] = 0.0f;
#pragma unroll
            for (int j = 0; j < TAIL_NL; j++) acc[j & 7] = acc[j & 7] + c[j] * w[PRE[g] + j];
            res[g] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        if (sBI > 0) {
#pragma unroll
            for (int g = 0; g < 3; g++) {
                int rv = rz0 + 10 * (3 * cl + g);                      // window start inside its buffer (upsampled units)
                if (rv >= sBI) rv -= sBI;
                if (rv + p.rLp <= sBI) continue;                       // the window lies inside one buffer: a One output
                const int64_t m = 3 * (c0 + cl) + g;
                const int64_t v = m * 10;
                const int64_t edge = v + (sBI - rv);                   // the buffer boundary the window straddles
                if (seam_has_crossover(edge, 3, 10, p.rLp) && !late_output_is_one(m, edge, 3, 10, p.seam)) {
                    // resampleCrossHighLevel (FilterInternal.hs:410-423): stride 3 through the unpadded taps, sequential
                    const int64_t pos = (v + 2) / 3;
                    const int fo = (int)(pos * 3 - v);
                    const float* x = ys + (pos - y0);
                    const int nterms = (p.ntaps - fo + 2) / 3;            // taps fo, fo+3, .. < ntaps
                    const float* tp = rpl + fo;
                    float r = 0.0f;
#pragma unroll 8
                    for (int l = 0; l < nterms; l++) r = r + x[l] * tp[3 * l];
                    res[g] = r;
                }
            }
        }
        zs[3 * cl] = res[0];
        zs[3 * cl + 1] = res[1];
        zs[3 * cl + 2] = res[2];
    }
    __syncthreads();

    // ---- phase 3: symmetric audio filter (pair-add first, 8 lanes, tree) + gain; 4 consecutive outputs per thread.
    constexpr int NK = TAIL_LF / 2;
    const int t_lo = (int)((m_need_lo - qa) / 4);
    const int o_hi = (int)((p.q1 < qa + TAIL_A ? p.q1 : qa + TAIL_A) - qa);   // one past the last output of the tile this launch owns
#pragma unroll 1
    for (int t = t_lo + tid; 4 * t < o_hi; t += TAIL_NT) {
        const float* win = zs + 4 * t;
        float acc[4][8];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int l = 0; l < 8; l++) acc[r][l] = 0.0f;
#pragma unroll 1
        for (int j = 0; j < NK / 8; j++) {
            const tail_f8 c8 = tail_taps8(fhalf, j);
            const float* fp = win + 8 * j;
            const float* bp = win + 2 * NK - 8 - 8 * j;
            float fw[12], bw[12];
#pragma unroll
            for (int qd = 0; qd < 3; qd++) {
                const float4 a = *reinterpret_cast<const float4*>(fp + 4 * qd);
                fw[4 * qd] = a.x; fw[4 * qd + 1] = a.y; fw[4 * qd + 2] = a.z; fw[4 * qd + 3] = a.w;
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * qd);
                bw[4 * qd] = b.x; bw[4 * qd + 1] = b.y; bw[4 * qd + 2] = b.z; bw[4 * qd + 3] = b.w;
            }
#pragma unroll
            for (int kk = 0; kk < 8; kk++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r][kk] = acc[r][kk] + c8[kk] * (fw[r + kk] + bw[r + 7 - kk]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = 4 * t + r;
            const int64_t q = qa + o;
            float res = ((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3])) + ((acc[r][4] + acc[r][5]) + (acc[r][6] + acc[r][7]));
            if (p.seam > 0) {
                int rq = rq0 + o;                                      // window start inside its buffer of z samples
                if (rq >= (int)p.seam) rq -= (int)p.seam;
                if (rq + TAIL_LF > (int)p.seam) {
                    // filterCrossHighLevel (FilterInternal.hs:404-408) on coeffs ++ reverse coeffs: sequential, no pair-add
                    float s = 0.0f;
#pragma unroll 16
                    for (int j = 0; j < TAIL_LF; j++) s = s + zs[o + j] * fpl[j];
                    res = s;
                }
            }
            res = res * p.gain;
            if (o < TAIL_A && q >= p.q0 && q < p.q1) audio[q - p.q0] = res;
        }
    }
}


}  // namespace

bool launch_fm_tail_fused(hipStream_t s, const float* d_d, int64_t kd0, int64_t kd1, int64_t ky0, int64_t ky1, float* d_audio,
                          int64_t q0, int64_t q1, const float* d_groups, int row_stride, int nloop, const int* increments,
                          int ngroups, int I, int D, int rLp, const float* d_rplain, int ntaps, const float* d_fhalf, int nhalf,
                          const float* d_fplain, float gain, int64_t seam)
{
    // specialised for the FM chain's tail: 3/10 resampler with 64-float groups, 64 half-tap symmetric filter, AVX orders
    if (!(ngroups == 3 && nloop == TAIL_NL && I == 3 && D == 10 && increments[0] == 4 && increments[1] == 3 && increments[2] == 3)) return false;
    if (!(nhalf == TAIL_LF / 2 && rLp <= 3 * TAIL_NL && ntaps <= rLp)) return false;
    // a tile must span less than one buffer of the reference's Pipes (one boundary per tile at most, 32-bit seam arithmetic)
    if (seam != 0 && (seam * 3 < 10 * (int64_t)TAIL_NZ + rLp || seam < TAIL_A + 4 + TAIL_LF || seam > (1 << 28))) return false;
    if (q1 <= q0) return true;
    TailParams p;
    p.kd0 = kd0; p.kd1 = kd1; p.ky0 = ky0; p.ky1 = ky1; p.q0 = q0; p.q1 = q1;
    p.row_stride = row_stride; p.ntaps = ntaps; p.rLp = rLp; p.gain = gain; p.seam = seam;
    const int64_t qa0 = (q0 / 3) * 3;
    const int64_t tiles = (q1 - qa0 + TAIL_A - 1) / TAIL_A;
    hipLaunchKernelGGL(k_fm_tail, dim3((unsigned)tiles), dim3(TAIL_NT), 0, s, d_d, d_audio, d_groups, d_rplain, d_fhalf, d_fplain, p);
    return true;
}

}  // namespace sdrhip
