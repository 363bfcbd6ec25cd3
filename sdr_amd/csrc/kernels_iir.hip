// kernels_iir.hip -- dcBlocker (c_sources/filter.c:152-161, Pipe: hs_sources/SDR/Filter.hs:730-739), SURVEY.md 8(f) N2.
//
//     lastOutput = inBuf[i] - lastSample + 0.997 * lastOutput      (f32 subtract, then f64 multiply/add, rounded to f32)
//
// A one-pole IIR whose state is ROUNDED every step is not an associative scan, so no parallel prefix gives the reference's
// bits.  What does: the recurrence is a contraction (0.997), so a lane that starts W samples before its chunk from a wrong
// state forgets the error -- it shrinks below one ULP and the two trajectories snap together, after which they are identical
// forever (the map is deterministic).  So:
//   1. k_dc_speculate: one lane per chunk of C samples runs in from `W` samples earlier with state 0, records the state it
//      reaches at the chunk start, writes its chunk, records its end state.  Chunk 0 (and any chunk whose run-in reaches
//      sample 0) starts from the caller's true state.
//   2. Chunk j is exact if the state it started from equals chunk j-1's end state bit for bit and chunk j-1 is exact.
//      k_dc_repair (a few rounds, all chunks in parallel): a chunk whose start state differs from its predecessor's current
//      end state is recomputed from that state.  One round fixes every isolated miss; a miss whose correction changes the
//      chunk's own end state needs the next round for its successor.
//   3. k_dc_settle (one workgroup) checks the chain once more in parallel; whatever is still inconsistent (pathological
//      inputs only) is walked by one lane from the true state until its values meet what is stored.
// The result is the sequential result by construction, whatever the input; only the speed depends on the convergence.
#include <stdlib.h>

#include "kernels.hpp"

namespace sdrhip {

__device__ __forceinline__ float dc_step(float x, float xprev, float y)
{
    const float d = x - xprev;
    return (float)((double)d + 0.997 * (double)y);
}

__global__ void __launch_bounds__(64)
k_dc_speculate(int64_t num, int C, int W, float last_sample, float last_output, const float* __restrict__ d_state,
               const float* __restrict__ in, float* __restrict__ out, uint32_t* __restrict__ y_start,
               uint32_t* __restrict__ y_end, int nchunks, int vec4)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= nchunks) return;
    const int64_t begin = (int64_t)j * C;
    const int64_t end = begin + C < num ? begin + C : num;
    int64_t i = begin - W;
    float y, xp;
    if (i <= 0) {
        i = 0;
        y = d_state ? d_state[1] : last_output;   // a Pipe keeps (lastSample, lastOutput) on the device between blocks
        xp = d_state ? d_state[0] : last_sample;
    } else {
        y = 0.0f;
        xp = in[i - 1];
    }
    if (vec4) {   // C, W multiples of 4 and `in`/`out` 16-byte aligned: whole float4 groups except in the last chunk
        for (; i < begin; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            y = dc_step(v.x, xp, y);
            y = dc_step(v.y, v.x, y);
            y = dc_step(v.z, v.y, y);
            y = dc_step(v.w, v.z, y);
            xp = v.w;
        }
        y_start[j] = __float_as_uint(y);
        for (; i + 4 <= end; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            float4 o;
            o.x = y = dc_step(v.x, xp, y);
            o.y = y = dc_step(v.y, v.x, y);
            o.z = y = dc_step(v.z, v.y, y);
            o.w = y = dc_step(v.w, v.z, y);
            xp = v.w;
            *reinterpret_cast<float4*>(out + i) = o;
        }
    } else {
        for (; i < begin; i++) {
            const float x = in[i];
            y = dc_step(x, xp, y);
            xp = x;
        }
        y_start[j] = __float_as_uint(y);
    }
    for (; i < end; i++) {
        const float x = in[i];
        y = dc_step(x, xp, y);
        xp = x;
        out[i] = y;
    }
    y_end[j] = __float_as_uint(y);
}

// One repair round.  y_end_prev is read, y_end_next written (ping-pong: a round must not see its own updates).
__global__ void __launch_bounds__(64)
k_dc_repair(int64_t num, int C, int W, const float* __restrict__ in, float* __restrict__ out, uint32_t* __restrict__ y_start,
            const uint32_t* __restrict__ y_end_prev, uint32_t* __restrict__ y_end_next, int nchunks,
            uint32_t* __restrict__ stats)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= nchunks) return;
    const int64_t begin = (int64_t)j * C;
    uint32_t e = y_end_prev[j];
    if (begin - W > 0) {
        const uint32_t s_true = y_end_prev[j - 1];
        if (y_start[j] != s_true) {
            const int64_t end = begin + C < num ? begin + C : num;
            float y = __uint_as_float(s_true), xp = in[begin - 1];
            for (int64_t i = begin; i < end; i++) {
                const float x = in[i];
                y = dc_step(x, xp, y);
                xp = x;
                out[i] = y;
            }
            y_start[j] = s_true;
            e = __float_as_uint(y);
            atomicAdd(&stats[2], 1u);
        }
    }
    y_end_next[j] = e;
}

// One workgroup.  stats[0] = chunks still inconsistent after the parallel rounds, stats[1] = samples this lane rewrote
// (stats[2], counted by k_dc_repair = chunks recomputed in the parallel rounds).
__global__ void __launch_bounds__(1024)
k_dc_settle(int64_t num, int C, int W, const float* __restrict__ in, float* __restrict__ out,
            const uint32_t* __restrict__ y_start, const uint32_t* __restrict__ y_end, int nchunks,
            uint8_t* __restrict__ bad, float* __restrict__ fin, uint32_t* __restrict__ stats)
{
    __shared__ int nbad;
    if (threadIdx.x == 0) nbad = 0;
    __syncthreads();
    int mine = 0;
    for (int j = threadIdx.x; j < nchunks; j += blockDim.x) {
        const bool exact_start = (int64_t)j * C - W <= 0;
        const bool b = !exact_start && y_start[j] != y_end[j - 1];
        bad[j] = b;
        mine += b;
    }
    if (mine) atomicAdd(&nbad, mine);
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t rewritten = 0;
    if (nbad > 0) {
        int64_t reach = -1;   // everything up to here is final
        for (int j = 1; j < nchunks; j++) {
            if (!bad[j]) continue;
            int64_t i = (int64_t)j * C;
            if (i <= reach) continue;
            float y = out[i - 1], xp = in[i - 1];
            for (; i < num; i++) {
                const float x = in[i];
                y = dc_step(x, xp, y);
                xp = x;
                if (__float_as_uint(y) == __float_as_uint(out[i])) break;   // met the stored trajectory: the rest is right
                out[i] = y;
                rewritten++;
            }
            reach = i;
        }
    }
    stats[0] = (uint32_t)nbad;
    stats[1] = rewritten;
    fin[0] = in[num - 1];
    fin[1] = out[num - 1];
}

// plain sequential walk: short blocks (the run-in would cost more than the block).  Loads run ahead of the
// dependent f64 chain: the next float4 is requested before the current one is consumed.
__global__ void k_dc_sequential(int64_t num, float last_sample, float last_output, const float* d_state,
                                const float* __restrict__ in, float* __restrict__ out, float* fin,
                                uint32_t* __restrict__ stats, int vec4)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float y = d_state ? d_state[1] : last_output, xp = d_state ? d_state[0] : last_sample;
    int64_t i = 0;
    if (vec4 && num >= 8) {
        const int64_t n4 = num & ~(int64_t)3;
        float4 v = *reinterpret_cast<const float4*>(in);
        for (; i < n4; i += 4) {
            const int64_t nx = i + 4 < n4 ? i + 4 : i;
            const float4 vn = *reinterpret_cast<const float4*>(in + nx);
            float4 o;
            o.x = y = dc_step(v.x, xp, y);
            o.y = y = dc_step(v.y, v.x, y);
            o.z = y = dc_step(v.z, v.y, y);
            o.w = y = dc_step(v.w, v.z, y);
            xp = v.w;
            *reinterpret_cast<float4*>(out + i) = o;
            v = vn;
        }
    }
    for (; i < num; i++) {
        const float x = in[i];
        y = dc_step(x, xp, y);
        xp = x;
        out[i] = y;
    }
    fin[0] = xp;
    fin[1] = y;
    if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; }
}

namespace {
constexpr int DC_RUN_IN = 12288;   // samples of run-in: 0.997^12288 ~ 1e-16 of the initial error, plus room for the last-ULP snap
constexpr int DC_REPAIR_ROUNDS = 3;
// Every lane walks run-in + chunk samples at the latency of the dependent f64 chain, so the time is ~(C + W) steps whatever
// the lane count; more lanes than this only add redundant run-in traffic (measured: 2^18 lanes thrash the L2 at n = 2^26).
constexpr int64_t kDcMaxLanes = 32768;
struct DcPlan { int C; int nchunks; };
DcPlan dc_plan(int64_t num)
{
    const int64_t lanes = kDcMaxLanes;
    int64_t C = (num + lanes - 1) / lanes;
    if (C < 256) C = 256;
    C = (C + 3) & ~(int64_t)3;
    return {(int)C, (int)((num + C - 1) / C)};
}
}  // namespace

size_t dc_blocker_workspace_bytes(int64_t num)
{
    const DcPlan p = dc_plan(num > 0 ? num : 1);
    return (size_t)p.nchunks * 13 + 64;   // stats, y_start, y_end x2 (u32 each), bad (u8)
}

void launch_dc_blocker(hipStream_t s, int64_t num, float last_sample, float last_output, const float* d_in,
                       float* d_out, float* d_final, void* d_ws, int run_in, const float* d_state)
{
    if (num <= 0) return;
    const int W = run_in > 0 ? (run_in + 3) & ~3 : DC_RUN_IN;
    const int vec4 = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
    uint32_t* stats = reinterpret_cast<uint32_t*>(d_ws);
    if (d_ws == nullptr || num < 2 * (int64_t)W) {
        hipLaunchKernelGGL(k_dc_sequential, dim3(1), dim3(64), 0, s, num, last_sample, last_output, d_state, d_in, d_out, d_final,
                           stats, vec4);
        return;
    }
    const DcPlan p = dc_plan(num);
    uint32_t* y_start = stats + 16;
    uint32_t* y_end[2] = {y_start + p.nchunks, y_start + 2 * (size_t)p.nchunks};
    uint8_t* bad = reinterpret_cast<uint8_t*>(y_start + 3 * (size_t)p.nchunks);
    const dim3 grid((p.nchunks + 63) / 64);
    (void)hipMemsetAsync(stats, 0, 16, s);
    hipLaunchKernelGGL(k_dc_speculate, grid, dim3(64), 0, s, num, p.C, W, last_sample, last_output, d_state, d_in, d_out,
                       y_start, y_end[0], p.nchunks, vec4);
    int cur = 0;
    for (int r = 0; r < DC_REPAIR_ROUNDS; r++, cur ^= 1)
        hipLaunchKernelGGL(k_dc_repair, grid, dim3(64), 0, s, num, p.C, W, d_in, d_out, y_start, y_end[cur], y_end[cur ^ 1],
                           p.nchunks, stats);
    hipLaunchKernelGGL(k_dc_settle, dim3(1), dim3(1024), 0, s, num, p.C, W, d_in, d_out, y_start, y_end[cur], p.nchunks, bad,
                       d_final, stats);
}

}  // namespace sdrhip
