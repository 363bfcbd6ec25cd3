// kernels.hpp -- host-callable launchers of the gfx950 kernels (internal header).
//
// Every launcher is asynchronous on `stream` and works on DEVICE pointers.  The
// arithmetic contract (summation orders, no FMA) is documented per kernel in
// kernels_generic.hip / kernels_fast.hip; the file is compiled with
// -ffp-contract=off, which the parity of every result depends on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrhip {

// Which reference variant's summation order is reproduced.
//  real data:    lanes 1 (scalar), 4 (SSE), 8 (AVX)            common.h:34-72
//  complex data: CO_SEQ   scalar                                common.h:95-106
//                CO_L2/L4 "RC":  duplicated taps, 2/4 complex lanes  (decimate.c:84-113)
//                CO_X2/X4 "RC2": plain taps, 4/8 complex partials folded (common.h:108-155)
enum ComplexOrder { CO_SEQ = 0, CO_L2 = 1, CO_L4 = 2, CO_X2 = 3, CO_X4 = 4 };

// Stream geometry of one launch (see include/sdr_hip.h "Stream semantics").
struct Geom {
    int64_t in_base;   // global index of in[0]
    int64_t k_begin;   // global index of out[0]
    int     count;     // outputs in this launch
    int     I;         // interpolation (1 for filter / decimator)
    int     D;         // decimation
    int     Lp;        // Pipe-visible (padded) length in upsampled units
    int64_t seamBI;    // seam_block * I; 0 = contiguous (all One); < 0 = every output Cross
    int64_t outB = 0;  // the Pipe's output block size (0 = unbounded): see late_output_is_one below
};

// Does the reference's Pipe enter its crossover (sequential) code at the buffer boundary `edge` (upsampled units, a
// multiple of seamBI)?  m* = the first output whose window no longer fits the buffer ending at `edge`; the Pipe crosses
// over iff m*'s first input sample, ceil(m* D / I), still lies in that buffer -- and then computes EVERY output whose
// virtual start m D is before the edge sequentially (firResampler: `outputsComputable`, Filter.hs:712-716).  Otherwise
// (`VG.length bufIn' == 0 -> simple next`, Filter.hs:707-709; only possible when I > 1) m* is simply the first SIMD
// output of the next buffer and the seam has no sequential outputs at all.  For I == 1 both tests coincide.
__host__ __device__ inline bool seam_has_crossover(int64_t edge, int I, int D, int Lp)
{
    if (I == 1) return true;
    const int64_t m_star = edge >= Lp ? (edge - Lp) / D + 1 : 0;
    const int64_t first_in = (m_star * D + I - 1) / I;
    return first_in * I < edge;
}

// Inside a crossover the Pipe computes every output whose virtual start precedes the boundary -- but only as far as
// its OUTPUT block has room (`count = min outputsComputable (space bufferOut)`, Filter.hs:715); when it comes back with a
// fresh output block it re-decides by the first input sample (`inputUsed >= VG.length bufLast -> simple`, Filter.hs:722-724).
// So the one output whose virtual start lies in the last I-1 zero-stuffed positions before the boundary (first input
// already in the next buffer) is sequential -- unless it is the first output of an output block, in which case it is the
// next buffer's first SIMD output.  outB = the output block size (global output index m = 0 starts a block).
__host__ __device__ inline bool late_output_is_one(int64_t m, int64_t edge, int I, int D, int64_t outB)
{
    if (I == 1 || outB <= 0 || m % outB != 0) return false;
    const int64_t first_in = (m * D + I - 1) / I;
    return first_in * I >= edge;
}

// Is output m of a seamed stream computed in the reference's sequential ("Cross") order?  One when the window fits the buffer it
// starts in; else Cross -- unless the Pipe never crosses over at that boundary (seam_has_crossover), or the output is the late
// first One of an output block.
__host__ __device__ inline bool is_cross(const Geom& g, int64_t m)
{
    if (g.seamBI == 0) return false;
    if (g.seamBI < 0) return true;   // every output of this launch is a seam straddler
    const int64_t v = m * (int64_t)g.D;
    const int64_t edge = (v / g.seamBI + 1) * g.seamBI;
    if (v + g.Lp <= edge) return false;
    if (late_output_is_one(m, edge, g.I, g.D, g.outB)) return false;
    return seam_has_crossover(edge, g.I, g.D, g.Lp);
}

// Real data ------------------------------------------------------------------
// taps: `ntaps` floats (multiple of lanes).  sym: taps are the HALF filter.
// cross_taps: Lp floats used by the sequential "Cross" outputs (may be null when seamBI == 0).
void launch_fir_real(hipStream_t s, const Geom& g, int lanes, bool sym, const float* d_taps, int ntaps,
                     const float* d_cross_taps, const float* d_in, float* d_out);

// Complex data, real taps.  For CO_L2/CO_L4 d_taps holds 2*P interleaved
// (duplicated) floats and ntaps = 2P; otherwise P plain taps.  sym => half taps
// (only with CO_X2/CO_X4: the reference's SymmetricRC kernels).
void launch_fir_cplx(hipStream_t s, const Geom& g, ComplexOrder order, bool sym, const float* d_taps, int ntaps,
                     const float* d_cross_taps, const float* d_in, float* d_out);
// same with interleaved u8 IQ input (convert fused into the load)
void launch_fir_cplx_u8(hipStream_t s, const Geom& g, ComplexOrder order, const float* d_taps, int ntaps,
                        const float* d_cross_taps, const uint8_t* d_in, float* d_out);

// Polyphase resampler.  Output i of the launch uses group (group0 + i) % ngroups
// and starts at input  pos0 + (i / ngroups) * period + pre[i % ngroups]  (relative
// to d_in).  groups: ngroups rows of `row_stride` floats; the dot product runs
// over nloop floats (roundUp(num_coeffs, lanes)).
struct ResampTable {
    int ngroups;
    int group0;
    int64_t pos0;
    int period;       // inputs consumed by one full cycle of groups
    int pre[64];      // prefix of increments starting at group0
    int row_stride;
    int nloop;
    // for Cross outputs (seams) and the legacy sequential resampler: plain taps,
    // filter offset of each group, and force_seq = every output sequential
    int ntaps_plain;
    int fo[64];
    int force_seq;
    // more than 64 groups (the reference's C runs any count, resample.c:34-142): the per-group tables live in device memory
    // instead of the kernel arguments -- ext[0 .. ngroups] = prefix sums of the increments from group 0 (ext[ngroups] = period),
    // ext[ngroups + 1 + g] = filter offset of group g -- and pre[] / fo[] above are unused.  Generic kernels only.
    const int* ext = nullptr;
};
void launch_resample_real(hipStream_t s, const Geom& g, int lanes, const ResampTable& t, const float* d_groups,
                          const float* d_plain_taps, const float* d_in, float* d_out);
void launch_resample_cplx(hipStream_t s, const Geom& g, ComplexOrder order, const ResampTable& t,
                          const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out);

// Element-wise ----------------------------------------------------------------
void launch_convert_u8(hipStream_t s, const uint8_t* d_in, float* d_out, int64_t n);
void launch_convert_i16(hipStream_t s, const int16_t* d_in, float* d_out, int64_t n);
void launch_convert_f32_to_i16_bladerf(hipStream_t s, const float* d_in, int16_t* d_out, int64_t n);
void launch_scale(hipStream_t s, float factor, const float* d_in, float* d_out, int64_t n);
// y[i] = phase(x[i] * conj(x[i-1])), i < count; x[-1] = d_in[-1] if has_prev else (last_re,last_im)
void launch_fm_demod(hipStream_t s, const float* d_in_iq, float* d_out, int64_t count, bool has_prev,
                     float last_re, float last_im);
// filter.c:152-161 (kernels_iir.hip: speculative chunks + verification, bit-exact with the sequential walk).
// d_final receives {finalSample, finalOutput}; d_ws (dc_blocker_workspace_bytes, may be null = sequential walk)
// starts with three u32 statistics {chunks left to the sequential settle, samples it rewrote, chunks recomputed in the
// parallel repair rounds}; run_in <= 0 selects the default.  d_state, when given, holds {lastSample, lastOutput} on
// the device and overrides the two scalars (it may alias d_final: a Pipe chains its blocks that way).
size_t dc_blocker_workspace_bytes(int64_t num);
void launch_dc_blocker(hipStream_t s, int64_t num, float last_sample, float last_output, const float* d_in,
                       float* d_out, float* d_final, void* d_ws, int run_in, const float* d_state = nullptr);

// Lane-split tiled kernels (kernels_split.hip): every SIMD order of the filter / decimator / resampler families,
// any factor and tap count that fits a tile.  Plain taps (sym: the half taps).  false = not applicable.
bool launch_fir_split(hipStream_t s, const Geom& g, bool cplx, int lanes, ComplexOrder corder, bool sym, const float* d_taps,
                      int ntaps, const float* d_cross_taps, const float* d_in, float* d_out, float gain, bool apply_gain);
bool launch_resample_split(hipStream_t s, const Geom& g, bool cplx, int lanes, ComplexOrder corder, const ResampTable& t,
                           const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out);

long long split_launch_count();   // diagnostics: launches the lane-split kernels have taken so far

// kernels_decimate_real.hip: real decimators by 2 / 4 / 8 / 16, AVX / SSE lane order, up to 2048 taps (sym: d_taps = nk half-taps,
// a multiple of 8, g.Lp = 2 nk): 16 / D
// outputs per thread, scalar-loaded taps, rolled walk.  False = not this kernel's shape.
// ncross: taps of the sequential (Cross) outputs when they are not the g.Lp padded ones (a resampler with interpolation 1)
bool launch_decimate_real16_fast(hipStream_t s, const Geom& g, int lanes, const float* d_taps, int nk, const float* d_cross_taps, const float* d_in,
                                 float* d_out, float gain, bool apply_gain, int ncross = 0, bool sym = false);
long long decimate_real16_launch_count();
// kernels_resample_cycle.hip: real resamplers I/D with an odd decimation (I <= 6, D in {3,5,7}), any filter length up to
// 1024 taps per group, AVX / SSE lane order: one thread per polyphase cycle, rolled walk with taps from LDS.  False = not
// this kernel's shape (the caller falls through to the split / generic kernels).
// cplx: complex data in the "RC2" orders (corder CO_X4 / CO_X2: eight / four complex partials)
bool launch_resample_cycle_fast(hipStream_t s, const Geom& g, int lanes, const ResampTable& t, const int* increments, const float* d_groups,
                                const float* d_plain_taps, const float* d_in, float* d_out, bool cplx = false, ComplexOrder corder = CO_SEQ);
long long resample_cycle_launch_count();

// Fast paths (kernels_fast.hip).  Return false when the configuration is not one
// they are specialised for; the caller then uses the generic kernel.
// kernels_chain.hip: fast paths of the low-rate stages
void launch_fm_demod_fast(hipStream_t s, const float* d_in_iq, float* d_out, int64_t count, bool has_prev,
                          float last_re, float last_im);
// real filters (D == 1), AVX order, nk taps walked (half-taps when sym), nk % 8 == 0
// lanes: 8 = AVX order, 4 = SSE order (the same kernel with four lane partials per output)
bool launch_fir_real8_fast(hipStream_t s, const Geom& g, bool sym, const float* d_taps, int nk, const float* d_cross_taps,
                          const float* d_in, float* d_out, float gain, bool apply_gain, int lanes = 8);
// complex filter (D == 1), AVX "RC" order, duplicated taps (2P floats), P % 4 == 0
bool launch_filter_cplx4_fast(hipStream_t s, const Geom& g, const float* d_dup_taps, int P, const float* d_cross_taps,
                              const float* d_in, float* d_out);
// kernels_fast_filter.hip: complex filters of exactly 128 / 64 taps (AVX "RC" order, plain taps) on the tiled decimator with D = 1
bool launch_filter_c4_tile(hipStream_t s, const Geom& g, const float* d_plain_taps, int P, const float* d_cross_taps, const float* d_in,
                           float* d_out);
// kernels_cplx.hip: the same shape on complex data ("RC2" orders of resampleAVXRC / resampleSSERC)
bool launch_resample3c_fast(hipStream_t s, const Geom& g, ComplexOrder order, const ResampTable& t, const int* increments, const float* d_groups,
                            const float* d_plain_taps, const float* d_in, float* d_out);
// d_iq != nullptr: fmDemod fused into the loader (d_iq = decimator output at input 0 of the launch, y_count inputs; d_in is
// then the y buffer, filled only where the lead / tail / seam kernels read it); false = not this shape, nothing launched.
// lanes = 8: AVX order; 4: SSE order (64-float groups, no fused demodulator)
bool launch_resample_3_10_fast(hipStream_t s, const Geom& g, const ResampTable& t, const int* increments,
                               const float* d_groups, const float* d_plain_taps, const float* d_in, float* d_out,
                               const float* d_iq = nullptr, bool iq_has_prev = false, int64_t y_count = 0, int lanes = 8);
// kernels_tail.hip: fmDemod -> 3/10 resampler -> symmetric filter (* gain) as one kernel (y and z never leave LDS); false = the
// configuration is not the FM chain's tail (3 groups of 64, increments {4,3,3}, 64 half-taps, buffers longer than a tile)
constexpr int kTailTileOutputs = 2046;   // audio outputs one workgroup of the fused tail kernel produces
bool launch_fm_tail_fused(hipStream_t s, const float* d_d, int64_t kd0, int64_t kd1, int64_t ky0, int64_t ky1, float* d_audio,
                          int64_t q0, int64_t q1, const float* d_groups, int row_stride, int nloop, const int* increments,
                          int ngroups, int I, int D, int rLp, const float* d_rplain, int ntaps, const float* d_fhalf, int nhalf,
                          const float* d_fplain, float gain, int64_t seam);
// kernels_small.hip: the WHOLE chain (u8 IQ -> /8 decimator -> fmDemod -> 3/10 resampler -> symmetric filter * gain) as one
// kernel for launch-bound runs; d_in holds samples [s0, s0 + n_in).  tile_outputs: audio outputs per workgroup (0 = chosen
// from the size of the run).  false = the configuration is not the FM chain's, nothing launched
bool launch_fm_chain_small(hipStream_t s, const uint8_t* d_in, int64_t s0, int64_t n_in, float* d_audio, int64_t q0, int64_t q1,
                           int dD, int dP, const float* d_dscaled, bool last_tap_zero, const float* d_groups, int row_stride, int nloop,
                           const int* increments, int ngroups, int I, int D, int rLp, const float* d_rplain, int ntaps,
                           const float* d_fhalf, int nhalf, const float* d_fplain, float gain, int64_t seam, int tile_outputs);
long long fm_chain_small_launch_count();   // diagnostics: launches of the one-kernel chain so far
// abi_device.cpp: the short-seamed-launch scale v (sdrhip_set_small_launch_outputs)
int small_launch_outputs();

// last_tap_zero: tap P-1 is the zero the constructor padded the filter with (lets the u8 path skip its MACs)
bool launch_decimate_c4_fast(hipStream_t s, const Geom& g, const float* d_plain_taps, int P, const float* d_cross_taps,
                             const void* d_in, bool in_is_u8, float* d_out, bool last_tap_zero = false);
// kernels_systolic.hip (round 4): the One outputs of a decimate-by-8, 128-tap, AVX-order launch by the register-resident systolic
// walk (d_taps: plain taps, pre-scaled by 1/128 for u8 input).  false = not this shape / too small, nothing launched.
// seams_done (with d_cross_taps, the plain taps of the sequential outputs): set when the launch also computed its Cross outputs
// (fix-up workgroups interleaved into the same launch, round 6) -- the caller then skips its fix-up launch.
bool launch_decimate_c4_systolic(hipStream_t s, const Geom& g, const float* d_taps, int P, const void* d_in, bool in_is_u8, float* d_out,
                                 bool last_tap_zero, const float* d_cross_taps = nullptr, bool* seams_done = nullptr);
void systolic_plan(int count, int* nstrips, int* nwhole);   // host arithmetic of the strip cut (CPU-testable)
void set_systolic(int mode);        // 0 = the tile kernel everywhere, 1 = the systolic kernel wherever its shape fits, 2 (default) = by launch size
long long systolic_launch_count();  // diagnostics
// kernels_fast_orders.hip: the same tiled decimator for the SSE "RC" and the "RC2" summation orders (CO_L2, CO_X4, CO_X2)
bool launch_decimate_c_orders_fast(hipStream_t s, const Geom& g, ComplexOrder order, const float* d_plain_taps, int P,
                                   const float* d_cross_taps, const void* d_in, bool in_is_u8, float* d_out);

}  // namespace sdrhip
