/*
 * sdr_hip.h -- C ABI of libsdr_hip.so: the MI355X (gfx950) implementation of the
 * streaming FIR / decimate / polyphase-resample / FM-demod / u8->cfloat hot path
 * of adamwalker/sdr.
 *
 * Three layers, all `extern "C"`, plain pointers and sizes only:
 *
 *  (1) DROP-IN SYMBOLS -- the reference's own native entry points, same names and
 *      signatures as c_sources/{convert,decimate,filter,resample,scale}.c, so the
 *      reference's `foreign import ccall unsafe "<sym>"` declarations
 *      (hs_sources/SDR/FilterInternal.hs:80-150,179-249,346-391; SDR/Util.hs:100-125)
 *      resolve against this library unchanged.  HOST pointers in, HOST pointers
 *      out, synchronous; results are bit-identical to the reference's x86 build
 *      (each variant keeps its own summation order).
 *
 *  (2) DEVICE STREAM API (sdrhip_*) -- descriptors that own prepared taps in HBM
 *      (the reference's Filter/Decimator/Resampler records, Filter.hs:116-144) and
 *      `_run` calls on DEVICE pointers, asynchronous on a caller-supplied HIP
 *      stream, addressed by GLOBAL stream index so that a contiguous sample
 *      stream can be cut into launches / shards anywhere without changing a bit.
 *
 *  (3) PIPE OPERATORS (sdrhip_pipe_*) -- the host-block push interface mirroring
 *      firFilter / firDecimator / firResampler / fmDemod /
 *      interleavedIQUnsignedByteToFloat (Filter.hs:532-727, Demod.hs:40-46,
 *      Util.hs:104-138): feed host blocks, receive host blocks of exactly
 *      blockSizeOut elements, pinned double-buffered hipMemcpyAsync inside.
 *
 * Stream semantics shared by (2) and (3).  Let a stage have interpolation I
 * (1 for filters/decimators), decimation D, and Pipe-visible length Lp
 * (numCoeffsF / numCoeffsD / numCoeffsR, i.e. the PADDED length).  Output m of
 * the whole stream has virtual window [m*D, m*D + Lp) in the I-times upsampled
 * input index space.  With `seam_block` = B > 0 (the size of the input blocks
 * the reference Pipe would have been fed), output m is
 *    "One"   (computed by the C SIMD kernel, lane order of `order`)  when its
 *            virtual window lies inside one input block, and
 *    "Cross" (computed by the pure-Haskell sequential kernel,
 *            FilterInternal.hs:397-423) when it straddles a multiple of B*I
 * -- exactly the split Filter.hs:536-727 makes, including its one irregular
 * case: when the first output that no longer fits a block already has its
 * first INPUT sample ceil(m*D/I) in the next block (only possible for I > 1
 * with filters shorter than the decimation step), the Pipe does not cross
 * over at that boundary (Filter.hs:707-709) and that output is the next
 * block's first "One".  seam_block = 0 means one
 * contiguous buffer: every output is "One" (what a single FFI call computes).
 *
 * Errors: drop-in symbols cannot report (void returns): on a failure they call the
 * handler of sdrhip_set_error_handler and return, or -- without one -- print to stderr and abort().  sdrhip_* return SDRHIP_OK (0) or a negative code;
 * sdrhip_last_error() returns a thread-local message.
 */
#ifndef SDR_HIP_H
#define SDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* (1) Drop-in symbols.  Signatures are the reference's, verbatim.          */
/* ------------------------------------------------------------------------ */

/* c_sources/convert.c:15,22,37 -- num = number of BYTES (= floats out). */
void convertC(int num, uint8_t *in, float *out);
void convertCSSE(int num, uint8_t *in, float *out);
void convertCAVX(int num, uint8_t *in, float *out);
/* c_sources/convert.c:52,59,72 (BladeRF i16 -> f32) and :87 (f32 -> i16 TX) */
void convertCBladeRF(int num, int16_t *in, float *out);
void convertCSSEBladeRF(int num, int16_t *in, float *out);
void convertCAVXBladeRF(int num, int16_t *in, float *out);
void convertBladeRFTransmit(int num, float *in, int16_t *out);

/* c_sources/scale.c:15,22,30 */
void scale(int num, float factor, float *in_buf, float *out_buf);
void scaleSSE(int num, float factor, float *in_buf, float *out_buf);
void scaleAVX(int num, float factor, float *in_buf, float *out_buf);

/* c_sources/filter.c:16-68 -- real taps, real data.  num = outputs. */
void filterRR(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterSSERR(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterAVXRR(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
/* numCoeffs = HALF length (filter.c:50-68) */
void filterSSESymmetricRR(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterAVXSymmetricRR(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
/* c_sources/filter.c:73-147 -- real taps, complex data.  RC: SSE/AVX take
 * DUPLICATED taps (numCoeffs = 2P floats); RC2 / SymmetricRC take plain taps. */
void filterRC(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterSSERC(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterSSERC2(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterAVXRC(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterAVXRC2(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterSSESymmetricRC(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void filterAVXSymmetricRC(int num, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
/* c_sources/filter.c:152-161 */
void dcBlocker(int num, float lastSample, float lastOutput, float *finalSample, float *finalOutput,
               float *inBuf, float *outBuf);

/* c_sources/decimate.c:16-146 */
void decimateRR(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateSSERR(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateAVXRR(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateSSESymmetricRR(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateAVXSymmetricRR(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateRC(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateSSERC(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateSSERC2(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateAVXRC(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateAVXRC2(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateSSESymmetricRC(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);
void decimateAVXSymmetricRC(int num, int factor, int numCoeffs, float *coeffs, float *inBuf, float *outBuf);

/* c_sources/resample.c:16-142.  coeffs = table of num_groups HOST pointers to
 * zero-padded polyphase groups (FilterInternal.hs:335-342); returns end group.
 * Any number of polyphase groups, as in the reference (up to 64 the per-group offset tables travel as kernel arguments,
 * beyond that in device memory).  resampleRR needs decimation > interpolation and 0 <= filter_offset < interpolation
 * (outside these the reference's own recurrence leaves its arrays): then it prints the reason and abort()s -- it cannot
 * return an error.  The descriptor API (sdrhip_resampler_*) reports SDRHIP_ERR_ARG instead. */
void resampleRR(int buf_size, int coeff_size, int interpolation, int decimation, int filter_offset,
                float *coeffs, float *in_buf, float *out_buf);
int resample2RR(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                float **coeffs, float *in_buf, float *out_buf);
int resampleSSERR(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                  float **coeffs, float *in_buf, float *out_buf);
int resampleAVXRR(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                  float **coeffs, float *in_buf, float *out_buf);
int resample2RC(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                float **coeffs, float *in_buf, float *out_buf);
int resampleSSERC(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                  float **coeffs, float *in_buf, float *out_buf);
int resampleAVXRC(int buf_size, int num_coeffs, int starting_group, int num_groups, int *increments,
                  float **coeffs, float *in_buf, float *out_buf);

/* NEW seam (the reference's fmDemod is pure Haskell, Demod.hs:21-46; SURVEY.md
 * 8(b) defines this FFI entry): out[i] = phase(in[i] * conj(in[i-1])), in[-1] =
 * (last_re,last_im).  HOST pointers. */
void fmDemodF(int num, float last_re, float last_im, const float *in_iq, float *out);

/* ------------------------------------------------------------------------ */
/* (2) Device stream API                                                    */
/* ------------------------------------------------------------------------ */

#define SDRHIP_OK            0
#define SDRHIP_ERR_ARG      (-1)  /* precondition violated (the reference's `assert` -> error) */
#define SDRHIP_ERR_HIP      (-2)  /* a HIP runtime call failed */
#define SDRHIP_ERR_NOMEM    (-3)
#define SDRHIP_ERR_STATE    (-4)

/* Summation order = which of the reference's variants is reproduced
 * (SDR.CPUID.featureSelect picks AVX on any host that has it, CPUID.hs:100-104). */
#define SDRHIP_ORDER_SCALAR  0    /* filterRR / decimateRC / resample2RR ...      */
#define SDRHIP_ORDER_SSE     1    /* 4 real / 2 complex lanes                     */
#define SDRHIP_ORDER_AVX     2    /* 8 real / 4 complex lanes                     */

const char *sdrhip_version(void);
const char *sdrhip_last_error(void);
/* Drop-in symbols return void, so a failure inside one (HIP error, out of memory, an argument the reference itself would
 * index out of bounds with) cannot be reported: by default they print the message and abort().  With a handler installed
 * (process-wide; NULL restores the default) the failing call instead records the message (sdrhip_last_error), calls
 * handler(code, message) and -- when the handler comes back -- returns to ITS caller at once, outputs unspecified.  A Haskell
 * host sets a flag in the handler and raises after the foreign call (haskell/SDR/GPU.hs: the reference raises from its
 * Pipes too, Filter.hs:526-527).  The handler must RETURN: it runs with C++ objects of the library live on the stack (a leased
 * scratch context among them), so a longjmp out of it is undefined behaviour and would strand that context. */
void sdrhip_set_error_handler(void (*handler)(int code, const char *message));
int sdrhip_device_count(void);
int sdrhip_set_device(int dev);
int sdrhip_device_name(char *buf, int buflen);
/* Tuning / test knob: the "short seamed launch" scale v.  A launch with seam_block > 0 that is short enough to be
 * launch-bound decides its One / Cross outputs inside ONE kernel instead of a tiled kernel plus a fix-up launch for the
 * seam outputs: real filters up to v/2 outputs and real resamplers up to 2v outputs take the generic kernel, the tiled
 * complex decimator computes its seam outputs in place up to 5v outputs.  Results are identical either way.
 * Default v = 32768 (also: environment SDRHIP_SMALL_LAUNCH); 0 = always the tiled kernels + fix-up launches; negative =
 * restore the default.  Returns the previous value. */
int sdrhip_set_small_launch_outputs(int outputs);

/* Thin memory / stream helpers so a non-C++ host (Haskell, ctypes) needs no HIP
 * bindings of its own.  `stream` arguments are hipStream_t passed as void*
 * (NULL = the default stream). */
int sdrhip_malloc(void **dptr, size_t bytes);
int sdrhip_free(void *dptr);
int sdrhip_malloc_host(void **hptr, size_t bytes);   /* pinned */
int sdrhip_free_host(void *hptr);
int sdrhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int sdrhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int sdrhip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int sdrhip_stream_create(void **stream);
int sdrhip_stream_destroy(void *stream);
int sdrhip_stream_sync(void *stream);

/* ---- Filter (Filter.hs:116-120; constructors :163-261) ------------------- */
typedef struct sdrhip_filter sdrhip_filter;
/* fastFilter{C,SSE,AVX}{R,C}: zero-pads to the order's SIMD multiple
 * (mkFilter Filter.hs:167-175 / mkFilterC :196-209). data_complex: 0 real, 1 complex. */
int sdrhip_filter_create(sdrhip_filter **f, int order, int data_complex, const float *coeffs, int ncoeffs);
/* fastFilterSym{SSE,AVX}R (mkFilterSymR Filter.hs:234-245): half = first half of an
 * even-length linear-phase filter; nhalf must be a multiple of 4 (SSE) / 8 (AVX). */
int sdrhip_filter_sym_create(sdrhip_filter **f, int order, const float *half, int nhalf);
int sdrhip_filter_num_coeffs(const sdrhip_filter *f);            /* numCoeffsF */
void sdrhip_filter_destroy(sdrhip_filter *f);
/* out[k - k_begin] = stream output k, k in [k_begin,k_end); d_in[0] is stream input
 * element in_base (floats for real data, (re,im) pairs for complex).  The caller
 * guarantees d_in covers inputs [k_begin, k_end - 1 + numCoeffsF). */
int sdrhip_filter_run(const sdrhip_filter *f, void *stream, const float *d_in, int64_t in_base,
                      float *d_out, int64_t k_begin, int64_t k_end, int64_t seam_block);

/* ---- Decimator (Filter.hs:126-131; constructors :277-371) ---------------- */
typedef struct sdrhip_decimator sdrhip_decimator;
int sdrhip_decimator_create(sdrhip_decimator **d, int order, int data_complex, int factor,
                            const float *coeffs, int ncoeffs);
int sdrhip_decimator_sym_create(sdrhip_decimator **d, int order, int factor, const float *half, int nhalf);
int sdrhip_decimator_num_coeffs(const sdrhip_decimator *d);      /* numCoeffsD */
int sdrhip_decimator_factor(const sdrhip_decimator *d);
void sdrhip_decimator_destroy(sdrhip_decimator *d);
/* needs inputs [k_begin*factor, (k_end-1)*factor + numCoeffsD) */
int sdrhip_decimator_run(const sdrhip_decimator *d, void *stream, const float *d_in, int64_t in_base,
                         float *d_out, int64_t k_begin, int64_t k_end, int64_t seam_block);
/* same, reading interleaved u8 IQ (the u8->cfloat conversion of convert.c:37-50
 * fused into the loader; complex decimators only) */
int sdrhip_decimator_run_u8(const sdrhip_decimator *d, void *stream, const uint8_t *d_in_iq, int64_t in_base,
                            float *d_out, int64_t k_begin, int64_t k_end, int64_t seam_block);

/* ---- Resampler (Filter.hs:137-144; constructors :408-502) ---------------- */
typedef struct sdrhip_resampler sdrhip_resampler;
int sdrhip_resampler_create(sdrhip_resampler **r, int order, int data_complex, int interpolation,
                            int decimation, const float *coeffs, int ncoeffs);
int sdrhip_resampler_num_coeffs(const sdrhip_resampler *r);      /* numCoeffsR */
int sdrhip_resampler_num_groups(const sdrhip_resampler *r);
void sdrhip_resampler_destroy(sdrhip_resampler *r);
/* Closed form of the phase recurrence (Filter.hs:613-641): first input element and
 * filter offset of stream output m (stream starting at phase 0). */
int64_t sdrhip_resampler_in_offset(const sdrhip_resampler *r, int64_t m);
int sdrhip_resampler_filter_offset(const sdrhip_resampler *r, int64_t m);
int sdrhip_resampler_group(const sdrhip_resampler *r, int64_t m);
/* out_block: the output block size of the Pipe being reproduced (firResampler's blockSizeOut; 0 =
 * unbounded).  It matters for one output in ~10^4 seams: inside a crossover the reference computes
 * the output whose virtual start lies in the last I-1 zero-stuffed positions before the boundary
 * sequentially, unless its output block was full just before it (Filter.hs:715,722-724). */
int sdrhip_resampler_run(const sdrhip_resampler *r, void *stream, const float *d_in, int64_t in_base,
                         float *d_out, int64_t k_begin, int64_t k_end, int64_t seam_block,
                         int64_t out_block);

/* ---- element-wise stages -------------------------------------------------- */
/* interleavedIQUnsignedByteToFloat (Util.hs:104-138 / convert.c): n_bytes u8 -> n_bytes f32 */
int sdrhip_convert_u8_run(void *stream, const uint8_t *d_in, float *d_out, int64_t n_bytes);
int sdrhip_convert_i16_run(void *stream, const int16_t *d_in, float *d_out, int64_t n);
int sdrhip_scale_run(void *stream, float factor, const float *d_in, float *d_out, int64_t n);
/* fmDemod (Demod.hs:40-46).  y[k] for k in [k_begin,k_end); the sample preceding
 * k_begin is d_in[k_begin-1-in_base] when k_begin > in_base, else (last_re,last_im)
 * (the Pipe's carried state; (0,0) at stream start, Demod.hs:41). */
int sdrhip_fm_demod_run(void *stream, const float *d_in_iq, int64_t in_base, float *d_out,
                        int64_t k_begin, int64_t k_end, float last_re, float last_im);
/* Diagnostics: how many stream-API launches the general LDS-tiled kernels (kernels_split.hip) have served in
 * this process -- the tests use it to prove the tiled path, not the one-thread-per-output fallback, ran. */
long long sdrhip_debug_tiled_launches(void);

/* dcBlocker (c_sources/filter.c:152-161; Pipe dcBlockingFilter, Filter.hs:730-739) on device memory:
 * y[i] = (float)((double)(x[i] - x[i-1]) + 0.997 * (double)y[i-1]), x[-1] = last_sample, y[-1] = last_output.
 * Bit-identical to the sequential loop (speculative chunks, verified and settled on the device).
 * d_final receives {finalSample, finalOutput}.  d_workspace may be NULL (sequential walk); when given
 * it starts with three uint32 statistics {chunks left to the sequential settle pass, samples it rewrote,
 * chunks recomputed by the parallel repair rounds} -- all zero when every speculation converged.  run_in =
 * samples each chunk runs in before its first output (0 = default 12288; a shorter run-in does less
 * redundant work but leaves more chunks to settle; the result never depends on it).  Not in-place. */
size_t sdrhip_dc_blocker_workspace_bytes(int64_t n);
int sdrhip_dc_blocker_run(void *stream, const float *d_in, float *d_out, int64_t n, float last_sample,
                          float last_output, float *d_final, void *d_workspace, size_t workspace_bytes,
                          int run_in);

/* ---- the record seam on HOST vectors (hs_sources/SDR/Filter.hs:116-144) ---- */
/* The closures a Haskell constructor puts into the reference's own Filter / Decimator / Resampler records, so that
 * the reference's UNCHANGED Pipes (firFilter / firDecimator / firResampler, Filter.hs:532-727) drive the device:
 *   *_one   = the C SIMD kernel on one buffer: num outputs from in[0 ..] (FilterInternal.hs:66-71,172-177,335-342);
 *   *_cross = the sequential Haskell kernel on `drop i last ++ next` (decimate/filter/resampleCrossHighLevel,
 *             FilterInternal.hs:397-423).
 * HOST pointers, synchronous; n_* in elements (complex pairs for complex descriptors).  The resampler calls carry the
 * reference's state: _one takes the polyphase group and returns the group of the next output (what resampleAVXRR
 * returns), _cross takes the filter offset and returns the offset after the last output; negative = error. */
int sdrhip_filter_one(const sdrhip_filter *f, int num, const float *in, float *out);
int sdrhip_filter_cross(const sdrhip_filter *f, int num, const float *last, int n_last, const float *next, int n_next,
                        float *out);
int sdrhip_decimator_one(const sdrhip_decimator *d, int num, const float *in, float *out);
int sdrhip_decimator_cross(const sdrhip_decimator *d, int num, const float *last, int n_last, const float *next,
                           int n_next, float *out);
int sdrhip_resampler_one(const sdrhip_resampler *r, int group, int num, const float *in, int n_in, float *out);
int sdrhip_resampler_cross(const sdrhip_resampler *r, int filter_offset, int num, const float *last, int n_last,
                           const float *next, int n_next, float *out);

/* ---- the FM receiver chain (examples/fm/fm.hs:34-41) ----------------------- */
/* u8 IQ -> [convert] -> decimator -> fmDemod -> resampler -> symmetric filter
 * [-> * gain].  All four Pipes of fm.hs run with blockSizeOut = block, and the
 * source delivers `block`-sample buffers, so every stage's seams sit at
 * multiples of `block` of its own input index (block = 0: contiguous). */
typedef struct sdrhip_fm_chain sdrhip_fm_chain;
int sdrhip_fm_chain_create(sdrhip_fm_chain **c, int order, int decim_factor, const float *decim_taps,
                           int n_decim_taps, int interpolation, int decimation, const float *resamp_taps,
                           int n_resamp_taps, const float *audio_half_taps, int n_audio_half, float gain,
                           int64_t block);
void sdrhip_fm_chain_destroy(sdrhip_fm_chain *c);
/* Ownership rule for sharding: audio output q is owned by the shard in which its
 * receptive field STARTS.  Given a shard of input samples [s0,s1) of the global
 * stream (plus a right halo), report the owned range [*q0,*q1), and *halo = how
 * many samples past s1 the owned outputs read (the ntaps-1 overlap of every
 * stage composed).  total_in = length of the whole stream (outputs needing data
 * past it do not exist), or -1 for unbounded. */
int sdrhip_fm_chain_plan(const sdrhip_fm_chain *c, int64_t s0, int64_t s1, int64_t total_in,
                         int64_t *q0, int64_t *q1, int64_t *halo);
/* Number of audio outputs whose whole receptive field lies inside the first n_samples samples of the
 * stream, i.e. outputs [0, ready) can be computed from them.  A shard [s0,s1) can compute its outputs
 * [q0, min(q1, ready(s1))) before its halo has arrived. */
int64_t sdrhip_fm_chain_ready(const sdrhip_fm_chain *c, int64_t n_samples);
int64_t sdrhip_fm_chain_max_halo(const sdrhip_fm_chain *c);
size_t sdrhip_fm_chain_workspace_bytes(const sdrhip_fm_chain *c, int64_t n_in);
/* d_in_iq[0] is stream sample s0; n_in samples (shard + halo) are readable.
 * Writes audio outputs [q0,q1) to d_audio[0..q1-q0).  Asynchronous on `stream`. */
int sdrhip_fm_chain_run(sdrhip_fm_chain *c, void *stream, const uint8_t *d_in_iq, int64_t s0, int64_t n_in,
                        float *d_audio, int64_t q0, int64_t q1, void *d_workspace, size_t workspace_bytes);

/* One run with FIXED arguments (pointers, ranges) captured into a hipGraph: a launch-bound batch -- a 2^20-sample shard is
 * nine small kernels -- then costs one graph launch per pass.  Captured from the code path sdrhip_fm_chain_run takes (a
 * plain run precedes the capture: tap uploads and argument checks happen there).  Timing and pipelining must be off. */
typedef struct sdrhip_fm_graph sdrhip_fm_graph;
int sdrhip_fm_chain_graph_create(sdrhip_fm_graph **g, sdrhip_fm_chain *c, const uint8_t *d_in_iq, int64_t s0, int64_t n_in,
                                 float *d_audio, int64_t q0, int64_t q1, void *d_workspace, size_t workspace_bytes);
int sdrhip_fm_chain_graph_launch(sdrhip_fm_graph *g, void *stream);
void sdrhip_fm_chain_graph_destroy(sdrhip_fm_graph *g);

/* Two runs in flight (round 4; default off).  The receiver's successive batches are independent given their raw input
 * (examples/fm/fm.hs:34-41: no stage carries a RESULT from one batch into the next -- only input samples), so with on = 1
 * consecutive sdrhip_fm_chain_run calls alternate between two internal HIP streams and the two halves of the workspace: the
 * memory-heavy tail kernels of run k execute beside the power-bound decimator of run k+1.  Contract while it is on:
 *   - sdrhip_fm_chain_workspace_bytes returns twice the single-run size; pass that much to every run;
 *   - run k starts after everything queued on the caller's `stream` at the time of the call (the producer of its input);
 *   - when the call for run k returns, `stream` has been made to wait for run k-1 -- NOT for run k.  So consecutive runs
 *     must write different audio buffers (double-buffer them), and run k's audio may be consumed on `stream` after the call
 *     for run k+1, or after sdrhip_fm_chain_join(c, stream), which makes `stream` wait for every run still in flight;
 *   - the same holds for run k's INPUT: work queued on `stream` between the calls for run k and run k+1 is ordered after run k-1
 *     only, so the buffer run k reads may be overwritten on `stream` after the call for run k+1 (or after the join) -- double-buffer
 *     the input like the audio;
 *   - results are bit-identical to on = 0 (the same kernels on the same ranges); hipGraph capture needs on = 0.
 * sdrhip_fm_chain_set_overlap drains the runs in flight (host-side wait) before it changes the mode. */
int sdrhip_fm_chain_set_overlap(sdrhip_fm_chain *c, int on);
int sdrhip_fm_chain_join(sdrhip_fm_chain *c, void *stream);
/* fmDemod -> resampler -> audio filter (* gain) as ONE kernel (the demodulated and resampled streams never leave LDS)
 * when the chain has the FM receiver's shape (3/10 resampler with 64-float groups, 64 half-tap symmetric filter, AVX
 * order, buffers longer than one tile).  mode 0 = never (the three stage kernels), 1 = always, 2 = auto (default): only
 * for runs of at most 768 audio outputs (pushes of one or two 8192-sample source blocks), where one launch replaces
 * three; longer runs are faster on the stage kernels (every stage is VALU-bound, and a one-tile run is one workgroup).
 * Same bits. */
int sdrhip_fm_chain_set_fused_tail(sdrhip_fm_chain *c, int mode);
/* The WHOLE chain (convert + decimator -> fmDemod -> resampler -> audio filter * gain) as ONE kernel for launch-bound runs
 * -- BASELINE configs[4]'s 2^20-sample shard, a push of a few source blocks -- when the chain has the FM receiver's shape
 * (decimation 8 with 128 padded taps, 3/10 resampler with 64-float groups, 64 half-tap symmetric filter, AVX order, 16-byte
 * aligned input whose first sample index is a multiple of 8).  Nothing goes through the workspace; every workgroup recomputes
 * the overlap of its tile (the first stage is computed ~1.9 times over), which pays exactly while a run is bound by launch
 * latency, not by arithmetic.  mode 0 = never, 1 = always, 2 = auto (default): runs of at most max_outputs audio outputs
 * (0 = the built-in bound, 159 * 1728 outputs = ~7.3 M input samples, where the stage kernels catch up: tools/launch_sweep.py).  tile_outputs: audio outputs per workgroup (multiple of 3, at most
 * 159; 0 = chosen from the size of the run: about one workgroup per CU).  Same bits as the stage kernels
 * (examples/fm/fm.hs:34-41; c_sources/decimate.c:105-113, resample.c:70-87, filter.c:60-68; Demod.hs:21-46). */
int sdrhip_fm_chain_set_small_chain(sdrhip_fm_chain *c, int mode, int64_t max_outputs, int tile_outputs);
/* launches of that kernel so far, process-wide (tests assert that this path, not the stage kernels, ran) */
long long sdrhip_debug_small_chain_launches(void);
/* launches of the thread-per-polyphase-cycle resampler (real I/D with an odd decimation: 2/3, 5/7, ...; any filter length),
 * process-wide (tests assert that this kernel, not the lane-split one, served those ratios) */
long long sdrhip_debug_resample_cycle_launches(void);
/* launches of the real decimator kernel for factors 2 / 4 / 8 / 16 (kernels_decimate_real.hip), process-wide */
long long sdrhip_debug_decimate_real16_launches(void);
/* Which kernel serves the decimate-by-8, 128-tap, AVX-order first stage: 0 = the LDS-tiled kernel everywhere, 1 = the
 * register-resident systolic kernel (kernels_systolic.hip) wherever its shape fits, in its round-5 form (non-temporal loads, seam
 * fix-up as a second launch), 2 (default; SDRHIP_SYSTOLIC sets the initial value) = the library's own choice: the systolic kernel
 * with the seam fix-up's workgroups inside its launch and plain / non-temporal loads by launch size (tools/route_sweep_fine.py).
 * Results are identical.
 * sdrhip_debug_systolic_launches: launches the systolic kernel has served, process-wide; sdrhip_debug_systolic_plan: the strip cut of a
 * launch of `count` outputs (host arithmetic only): strips [0, nwhole) take the unguarded body. */
void sdrhip_debug_set_systolic(int mode);
long long sdrhip_debug_systolic_launches(void);
void sdrhip_debug_systolic_plan(int count, int *nstrips, int *nwhole);
/* fmDemod inside the resampler's tile loader for large batches (>= 2^18 resampler outputs per run): the demodulated stream
 * never makes its round trip through HBM (12 B per decimated sample less traffic); the per-stage timing then books the pair
 * under `resample` (and reports 0 for `fm_demod`).  Same bits.  ON by default since round 4 (SDRHIP_FUSE_DEMOD=0 turns it off):
 * the pair takes 0.269 ms against 0.159 + 0.089 for the two stage kernels -- the arithmetic is the same -- but the chip runs at its
 * power cap and 0.54 GB less traffic per pass leaves the decimator 4 % more clock: the whole pass gains 1.0 %. */
int sdrhip_fm_chain_set_demod_fusion(sdrhip_fm_chain *c, int enable);
/* Per-stage timing with HIP events recorded around each stage's kernels on the stream they are
 * launched on; stages {decimate(+seam fix-up), fmDemod, resample, filter(+gain), fused tail (the three in one kernel),
 * whole chain in one kernel (sdrhip_fm_chain_set_small_chain)}.
 * read_timing waits for the recorded runs, returns the SUM of elapsed ms per stage over
 * `*runs` runs and resets the recorder. */
int sdrhip_fm_chain_enable_timing(sdrhip_fm_chain *c, int enable);
int sdrhip_fm_chain_read_timing(sdrhip_fm_chain *c, double ms_sum[6], int *runs);

/* ---- multi-GPU: the ntaps-1 halo exchange of the sharded chain ---- */
/* The reference has no multi-device code; these entry points are what a sharded host (one thread or process per
 * GPU -- e.g. eight Haskell pipelines of examples/fm/fm.hs:34-41, one per device) binds to.  Rank r owns the samples
 * [r*S, (r+1)*S) of every super-block in a device buffer laid out [shard | halo region]; one exchange sends the FIRST
 * `bytes` of the shard to the LEFT neighbour (rank r-1, wrapping) and receives the right neighbour's head into the halo
 * region.  Transport: RCCL point-to-point (ncclGroupStart; ncclSend; ncclRecv; ncclGroupEnd) enqueued on the caller's HIP
 * stream -- librccl.so.1 is loaded on first use, the library has no link-time dependency on it -- or, for the devices of
 * ONE process, hipMemcpyPeerAsync.  The message is ~8 KB: latency-bound, so no collective is involved. */
typedef struct sdrhip_comm sdrhip_comm;
#define SDRHIP_COMM_ID_BYTES 128
/* rank 0 creates the rendezvous id and hands it to the other ranks by any out-of-band means (file, socket, MPI ...) */
int sdrhip_comm_get_unique_id(void *id /* SDRHIP_COMM_ID_BYTES */);
/* one communicator per GPU process / thread; binds the CURRENT device (sdrhip_set_device first).  Collective: every
 * rank must call it. */
int sdrhip_comm_init_rank(sdrhip_comm **c, int nranks, int rank, const void *id);
/* all `ndev` devices from ONE process: comms[i] is rank i on devices[i].  transport: SDRHIP_TRANSPORT_RCCL or
 * SDRHIP_TRANSPORT_PEER_COPY (hipMemcpyPeerAsync; no RCCL needed). */
#define SDRHIP_TRANSPORT_RCCL 1
#define SDRHIP_TRANSPORT_PEER_COPY 2
int sdrhip_comm_init_local(sdrhip_comm **comms, int ndev, const int *devices, int transport);
void sdrhip_comm_destroy(sdrhip_comm *c);
int sdrhip_comm_rank(const sdrhip_comm *c);
int sdrhip_comm_size(const sdrhip_comm *c);
const char *sdrhip_comm_transport(const sdrhip_comm *c);   /* "rccl" | "peer-copy" */
/* One rank's exchange, asynchronous on `stream` (the compute stream: kernels queued after it see the halo).  d_send /
 * d_recv are device pointers on this rank's GPU.  RCCL transport only (a per-rank call cannot see the peer's memory). */
int sdrhip_halo_exchange(sdrhip_comm *c, void *stream, const void *d_send, void *d_recv, size_t bytes);
/* All ranks of a single-process communicator at once: rank i sends d_send[i] and receives into d_recv[i] on streams[i].
 * RCCL: one group over every rank; peer copy: stream i waits for the event of rank i+1's stream (the head must have been
 * produced) and pulls it with hipMemcpyPeerAsync. */
int sdrhip_halo_exchange_all(sdrhip_comm *const *comms, int ndev, void *const *streams, const void *const *d_send,
                             void *const *d_recv, size_t bytes);
/* The chain's own message: head = d_buf[0 .. 2*halo), halo region = d_buf + 2*shard_samples, halo =
 * sdrhip_fm_chain_halo_samples(chain) (max_halo rounded up to 8 samples: the same on every rank). */
int64_t sdrhip_fm_chain_halo_samples(const sdrhip_fm_chain *c);
int sdrhip_fm_chain_halo_exchange(const sdrhip_fm_chain *chain, sdrhip_comm *comm, void *stream, uint8_t *d_buf,
                                  int64_t shard_samples);
/* The halos of `count` consecutive super-blocks in ONE message pair (round 5): the rank's shard of super-block k lies at
 * d_buf + k * row_bytes (head at +0, halo region at + 2 * shard_samples, as above).  The heads are gathered into d_staging
 * (hipMemcpy2DAsync), one ncclSend / ncclRecv of count * 2 * halo bytes moves them, and the received ones are scattered into the
 * halo regions -- all on `stream`.  A launch-bound shard (2^20 samples: ~11 us per pass) cannot hide a point-to-point round trip
 * per pass; with count passes per exchange the round trip is paid once per count passes at the price of count super-blocks
 * being resident before the first is processed.  d_staging:
 * sdrhip_fm_chain_halo_staging_bytes(chain, count) bytes of device memory (a send and a receive half). */
size_t sdrhip_fm_chain_halo_staging_bytes(const sdrhip_fm_chain *chain, int count);
int sdrhip_fm_chain_halo_exchange_batch(const sdrhip_fm_chain *chain, sdrhip_comm *comm, void *stream, uint8_t *d_buf,
                                        int64_t shard_samples, size_t row_bytes, int count, void *d_staging);

/* Host-block streaming front end of the chain: u8 IQ source blocks in (host memory, `block` samples
 * each or a whole multiple), audio blocks of exactly block_size_out floats out -- the five middle
 * stages of examples/fm/fm.hs:34-41 as ONE operator with every intermediate resident in HBM.  Pinned
 * staging in slots = submissions in flight: two when max_block_samples is too large to run in place (results lag one push;
 * H2D / compute / D2H on three HIP streams), four when even the largest push runs on one stream -- up to 200 source blocks: one compute
 * stream per slot; a lone source block is read by the kernel straight from the pinned buffer over PCIe, anything larger (a push of several
 * blocks, or pushes that piled up while the GPU was busy) crosses the link ONCE by a copy on the slot's stream and is then ONE kernel on
 * device memory; the audio is written to the pinned result buffer by the kernel either way (SDRHIP_STAGE_SAMPLES moves the bound) --
 * and results lag three pushes (flush drains; SDRHIP_STREAM_SLOTS=2..4 overrides the count).
 * The chain must outlive the stream and must not be run concurrently by another caller. */
typedef struct sdrhip_fm_stream sdrhip_fm_stream;
int sdrhip_fm_stream_create(sdrhip_fm_stream **st, sdrhip_fm_chain *chain, int max_block_samples, int block_size_out);
void sdrhip_fm_stream_destroy(sdrhip_fm_stream *st);
/* returns the number of complete audio blocks ready to pop (>= 0) or a negative error */
int sdrhip_fm_stream_push(sdrhip_fm_stream *st, const uint8_t *iq, int n_samples);
/* Zero-copy variant: the pinned staging buffer (2*max_block_samples bytes) the NEXT push will upload
 * from.  Let the source (e.g. the RTL-SDR read of SDR/RTLSDRStream.hs) write into it, then push that
 * same pointer: the host-side memcpy is skipped.  Valid until that push; NULL on error.  (With
 * coalescing it points just past the samples already staged; there is always room for
 * max_block_samples more.) */
uint8_t *sdrhip_fm_stream_input_buffer(sdrhip_fm_stream *st);
int sdrhip_fm_stream_flush(sdrhip_fm_stream *st);
/* Audio blocks ready to pop, after collecting (without waiting) every in-flight submission the GPU has finished: lets a
 * real-time caller fetch the audio of the push it just made some tens of microseconds later instead of at its next push
 * (pushes themselves collect finished work too; flush waits for all of it). */
int sdrhip_fm_stream_poll(sdrhip_fm_stream *st);
/* Latency / throughput knob: stage pushes in the pinned buffer and submit them to the GPU together once
 * `samples` samples (a multiple of the chain's block; 0 = every push, the default) have accumulated, or on
 * flush.  One 8192-sample push costs ~8-9 us (one kernel launch, four pushes in flight) whatever its size, so a caller
 * that must keep the reference's block size (fm.hs:17) gets ~6x the throughput from coalesce = 16 blocks, at 16 blocks
 * of latency; the audio blocks are the same.  Call with nothing staged. */
int sdrhip_fm_stream_set_coalesce(sdrhip_fm_stream *st, int samples);
/* The same knob turned by the GPU itself: with max_samples > 0 (a multiple of the chain's block, at least two pushes) a push
 * is submitted at once while the slot its submission would move on to is free, and staged behind the earlier ones while that
 * slot is still running -- up to max_samples, then the push waits.  A source slower than the GPU (a radio) sees every push
 * go out immediately; a faster one (file replay) sees its pushes leave as fewer, larger launches.  Which push returns which
 * audio block then depends on timing; the blocks themselves do not.  0 switches it off.  Call with nothing staged. */
int sdrhip_fm_stream_set_adaptive(sdrhip_fm_stream *st, int max_samples);
int sdrhip_fm_stream_pop(sdrhip_fm_stream *st, float *out, int capacity);
/* Checkpoint / resume.  Between two pushes the operator's state is the stream position, the last ~4k input samples and
 * the audio not yet popped (the reference keeps the equivalent in Pipe closures: overlap remainder, resampler phase, last
 * demod sample, output fill level -- Filter.hs:536-727, Demod.hs:21-38); everything else is a closed form of the position.
 * state_bytes: drains the operator (like flush) and returns the exact size the save that follows needs (0 = the drain failed);
 * save: drains the operator and writes the state (*used = its size);
 * restore: into a freshly created stream over a chain of the same taps and block sizes; returns the number of audio blocks
 * ready to pop.  A restored stream fed the remaining samples yields the audio the uninterrupted stream would have. */
size_t sdrhip_fm_stream_state_bytes(sdrhip_fm_stream *st);   /* not const: it drains (submits, waits, moves results) */
int sdrhip_fm_stream_save(sdrhip_fm_stream *st, void *buf, size_t capacity, size_t *used);
int sdrhip_fm_stream_restore(sdrhip_fm_stream *st, const void *buf, size_t bytes);

/* ---- spectrum path (hs_sources/SDR/FFT.hs:44-168) on hipFFT ---- */
/* fftw' / fftw (complex-to-complex forward DFT of n Complex Double), fftwReal' / fftwReal (n Double -> n/2+1 bins) and
 * fftwParallel (several buffers in flight: here a batched plan).  Double precision, unnormalised, forward sign
 * exp(-2 pi i jk/n): FFTW's conventions and bin order.  Tolerance contract (a different summation tree from FFTW's), not
 * bit parity.  libhipfft.so.0 is loaded on first use. */
typedef struct sdrhip_fft sdrhip_fft;
int sdrhip_fft_create(sdrhip_fft **f, int n, int real_input, int batch);
void sdrhip_fft_destroy(sdrhip_fft *f);
int sdrhip_fft_size(const sdrhip_fft *f);
int sdrhip_fft_bins(const sdrhip_fft *f);                                  /* n, or n/2 + 1 for real input */
/* HOST vectors, synchronous: in = batch x n complex doubles (interleaved) or batch x n doubles; out = batch x bins complex */
int sdrhip_fft_run(sdrhip_fft *f, const double *in, double *out);
/* DEVICE vectors, asynchronous on `stream` */
int sdrhip_fft_run_device(sdrhip_fft *f, void *stream, const double *d_in, double *d_out);

/* ------------------------------------------------------------------------ */
/* (3) Pipe operators on host blocks                                        */
/* ------------------------------------------------------------------------ */
typedef struct sdrhip_pipe sdrhip_pipe;
/* The constructors take ownership of nothing: the descriptor must outlive the pipe. */
int sdrhip_pipe_fir_filter(sdrhip_pipe **p, const sdrhip_filter *f, int block_size_out);          /* firFilter   */
int sdrhip_pipe_fir_decimator(sdrhip_pipe **p, const sdrhip_decimator *d, int block_size_out);    /* firDecimator */
int sdrhip_pipe_fir_resampler(sdrhip_pipe **p, const sdrhip_resampler *r, int block_size_out);    /* firResampler */
int sdrhip_pipe_fm_demod(sdrhip_pipe **p);                                                         /* fmDemod      */
int sdrhip_pipe_dc_blocker(sdrhip_pipe **p);                                     /* dcBlockingFilter, Filter.hs:730-739 */
/* Feed one upstream block (n elements: floats, or complex pairs for complex
 * stages).  Returns the number of complete output blocks now ready (>= 0) or a
 * negative error.  A block shorter than numCoeffs is the reference's
 * `assert "filter 1"` failure -> SDRHIP_ERR_ARG. */
int sdrhip_pipe_push(sdrhip_pipe *p, const float *block, int n);
/* Up to four submissions are in flight (SDRHIP_STREAM_SLOTS=2..4), so results lag up to three pushes behind: flush waits
 * for the in-flight blocks and returns the number of complete output blocks ready. */
/* Throughput knobs of the filter / decimator / resampler pipes (results never depend on them):
 *  - set_coalesce(blocks): equal-sized pushes are staged in the pinned buffer and submitted `blocks` at a
 *    time (one upload, one run over the batch with its interior seams, one download); a push of another
 *    size ends the uniform run.  0 / 1 = every push on its own.  blocks > 1 takes precedence over adaptive submission
 *    (exactly `blocks` per submission, whatever the GPU is doing).
 *  - set_adaptive(max_blocks): the same, decided by the GPU: a push is submitted at once while the slot its submission would
 *    move on to is free, and staged behind the earlier ones while that slot is still running (up to max_blocks, and never more
 *    than 4 MiB of staged input (SDRHIP_ADAPTIVE_BYTES; batches past 512 KiB go through the copy engines instead of being read
 *    in place over PCIe); a source slower than the GPU sees every push go out immediately, a faster
 *    one sees fewer, larger launches.  Which push returns which block then depends on timing.  0 = off.
 *  - input_buffer(n): the pinned staging memory the next push of n elements will be uploaded from; fill it
 *    and push that pointer to skip the host-side copy. */
int sdrhip_pipe_set_coalesce(sdrhip_pipe *p, int blocks);
int sdrhip_pipe_set_adaptive(sdrhip_pipe *p, int max_blocks);
float *sdrhip_pipe_input_buffer(sdrhip_pipe *p, int n);
int sdrhip_pipe_flush(sdrhip_pipe *p);
/* output blocks ready to pop, after collecting (without waiting) every in-flight submission the GPU has finished */
int sdrhip_pipe_poll(sdrhip_pipe *p);
/* Pop one ready block into out (capacity in elements); returns its length, 0 if none. */
int sdrhip_pipe_pop(sdrhip_pipe *p, float *out, int capacity);
/* Checkpoint / resume, as sdrhip_fm_stream_save / _restore: save drains the pipe (like flush) and writes its state -- position,
 * the last few input elements, the carried fmDemod sample / dcBlocker pair, the output not yet popped; restore loads it into a
 * freshly created pipe of the same kind over a descriptor of the same taps; returns the blocks ready to pop. */
/* state_bytes drains the pipe (like flush) and returns the exact size the save that follows needs (0 = the drain failed). */
size_t sdrhip_pipe_state_bytes(sdrhip_pipe *p);   /* not const: it drains (submits, waits, moves results) */
int sdrhip_pipe_save(sdrhip_pipe *p, void *buf, size_t capacity, size_t *used);
int sdrhip_pipe_restore(sdrhip_pipe *p, const void *buf, size_t bytes);
void sdrhip_pipe_destroy(sdrhip_pipe *p);

#ifdef __cplusplus
}
#endif
#endif /* SDR_HIP_H */
