/* sdr_hip_bench.h -- measurement utilities of bench.py and tools/ (libsdr_hip_bench.so; NOT part of the product library).
 *
 * Round 6: these lived in libsdr_hip.so until the product library was reduced to the product.  They are written against
 * include/sdr_hip.h only: ceilings of the memory system measured next to the kernels, and C timing loops over the host-block
 * operators (what a compiled caller pays per push; a Python loop adds 10-20 us per call). */
#ifndef SDR_HIP_BENCH_H
#define SDR_HIP_BENCH_H

#include "sdr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A streaming kernel with the traffic shape of the cfloat decimate-by-8 kernel (bytes_in read, bytes_in / 8 written;
 * bytes_in a multiple of 32 KiB), with plain or non-temporal loads, and a float4 copy: the ceilings the memory system
 * of this box delivers in this process, measured next to the kernel. */
int sdrhip_bench_stream_8to1(void *stream, const void *d_in, void *d_out, size_t bytes_in, int non_temporal);
int sdrhip_bench_copy(void *stream, const void *d_in, void *d_out, size_t bytes);
/* the same copy (4 x 16 bytes in flight per thread) with plain or non-temporal loads and stores */
int sdrhip_bench_copy2(void *stream, const void *d_in, void *d_out, size_t bytes, int non_temporal);
/* Timing loops over the host-block operators, written against this header only: what a compiled caller pays per push
 * (a Python loop adds 10-20 us per call).  fm_stream: `pushes` pushes of n_samples u8 IQ samples (zero_copy: through
 * sdrhip_fm_stream_input_buffer), every audio block popped, coalesce_samples < 0 = adaptive submission with that cap, 1 = adaptive
 * submission off (every push its own launch), 0 = the stream's default; pipe:
 * pushes of n elements into an existing pipe. */
struct sdrhip_pipe;
int sdrhip_bench_fm_stream(sdrhip_fm_chain *chain, int n_samples, int pushes, int zero_copy, int coalesce_samples,
                           double *samples_per_s, long long *audio_blocks);
/* push-to-audio latency of sdrhip_fm_stream (bench_host.cpp): out[0..4] = p50, p99, max, mean latency and the mean duration of
 * the push call, microseconds; pace_us > 0 paces the pushes like a real-time source (6400 us per 8192 samples at 1.28 MS/s) */
int sdrhip_bench_fm_stream_latency(sdrhip_fm_chain *chain, int n_samples, int pushes, double pace_us, int adaptive_off, double *out);
int sdrhip_bench_pipe(struct sdrhip_pipe *p, int n, int floats_per_element, int block_size_out, int pushes, int zero_copy,
                      double *elements_per_s);
/* the FM receiver composed of four Level-1 Pipes the way examples/fm/fm.hs:34-41 composes it (firDecimator -> fmDemod ->
 * firResampler -> firFilter, each re-blocking to `block` elements), fed `pushes` cfloat blocks of `block` samples; every output
 * block of a stage is popped and pushed into the next one by the loop.  *samples_per_s = source samples per second. */
int sdrhip_bench_fm_pipes(const sdrhip_decimator *dec, const sdrhip_resampler *res, const sdrhip_filter *fil, int block, int pushes,
                          double *samples_per_s, long long *audio_blocks);


#ifdef __cplusplus
}
#endif

#endif
