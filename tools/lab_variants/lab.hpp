// lab.hpp -- declarations of the lab variants' launchers (moved out of sdr_amd/csrc/kernels.hpp in round 6, when the variants
// left the product library).  The lab is compiled with -Dsdrhip=sdrlab_ns: the product's headers (kernels.hpp: Geom, ResampTable;
// the device code of demod_forms.hpp) are reused under another namespace, so liblab.so and libsdr_hip.so can live in one process.
#pragma once
#include "kernels.hpp"

namespace sdrhip {

// resample_systolic.hip (round 4): the whole cycles of a 3/10 launch (three 64-float groups, increments {4,3,3}, AVX order) by a
// register-resident systolic walk; cycle c starts at d_in[pos + 10c] and yields d_out[3c .. 3c + 2].  false = not taken.
bool launch_resample3_systolic(hipStream_t s, const float* d_in, int64_t pos, int ncycles, int64_t avail_total, const float* d_groups,
                               int row_stride, float* d_out);
void resample_systolic_plan(int ncycles, int64_t avail_total, int* nstrips, int* nwhole);
long long resample_systolic_launch_count();
void set_resample_systolic(int on);
// resample_demod_stream.hip (round 5): fmDemod + the whole cycles of a 3/10 launch as a streaming kernel.  d_iq: decimator output,
// sample 0 = input 0 of the launch (y_count of them; d_iq[-2..-1] exists when iq_has_prev); cycle c starts at input pos + 10 c; d_y
// receives the phases other kernels still read (within ykeep of a multiple of yseam in absolute position y_abs0 + n, and the first /
// last nedge).  false = switched off or too short a run, nothing launched.
bool launch_resample3_demod_stream(hipStream_t s, const float* d_iq, int64_t pos, int ncycles, bool iq_has_prev, int64_t y_count,
                                   const float* d_groups, int row_stride, float* d_out, float* d_y, int64_t y_abs0, int yseam, int ykeep,
                                   int nedge);
void set_resample_demod_stream(int on);   // 0 off, 1 runs long enough to stream, 2 every run it can take, n > 2 every run cut for n workgroups; 1000 + m: LDS-DMA prefetch
int resample_demod_stream_mode();
long long resample_demod_stream_launch_count();
void resample_demod_stream_plan(int ncycles, int cus, int* ntiles, int* tiles_per_wg, int* grid);
// decimate_demod_systolic.hip (round 4): K2 + K3 in one launch: decimator outputs [kd0, kd1) demodulated in place, y[k] for k in
// [ky0, kd1) stored at d_y[k - ky0]
bool launch_decimate_demod_systolic(hipStream_t s, const uint8_t* d_in, int64_t in_base, int64_t kd0, int64_t kd1, int64_t ky0,
                                    const float* d_scaled_taps, const float* d_plain_taps, int P, bool last_tap_zero, int64_t seam_block,
                                    float* d_y);
// demod_forms.hip: the stand-alone fmDemod kernel in each of the five restatements (0 ternaries, 1 selects, 2 common case + wave
// vote, 3 the same with the LDS table = the product's, 4 packed pairs)
void launch_fm_demod_form(hipStream_t s, int form, const float* d_in_iq, float* d_out, int64_t count, bool has_prev, float last_re, float last_im);

}  // namespace sdrhip
