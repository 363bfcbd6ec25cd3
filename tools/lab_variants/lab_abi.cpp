// lab_abi.cpp -- C entry points of liblab.so (tools/lab_variants/build_lab.py): raw device pointers in, one launch out, on the
// NULL stream.  check_lab.py compares every variant with the product library bit for bit.
#include "lab.hpp"

using namespace sdrhip;

extern "C" {

int sdrlab_resample3_systolic(const float* d_in, long long pos, int ncycles, long long avail_total, const float* d_groups, int row_stride, float* d_out)
{
    set_resample_systolic(1);
    return launch_resample3_systolic(nullptr, d_in, pos, ncycles, avail_total, d_groups, row_stride, d_out) ? 1 : 0;
}

int sdrlab_resample3_demod_stream(int mode, const float* d_iq, long long pos, int ncycles, int iq_has_prev, long long y_count, const float* d_groups,
                                  int row_stride, float* d_out, float* d_y, long long y_abs0, int yseam, int ykeep, int nedge)
{
    set_resample_demod_stream(mode);
    return launch_resample3_demod_stream(nullptr, d_iq, pos, ncycles, iq_has_prev != 0, y_count, d_groups, row_stride, d_out, d_y, y_abs0, yseam, ykeep,
                                         nedge) ? 1 : 0;
}

int sdrlab_decimate_demod_systolic(const unsigned char* d_in, long long in_base, long long kd0, long long kd1, long long ky0, const float* d_scaled_taps,
                                   const float* d_plain_taps, int P, int last_tap_zero, long long seam_block, float* d_y)
{
    return launch_decimate_demod_systolic(nullptr, d_in, in_base, kd0, kd1, ky0, d_scaled_taps, d_plain_taps, P, last_tap_zero != 0, seam_block, d_y) ? 1 : 0;
}

void sdrlab_fm_demod_form(int form, const float* d_in_iq, float* d_out, long long count, int has_prev, float last_re, float last_im)
{
    launch_fm_demod_form(nullptr, form, d_in_iq, d_out, count, has_prev != 0, last_re, last_im);
}

int sdrlab_sync(void) { return (int)hipDeviceSynchronize(); }

}
