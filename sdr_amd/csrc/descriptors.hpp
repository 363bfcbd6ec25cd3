// descriptors.hpp -- the device-side twins of the reference's Filter / Decimator /
// Resampler records (hs_sources/SDR/Filter.hs:116-144) with their taps prepared the
// way the reference's constructors prepare them (A10: Filter.hs:146-148,163-175,
// 234-245,277-290,317-331,408-425; FilterInternal.hs:277-319).
#pragma once
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace sdrhip {

struct FirDesc {
    int order = SDRHIP_ORDER_AVX;
    bool cplx = false;
    bool sym = false;
    int factor = 1;       // decimation (1 = filter)
    int Lp = 0;           // numCoeffsF / numCoeffsD: what the Pipe sees
    int lanes = 8;        // real data lane count
    ComplexOrder corder = CO_L4;
    int ntaps_kernel = 0; // length of d_taps as the kernel consumes it
    // Device copies are made on first use (ensure_device), so descriptors -- and the
    // planning arithmetic built on them -- can be created on a host without a GPU.
    mutable int device = -1;           // the device the taps were uploaded to (-1: not yet)
    mutable float* d_taps = nullptr;   // real: padded plain (or half for sym); complex RC: duplicated
    mutable float* d_cross = nullptr;  // Lp plain taps for the sequential Cross outputs
    mutable float* d_plain = nullptr;  // padded plain taps (aliases d_cross)
    mutable float* d_scaled = nullptr; // complex stages: plain taps / 128 for the u8-fused tiled kernel (null: some tap too small to scale exactly)
    std::vector<float> h_scaled;
    std::vector<float> h_plain;        // Lp plain taps (sym: c ++ reverse c)
    std::vector<float> h_kernel;       // what d_taps holds
    int ensure_device() const;
    ~FirDesc();
};

struct ResampDesc {
    int order = SDRHIP_ORDER_AVX;
    bool cplx = false;
    int I = 1, D = 1;
    int Lp = 0;           // numCoeffsR = roundUp(ntaps, I * simd)
    int lanes = 8;
    ComplexOrder corder = CO_X4;
    int ntaps = 0;        // unpadded
    int num_coeffs = 0;   // longest group, unpadded
    int num_groups = 0;
    int row_stride = 0;   // padded group length
    int nloop = 0;        // floats the SIMD loop actually walks
    std::vector<int> increments, offsets, lut;  // lut: filter offset -> group
    std::vector<float> h_groups, h_plain;
    mutable int device = -1;           // the device the taps were uploaded to (-1: not yet)
    mutable float* d_groups = nullptr;
    mutable float* d_plain = nullptr;
    mutable int* d_ext = nullptr;      // more than 64 groups: prefix sums of the increments + filter offsets (kernels.hpp ResampTable::ext)
    int ensure_device() const;
    int64_t in_offset(int64_t m) const { return ceil_div64(m * (int64_t)D, I); }
    int filter_offset(int64_t m) const { return (int)(in_offset(m) * I - m * (int64_t)D); }
    int group(int64_t m) const { return lut[filter_offset(m)]; }
    ~ResampDesc();
};

// polyphase split, FilterInternal.hs:297-319
void prepare_coeffs(int n, int I, int D, const float* coeffs, int ncoeffs, int& num_coeffs, int& row_stride,
                    std::vector<int>& increments, std::vector<int>& offsets, std::vector<float>& groups);

int fir_create(FirDesc* d, int order, bool cplx, int factor, const float* coeffs, int ncoeffs);
int fir_sym_create(FirDesc* d, int order, int factor, const float* half, int nhalf);
// gain: applied to every output after the filter's own rounding (a separate f32 multiply,
// exactly what `P.map (VG.map (* g))` after the Pipe computes); 1.0f = none.
int fir_run(const FirDesc* d, hipStream_t s, const void* d_in, bool in_u8, int64_t in_base, float* d_out,
            int64_t k_begin, int64_t k_end, int64_t seam_block, float gain = 1.0f);

int resamp_create(ResampDesc* r, int order, bool cplx, int I, int D, const float* coeffs, int ncoeffs);
// out_block: the Pipe's output block size (blockSizeOut; 0 = unbounded) -- it decides one corner case of the seam
// classification (kernels.hpp: late_output_is_one)
int resamp_run(const ResampDesc* r, hipStream_t s, const float* d_in, int64_t in_base, float* d_out,
               int64_t k_begin, int64_t k_end, int64_t seam_block, int64_t out_block = 0);
// fmDemod + resampler: d_iq = the decimator output whose phase step (fmDemod, Demod.hs:21-46) is input `in_base`, y_count
// inputs; d_in = the buffer the stand-alone fmDemod would fill (abi_device.cpp)
int resamp_run_demod(const ResampDesc* r, hipStream_t s, const float* d_iq, bool iq_has_prev, int64_t y_count, const float* d_in,
                     int64_t in_base, float* d_out, int64_t k_begin, int64_t k_end, int64_t seam_block, int64_t out_block,
                     bool* demod_fused);

}  // namespace sdrhip

struct sdrhip_filter : sdrhip::FirDesc {};
struct sdrhip_decimator : sdrhip::FirDesc {};
struct sdrhip_resampler : sdrhip::ResampDesc {};
