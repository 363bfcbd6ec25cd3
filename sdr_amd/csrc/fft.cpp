// fft.cpp -- the spectrum path next to the hot path: the reference's SDR.FFT (hs_sources/SDR/FFT.hs:44-168)
//     fftw' / fftw            complex-to-complex forward DFT of n Complex Double   (planDFT1d .. Forward)
//     fftwReal' / fftwReal    real-to-complex DFT of n Double -> n/2 + 1 bins        (planDFTR2C1d)
//     fftwParallel            the same c2c DFT, several buffers in flight            (a thread pool there; a BATCHED plan here)
// on hipFFT: double precision (the reference computes in Double), unnormalised, forward sign exp(-2 pi i jk/n) -- FFTW's
// conventions, so a waterfall / spectrum plot (SDR.Plot) sees the same bins.  This is floating-point work with a different
// summation tree from FFTW's: the contract is a tolerance (tests/test_gpu_fft.py: 1e-11 of the largest bin), not bit parity.
// libhipfft.so is bound at run time like RCCL, so libsdr_hip.so keeps loading where it is absent.
#include <dlfcn.h>
#if __has_include(<hipfft/hipfft.h>)
#include <hipfft/hipfft.h>
#else
// hipFFT's development headers are absent: the library is bound at run time anyway, and these are the only declarations of
// hipfft.h this file uses
typedef struct hipfftHandle_t* hipfftHandle;
typedef enum { HIPFFT_SUCCESS = 0 } hipfftResult;
typedef enum { HIPFFT_Z2Z = 0x69, HIPFFT_D2Z = 0x6a } hipfftType;
typedef double2 hipfftDoubleComplex;
typedef double hipfftDoubleReal;
#define HIPFFT_FORWARD -1
#endif
#include <string.h>

#include <mutex>

#include "common.hpp"

using namespace sdrhip;

namespace {

struct HipFft {
    void* handle = nullptr;
    hipfftResult (*PlanMany)(hipfftHandle*, int, int*, int*, int, int, int*, int, int, hipfftType, int) = nullptr;
    hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
    hipfftResult (*ExecZ2Z)(hipfftHandle, hipfftDoubleComplex*, hipfftDoubleComplex*, int) = nullptr;
    hipfftResult (*ExecD2Z)(hipfftHandle, hipfftDoubleReal*, hipfftDoubleComplex*) = nullptr;
    hipfftResult (*Destroy)(hipfftHandle) = nullptr;
    std::string why;
};

HipFft* hipfft()
{
    static HipFft f;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* n : {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0"}) {
            f.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (f.handle) break;
        }
        if (!f.handle) {
            const char* e = dlerror();        // one call: it clears the message it returns
            f.why = e ? e : "libhipfft.so.0 not found";
            return;
        }
        bool ok = true;
#define SYM(field, name) do { *reinterpret_cast<void**>(&f.field) = dlsym(f.handle, name); if (!f.field) { ok = false; f.why = std::string("missing symbol ") + name; } } while (0)
        SYM(PlanMany, "hipfftPlanMany");
        SYM(SetStream, "hipfftSetStream");
        SYM(ExecZ2Z, "hipfftExecZ2Z");
        SYM(ExecD2Z, "hipfftExecD2Z");
        SYM(Destroy, "hipfftDestroy");
#undef SYM
        if (!ok) { dlclose(f.handle); f.handle = nullptr; }
    });
    return &f;
}

#define SDRHIP_CHECK_FFT(expr)                                                              \
    do {                                                                                    \
        hipfftResult _r = (expr);                                                           \
        if (_r != HIPFFT_SUCCESS) {                                                         \
            set_error("%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
            return SDRHIP_ERR_HIP;                                                          \
        }                                                                                   \
    } while (0)

}  // namespace

struct sdrhip_fft {
    int n = 0, batch = 1;
    bool real_in = false;
    hipfftHandle plan = 0;
    bool have_plan = false;
    hipStream_t stream = nullptr;      // for the host-vector entry point
    DevBuf din, dout;
    PinBuf hin, hout;
    size_t in_bytes() const { return (size_t)batch * n * (real_in ? 8 : 16); }
    size_t out_bytes() const { return (size_t)batch * (real_in ? n / 2 + 1 : n) * 16; }
    ~sdrhip_fft()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        if (have_plan && hipfft()->handle) (void)hipfft()->Destroy(plan);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

int sdrhip_fft_create(sdrhip_fft** out, int n, int real_input, int batch)
{
    SDRHIP_REQUIRE(out != nullptr && n >= 2 && n <= (1 << 27) && batch >= 1 && batch <= (1 << 20), "sdrhip_fft_create");
    *out = nullptr;
    HipFft* h = hipfft();
    if (!h->handle) {
        set_error("sdrhip_fft_create: hipFFT is not available (%s)", h->why.c_str());
        return SDRHIP_ERR_STATE;
    }
    sdrhip_fft* f = new sdrhip_fft();
    f->n = n;
    f->batch = batch;
    f->real_in = real_input != 0;
    int len[1] = {n};
    hipfftResult r = h->PlanMany(&f->plan, 1, len, nullptr, 1, n, nullptr, 1, f->real_in ? n / 2 + 1 : n,
                                 f->real_in ? HIPFFT_D2Z : HIPFFT_Z2Z, batch);
    if (r != HIPFFT_SUCCESS) {
        set_error("hipfftPlanMany(n = %d, batch = %d) failed: hipfftResult %d", n, batch, (int)r);
        delete f;
        return SDRHIP_ERR_HIP;
    }
    f->have_plan = true;
    *out = f;
    return SDRHIP_OK;
}

void sdrhip_fft_destroy(sdrhip_fft* f) { delete f; }
int sdrhip_fft_size(const sdrhip_fft* f) { return f ? f->n : -1; }
int sdrhip_fft_bins(const sdrhip_fft* f) { return f ? (f->real_in ? f->n / 2 + 1 : f->n) : -1; }

// device vectors, asynchronous on `stream`: d_in = batch x n complex doubles (or n doubles), d_out = batch x bins complex doubles
int sdrhip_fft_run_device(sdrhip_fft* f, void* stream, const double* d_in, double* d_out)
{
    SDRHIP_REQUIRE(f != nullptr && d_in != nullptr && d_out != nullptr, "sdrhip_fft_run_device");
    HipFft* h = hipfft();
    SDRHIP_CHECK_FFT(h->SetStream(f->plan, (hipStream_t)stream));
    if (f->real_in) SDRHIP_CHECK_FFT(h->ExecD2Z(f->plan, const_cast<double*>(d_in), reinterpret_cast<hipfftDoubleComplex*>(d_out)));
    else SDRHIP_CHECK_FFT(h->ExecZ2Z(f->plan, reinterpret_cast<hipfftDoubleComplex*>(const_cast<double*>(d_in)),
                                     reinterpret_cast<hipfftDoubleComplex*>(d_out), HIPFFT_FORWARD));
    return SDRHIP_OK;
}

// host vectors, synchronous: what fftw' / fftwReal' return (FFT.hs:44-108), `batch` transforms per call
int sdrhip_fft_run(sdrhip_fft* f, const double* in, double* out)
{
    SDRHIP_REQUIRE(f != nullptr && in != nullptr && out != nullptr, "sdrhip_fft_run");
    if (!f->stream) SDRHIP_CHECK_HIP(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    int rc;
    if ((rc = f->din.ensure(f->in_bytes())) != SDRHIP_OK) return rc;
    if ((rc = f->dout.ensure(f->out_bytes())) != SDRHIP_OK) return rc;
    if ((rc = f->hin.ensure(f->in_bytes())) != SDRHIP_OK) return rc;
    if ((rc = f->hout.ensure(f->out_bytes())) != SDRHIP_OK) return rc;
    memcpy(f->hin.p, in, f->in_bytes());
    SDRHIP_CHECK_HIP(hipMemcpyAsync(f->din.p, f->hin.p, f->in_bytes(), hipMemcpyHostToDevice, f->stream));
    if ((rc = sdrhip_fft_run_device(f, f->stream, (const double*)f->din.p, (double*)f->dout.p)) != SDRHIP_OK) return rc;
    SDRHIP_CHECK_HIP(hipMemcpyAsync(f->hout.p, f->dout.p, f->out_bytes(), hipMemcpyDeviceToHost, f->stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(f->stream));
    memcpy(out, f->hout.p, f->out_bytes());
    return SDRHIP_OK;
}

}  // extern "C"
