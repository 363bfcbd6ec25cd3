// common.hpp -- error plumbing and small RAII helpers shared by the ABI layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/sdr_hip.h"

namespace sdrhip {

void set_error(const char* fmt, ...);
const char* get_error();

#define SDRHIP_CHECK_HIP(expr)                                                                  \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ::sdrhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SDRHIP_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)

#define SDRHIP_REQUIRE(cond, msg)                                        \
    do {                                                                 \
        if (!(cond)) {                                                   \
            ::sdrhip::set_error("%s: requirement `%s` violated", msg, #cond); \
            return SDRHIP_ERR_ARG;                                       \
        }                                                                \
    } while (0)

// Failure inside a drop-in symbol (void returns: they cannot report).  dropin_fail records the message (sdrhip_last_error),
// then calls the process-wide handler installed with sdrhip_set_error_handler and unwinds to the extern "C" boundary, which
// returns to the caller with the outputs unspecified; without a handler it prints and abort()s, as rounds 1-3 did.
struct DropinAbort {};
[[noreturn]] void dropin_fail(int code, const char* fmt, ...);
#define SDRHIP_DIE_HIP(expr)                                                                         \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            ::sdrhip::dropin_fail(SDRHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

inline int round_up(int n, int d) { return ((n + d - 1) / d) * d; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }  // a >= 0, b > 0

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SDRHIP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return SDRHIP_ERR_NOMEM;
        }
        cap = want;
        return SDRHIP_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
};

// Grow-only pinned host buffer.  dev = the address kernels use to read / write it in place (zero-copy over PCIe).
struct PinBuf {
    void* p = nullptr;
    void* dev = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SDRHIP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return SDRHIP_ERR_NOMEM;
        }
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
            (void)hipGetLastError();
            dev = p;                       // unified addressing: the host pointer is the device pointer
        }
        cap = want;
        return SDRHIP_OK;
    }
    void* dev_ptr(const void* host) const { return (char*)dev + ((const char*)host - (const char*)p); }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
    }
    ~PinBuf() { release(); }
};

int upload_floats(float** d, const std::vector<float>& h);

}  // namespace sdrhip
