// scratch_pool.hpp -- per-call scratch contexts of the host-pointer entry points (the drop-in symbols, abi_dropin.cpp,
// and the record seam, abi_records.cpp).  Every call leases a context -- one HIP stream, pinned staging, device buffers,
// the taps it uploaded last -- from a per-device pool, so concurrent pipeline threads run side by side instead of queueing
// on one mutex and one stream.  Contexts live as long as the process (no destructor-order games at exit).
#pragma once
#include <vector>

#include "common.hpp"

namespace sdrhip {

struct ScratchCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    PinBuf hin, hout;
    DevBuf in, out, taps, taps2, work;
    std::vector<unsigned char> taps_now, taps2_now;   // the bytes taps / taps2 hold
};

// nullptr (with the error text set) when the current device cannot be queried or no stream can be created
ScratchCtx* scratch_acquire();
void scratch_release(ScratchCtx* c);

struct ScratchLease {
    ScratchCtx* c;
    ScratchLease() : c(scratch_acquire()) {}
    ~ScratchLease()
    {
        if (c) scratch_release(c);
    }
    ScratchLease(const ScratchLease&) = delete;
    ScratchLease& operator=(const ScratchLease&) = delete;
};

}  // namespace sdrhip
