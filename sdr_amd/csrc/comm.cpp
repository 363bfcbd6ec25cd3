// comm.cpp -- the multi-GPU halo exchange of the sharded FM chain behind the C ABI.
//
// The stream shards by sample block; the only remote data a rank needs is a RIGHT halo -- the first ~4k samples (8 KB of
// u8 IQ) of its right neighbour's shard, the ntaps-1 overlaps of the four stages composed (sdrhip_fm_chain_plan).  One
// neighbour message per step: latency-bound, so it is a point-to-point pair, never a collective:
//     ncclGroupStart(); ncclSend(head -> left); ncclRecv(halo <- right); ncclGroupEnd();      on the caller's stream
// or, when one process drives every device, hipMemcpyPeerAsync of the same bytes over xGMI.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): libsdr_hip.so keeps loading on hosts without it, and inside a process
// that already holds a copy (PyTorch ships one under the same soname) the loader hands back that copy, so a process never
// runs two RCCLs.  The reference has no multi-device code; there is nothing this file could have been translated from.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// RCCL's development headers are absent: the library is bound at run time anyway, and these are the only declarations of
// rccl.h this file uses (NCCL's ABI for them has not changed since 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
#endif
#include <string.h>

#include <mutex>
#include <vector>

#include "descriptors.hpp"

using namespace sdrhip;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;     // why loading failed
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();        // one call: it clears the message it returns
            r.why = e ? e : "librccl.so.1 not found";
            return;
        }
        bool ok = true;
#define SYM(field, name)                                                             \
    do {                                                                             \
        *reinterpret_cast<void**>(&r.field) = dlsym(r.handle, name);                 \
        if (!r.field) { ok = false; r.why = std::string("missing symbol ") + name; } \
    } while (0)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(Send, "ncclSend");
        SYM(Recv, "ncclRecv");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return &r;
}

int need_rccl(const char* who)
{
    Rccl* r = rccl();
    if (r->handle) return SDRHIP_OK;
    set_error("%s: RCCL is not available (%s)", who, r->why.c_str());
    return SDRHIP_ERR_STATE;
}

#define SDRHIP_CHECK_NCCL(expr)                                                                                          \
    do {                                                                                                                 \
        ncclResult_t _r = (expr);                                                                                        \
        if (_r != ncclSuccess) {                                                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return SDRHIP_ERR_HIP;                                                                                       \
        }                                                                                                                \
    } while (0)

}  // namespace

struct sdrhip_comm {
    int nranks = 1, rank = 0, device = 0;
    int transport = SDRHIP_TRANSPORT_RCCL;
    ncclComm_t nccl = nullptr;
    hipEvent_t ev_head = nullptr;       // peer copy: "this rank's stream has produced its head"
    hipEvent_t ev_pulled = nullptr;     // peer copy: "this rank has finished reading its RIGHT neighbour's head"
    bool local = false;                 // created by init_local (all ranks in this process)
};

extern "C" {

int sdrhip_comm_get_unique_id(void* id)
{
    SDRHIP_REQUIRE(id != nullptr, "sdrhip_comm_get_unique_id");
    static_assert(sizeof(ncclUniqueId) == SDRHIP_COMM_ID_BYTES, "the id travels as an opaque 128-byte blob");
    int rc = need_rccl("sdrhip_comm_get_unique_id");
    if (rc != SDRHIP_OK) return rc;
    ncclUniqueId u;
    SDRHIP_CHECK_NCCL(rccl()->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return SDRHIP_OK;
}

int sdrhip_comm_init_rank(sdrhip_comm** c, int nranks, int rank, const void* id)
{
    SDRHIP_REQUIRE(c != nullptr && id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "sdrhip_comm_init_rank");
    *c = nullptr;
    int rc = need_rccl("sdrhip_comm_init_rank");
    if (rc != SDRHIP_OK) return rc;
    int dev = 0;
    SDRHIP_CHECK_HIP(hipGetDevice(&dev));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    sdrhip_comm* cm = new sdrhip_comm();
    cm->nranks = nranks;
    cm->rank = rank;
    cm->device = dev;
    ncclResult_t r = rccl()->CommInitRank(&cm->nccl, nranks, u, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, dev, rccl()->GetErrorString(r));
        delete cm;
        return SDRHIP_ERR_HIP;
    }
    *c = cm;
    return SDRHIP_OK;
}

int sdrhip_comm_init_local(sdrhip_comm** comms, int ndev, const int* devices, int transport)
{
    SDRHIP_REQUIRE(comms != nullptr && ndev >= 1 && ndev <= 64, "sdrhip_comm_init_local");
    SDRHIP_REQUIRE(transport == SDRHIP_TRANSPORT_RCCL || transport == SDRHIP_TRANSPORT_PEER_COPY, "sdrhip_comm_init_local");
    for (int i = 0; i < ndev; i++) comms[i] = nullptr;
    int count = 0;
    SDRHIP_CHECK_HIP(hipGetDeviceCount(&count));
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) {
        devs[i] = devices ? devices[i] : i;
        SDRHIP_REQUIRE(devs[i] >= 0 && devs[i] < count, "sdrhip_comm_init_local: device index");
    }
    std::vector<ncclComm_t> nc(ndev, nullptr);
    if (transport == SDRHIP_TRANSPORT_RCCL) {
        int rc = need_rccl("sdrhip_comm_init_local");
        if (rc != SDRHIP_OK) return rc;
        SDRHIP_CHECK_NCCL(rccl()->CommInitAll(nc.data(), ndev, devs.data()));
    }
    int prev = 0;
    {
        const hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) {
            set_error("sdrhip_comm_init_local: hipGetDevice: %s", hipGetErrorString(e));
            for (ncclComm_t c : nc)                 // the communicators CommInitAll just made must not leak
                if (c) (void)rccl()->CommDestroy(c);
            return SDRHIP_ERR_HIP;
        }
    }
    for (int i = 0; i < ndev; i++) {
        sdrhip_comm* cm = new sdrhip_comm();
        cm->nranks = ndev;
        cm->rank = i;
        cm->device = devs[i];
        cm->transport = transport;
        cm->nccl = nc[i];
        cm->local = true;
        comms[i] = cm;
        if (transport == SDRHIP_TRANSPORT_PEER_COPY) {
            hipError_t e = hipSetDevice(devs[i]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_head, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_pulled, hipEventDisableTiming);
            if (e != hipSuccess) {
                set_error("sdrhip_comm_init_local: device %d: %s", devs[i], hipGetErrorString(e));
                for (int k = 0; k <= i; k++) {
                    sdrhip_comm_destroy(comms[k]);
                    comms[k] = nullptr;
                }
                (void)hipSetDevice(prev);
                return SDRHIP_ERR_HIP;
            }
            // direct xGMI access to the right neighbour's memory where the topology offers it (the copy works without)
            const int right = devs[(i + 1) % ndev];
            int can = 0;
            if (right != devs[i] && hipDeviceCanAccessPeer(&can, devs[i], right) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(right, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    (void)hipSetDevice(prev);
    return SDRHIP_OK;
}

void sdrhip_comm_destroy(sdrhip_comm* c)
{
    if (!c) return;
    if (c->nccl && rccl()->handle) (void)rccl()->CommDestroy(c->nccl);
    if (c->ev_head) (void)hipEventDestroy(c->ev_head);
    if (c->ev_pulled) (void)hipEventDestroy(c->ev_pulled);
    delete c;
}

int sdrhip_comm_rank(const sdrhip_comm* c) { return c ? c->rank : -1; }
int sdrhip_comm_size(const sdrhip_comm* c) { return c ? c->nranks : -1; }
const char* sdrhip_comm_transport(const sdrhip_comm* c)
{
    if (!c) return "none";
    return c->transport == SDRHIP_TRANSPORT_RCCL ? "rccl" : "peer-copy";
}

int sdrhip_halo_exchange(sdrhip_comm* c, void* stream, const void* d_send, void* d_recv, size_t bytes)
{
    SDRHIP_REQUIRE(c != nullptr && d_send != nullptr && d_recv != nullptr, "sdrhip_halo_exchange");
    if (bytes == 0) return SDRHIP_OK;
    if (c->transport != SDRHIP_TRANSPORT_RCCL) {
        set_error("sdrhip_halo_exchange: a peer-copy communicator moves data with sdrhip_halo_exchange_all (one rank cannot "
                  "reach its neighbour's memory on its own)");
        return SDRHIP_ERR_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int left = (c->rank + c->nranks - 1) % c->nranks, right = (c->rank + 1) % c->nranks;
    SDRHIP_CHECK_NCCL(rccl()->GroupStart());
    ncclResult_t r1 = rccl()->Send(d_send, bytes, ncclUint8, left, c->nccl, s);
    ncclResult_t r2 = rccl()->Recv(d_recv, bytes, ncclUint8, right, c->nccl, s);
    ncclResult_t r3 = rccl()->GroupEnd();
    SDRHIP_CHECK_NCCL(r1);
    SDRHIP_CHECK_NCCL(r2);
    SDRHIP_CHECK_NCCL(r3);
    return SDRHIP_OK;
}

int sdrhip_halo_exchange_all(sdrhip_comm* const* comms, int ndev, void* const* streams, const void* const* d_send,
                             void* const* d_recv, size_t bytes)
{
    SDRHIP_REQUIRE(comms != nullptr && streams != nullptr && d_send != nullptr && d_recv != nullptr && ndev >= 1,
                   "sdrhip_halo_exchange_all");
    for (int i = 0; i < ndev; i++)
        SDRHIP_REQUIRE(comms[i] != nullptr && comms[i]->local && comms[i]->nranks == ndev && comms[i]->rank == i &&
                           comms[i]->transport == comms[0]->transport,
                       "sdrhip_halo_exchange_all: the communicators of one sdrhip_comm_init_local call, in rank order");
    if (bytes == 0) return SDRHIP_OK;
    if (comms[0]->transport == SDRHIP_TRANSPORT_RCCL) {
        SDRHIP_CHECK_NCCL(rccl()->GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int i = 0; i < ndev; i++) {
            const int left = (i + ndev - 1) % ndev, right = (i + 1) % ndev;
            ncclResult_t r1 = rccl()->Send(d_send[i], bytes, ncclUint8, left, comms[i]->nccl, (hipStream_t)streams[i]);
            ncclResult_t r2 = rccl()->Recv(d_recv[i], bytes, ncclUint8, right, comms[i]->nccl, (hipStream_t)streams[i]);
            if (r1 != ncclSuccess) bad = r1;
            if (r2 != ncclSuccess) bad = r2;
        }
        ncclResult_t r3 = rccl()->GroupEnd();
        SDRHIP_CHECK_NCCL(bad);
        SDRHIP_CHECK_NCCL(r3);
        return SDRHIP_OK;
    }
    // peer copy: every rank first marks "my head is final" on its own stream, then each rank's stream waits for its RIGHT
    // neighbour's mark and pulls the head across; the neighbour's stream in turn waits for that pull before anything queued
    // after this call may overwrite the head (with RCCL the send sits on the owner's own stream and orders itself)
    int prev = 0;
    SDRHIP_CHECK_HIP(hipGetDevice(&prev));
    struct Restore {                       // the early returns of the checks below must not leave the caller on another device
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{prev};
    for (int i = 0; i < ndev; i++) {
        SDRHIP_CHECK_HIP(hipSetDevice(comms[i]->device));
        SDRHIP_CHECK_HIP(hipEventRecord(comms[i]->ev_head, (hipStream_t)streams[i]));
    }
    for (int i = 0; i < ndev; i++) {
        const int right = (i + 1) % ndev;
        SDRHIP_CHECK_HIP(hipSetDevice(comms[i]->device));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)streams[i], comms[right]->ev_head, 0));
        SDRHIP_CHECK_HIP(hipMemcpyPeerAsync(d_recv[i], comms[i]->device, d_send[right], comms[right]->device, bytes,
                                            (hipStream_t)streams[i]));
        SDRHIP_CHECK_HIP(hipEventRecord(comms[i]->ev_pulled, (hipStream_t)streams[i]));
    }
    for (int i = 0; i < ndev; i++) {
        const int right = (i + 1) % ndev;
        if (right == i) continue;
        SDRHIP_CHECK_HIP(hipSetDevice(comms[right]->device));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent((hipStream_t)streams[right], comms[i]->ev_pulled, 0));
    }
    return SDRHIP_OK;
}

}  // extern "C"
