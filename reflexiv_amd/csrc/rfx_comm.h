// rfx_comm.h -- what rfx_comm.hip (the count stage's shuffle) and rfx_shard.hip (the extend stage's range shuffle) share:
// the RCCL binding, the communicator, and the all-to-all(v) of 8-byte words.
#pragma once
#include <rccl/rccl.h>
#include <algorithm>
#include "rfx_internal.h"

struct NcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Gather_unused)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

NcclApi &nccl();

constexpr int TABW = 512;                         // counts a rank contributes to the count matrix: (generation, owner) bins x pieces

struct rfx_comm {
    rfx_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t xs = nullptr;                     // the exchange runs on its own stream
    std::vector<hipEvent_t> ev;                   // one per generation: "generation g has landed"
    hipEvent_t ev_ready = nullptr;                // "the send buffer is complete" (context stream -> exchange stream)
    // grow-only device buffers: what this rank sends, what it receives (all generations back to back), small tables
    void *send = nullptr, *recv = nullptr;
    size_t send_bytes = 0, recv_bytes = 0;
    int64_t *d_tab = nullptr;                     // [TABW] this rank's counts, [TABW * world] everybody's, then 16 scalars
    int64_t *h_tab = nullptr;                     // pinned mirror
    size_t tab_n = 0;
    double units_per_read = 0;                    // capacity planning across calls (only ever grows)
    int64_t bytes_bucketed = 0;                   // of the last call
    size_t limit_bytes = (size_t)1 << 29;         // per peer and call (RCCL 2.26 corrupts messages above 1 GiB)
    bool self_via_rccl = false;                   // tests: send the rank's own bucket through ncclSend / ncclRecv too
    int virtual_world = 1;                        // one-rank rehearsal of an N-rank node (see rfx_dev_sharded_count)
    int64_t *d_sh = nullptr, *h_sh = nullptr;     // the extend stage's range shuffle (rfx_shard.hip): rows, samples, splitters, carry
};

#define RFX_NCCL(call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            char buf_[512];                                                                                \
            snprintf(buf_, sizeof buf_, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,                      \
                     nccl().GetErrorString ? nccl().GetErrorString(r_) : "RCCL error");                    \
            if (ctx) ctx->last_error = buf_;                                                               \
            return RFX_E_HIP;                                                                              \
        }                                                                                                  \
    } while (0)

// An open ncclGroupStart is closed on every way out (an RFX_NCCL return from inside a group used to leave the communicator
// in group state).
struct GroupGuard {
    bool open = false;
    ~GroupGuard() { if (open && nccl().GroupEnd) (void)nccl().GroupEnd(); }
};

// The HIP "last error" of this thread after RCCL has run its own runtime calls on it (stream / event queries, pointer
// attribute probes): hipGetLastError is thread-local, every HIP call of THIS library is checked where it is made and RCCL
// reports its own failures through ncclResult_t, so what is found here was raised and handled inside RCCL.  It must not
// reach the kernels' launch checks (hipGetLastError after a launch); it is read, and anything but the two codes RCCL's
// polling leaves behind is kept for rfx_last_error() ("[after RCCL: ...]") instead of being thrown away unread.
inline void note_foreign_hip_error(rfx_ctx *ctx, const char *where) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess || e == hipErrorNotReady || e == hipErrorPeerAccessAlreadyEnabled) return;
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
    ctx->foreign_hip_error = buf;
}

void comm_options(rfx_comm *c);
int comm_grow(rfx_ctx *ctx, void **p, size_t *have, size_t want, hipStream_t s1, hipStream_t s2);
// One all-to-all(v) of 8-byte words queued on `stream` (rfx_comm.hip)
int alltoallv_words(rfx_comm *c, const uint64_t *d_send, const int64_t *send_off, const int64_t *send_cnt, uint64_t *d_recv,
                    const int64_t *recv_off, const int64_t *recv_cnt, int64_t rounds, int S, hipStream_t stream);
