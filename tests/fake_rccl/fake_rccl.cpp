// TEST INFRASTRUCTURE -- not on the product path.
//
// A stand-in for the ten RCCL entry points libreflexiv_hip.so binds (rfx_comm.hip), so that SEVERAL ranks can run the
// multi-GPU branch of the C ABI on the ONE GPU of a test box: RCCL itself refuses two ranks on one device.  Selected only
// through RFX_RCCL_LIB=<path to this library> (tests/test_gpu_multirank.py); nothing in reflexiv_amd/ refers to it.
//
// Semantics kept: point-to-point messages between a pair of ranks match in the order they were posted (per direction);
// a group's sends never wait for its receives; collectives are in call order.  Semantics NOT kept: nothing here is
// asynchronous -- an operation first drains the stream it was given, then moves the bytes through a file in /dev/shm
// (device -> host -> file -> host -> device), so data is in place when the call (or the group) returns.  Slow, and fine
// for megabytes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct Comm {
    int rank = 0, world = 1;
    char tag[64];
    std::vector<long> sent, got;          // per peer: messages posted / consumed so far
    long coll = 0;
};

struct Op { bool send; void *p; size_t bytes; int peer; Comm *c; hipStream_t s; };
thread_local int depth = 0;
thread_local std::vector<Op> pending;

size_t type_bytes(ncclDataType_t t) {
    switch ((int)t) {
        case 0: case 1: return 1;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        case 6: case 9: return 2;
        default: return 0;
    }
}

std::string path(const Comm *c, const char *kind, long seq, int src, int dst) {
    char b[256];
    snprintf(b, sizeof b, "/dev/shm/%s.%s%ld.%d.%d", c->tag, kind, seq, src, dst);
    return b;
}

bool put_file(const std::string &name, const void *h, size_t bytes) {
    const std::string tmp = name + ".tmp";
    int fd = open(tmp.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
    if (fd < 0) return false;
    const char *p = (const char *)h;
    size_t left = bytes;
    while (left) {
        ssize_t w = write(fd, p, left);
        if (w <= 0) { close(fd); return false; }
        p += w; left -= (size_t)w;
    }
    close(fd);
    return rename(tmp.c_str(), name.c_str()) == 0;           // appears complete or not at all
}

bool get_file(const std::string &name, void *h, size_t bytes) {
    const double limit = getenv("FAKE_RCCL_TIMEOUT_S") ? atof(getenv("FAKE_RCCL_TIMEOUT_S")) : 300.0;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int fd;
    while ((fd = open(name.c_str(), O_RDONLY)) < 0) {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) + 1e-9 * (t.tv_nsec - t0.tv_nsec) > limit) {
            fprintf(stderr, "fake_rccl: gave up waiting for %s\n", name.c_str());
            return false;
        }
        usleep(200);
    }
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && (size_t)st.st_size == bytes;
    if (!ok) fprintf(stderr, "fake_rccl: %s holds %ld bytes, the receiver expects %zu\n", name.c_str(), (long)st.st_size, bytes);
    char *p = (char *)h;
    size_t left = bytes;
    while (ok && left) {
        ssize_t r = read(fd, p, left);
        if (r <= 0) ok = false; else { p += r; left -= (size_t)r; }
    }
    close(fd);
    unlink(name.c_str());
    return ok;
}

ncclResult_t run(std::vector<Op> &ops) {
    for (auto &o : ops) if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> h;
    for (auto &o : ops) {
        if (!o.send) continue;
        h.resize(o.bytes);
        if (hipMemcpy(h.data(), o.p, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!put_file(path(o.c, "p", o.c->sent[o.peer]++, o.c->rank, o.peer), h.data(), o.bytes)) return ncclSystemError;
    }
    for (auto &o : ops) {
        if (o.send) continue;
        h.resize(o.bytes);
        if (!get_file(path(o.c, "p", o.c->got[o.peer]++, o.peer, o.c->rank), h.data(), o.bytes)) return ncclSystemError;
        if (hipMemcpy(o.p, h.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    // (a copy from pageable memory may return once the bytes are staged: kernels on a non-blocking stream must not start
    // before they have landed)
    return hipDeviceSynchronize() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t post(Op o) {
    if (!o.c || o.peer < 0 || o.peer >= o.c->world) return ncclInvalidArgument;
    pending.push_back(o);
    if (depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(pending);
    return run(ops);
}

// every rank's `bytes` from h_mine into h_all[r * bytes]
ncclResult_t exchange_all(Comm *c, const void *h_mine, char *h_all, size_t bytes) {
    const long seq = c->coll++;
    for (int p = 0; p < c->world; p++)
        if (p != c->rank && !put_file(path(c, "c", seq, c->rank, p), h_mine, bytes)) return ncclSystemError;
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) memcpy(h_all + (size_t)p * bytes, h_mine, bytes);
        else if (!get_file(path(c, "c", seq, p, c->rank), h_all + (size_t)p * bytes, bytes)) return ncclSystemError;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "frccl-%d-%ld%09ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->world = nranks;
    memcpy(c->tag, id.internal, sizeof c->tag);
    c->tag[sizeof c->tag - 1] = 0;
    if (strncmp(c->tag, "frccl-", 6) != 0) { delete c; return ncclInvalidArgument; }
    c->sent.assign(nranks, 0); c->got.assign(nranks, 0);
    *comm = (ncclComm_t)c;
    char one = 1;
    std::vector<char> all(nranks);
    return exchange_all(c, &one, all.data(), 1);             // the rendezvous: nobody returns before everybody came
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (Comm *)comm; return ncclSuccess; }

ncclResult_t ncclGroupStart() { depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (depth <= 0) return ncclInvalidUsage;
    if (--depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(pending);
    return run(ops);
}

ncclResult_t ncclSend(const void *p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    if (!type_bytes(t)) return ncclInvalidArgument;
    return post(Op{true, (void *)p, count * type_bytes(t), peer, (Comm *)comm, s});
}

ncclResult_t ncclRecv(void *p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    if (!type_bytes(t)) return ncclInvalidArgument;
    return post(Op{false, p, count * type_bytes(t), peer, (Comm *)comm, s});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_bytes(t);
    if (!c || !bytes || depth > 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> mine(bytes), all(bytes * c->world);
    if (hipMemcpy(mine.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = exchange_all(c, mine.data(), all.data(), bytes);
    if (r != ncclSuccess) return r;
    if (hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return hipDeviceSynchronize() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s) {
    Comm *c = (Comm *)comm;
    if (!c || !count || depth > 0 || t != ncclInt64 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<long long> mine(count), all(count * c->world), out(count);
    if (hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = exchange_all(c, mine.data(), (char *)all.data(), count * 8);
    if (r != ncclSuccess) return r;
    for (size_t i = 0; i < count; i++) {
        long long v = all[i];
        for (int p = 1; p < c->world; p++) {
            const long long w = all[(size_t)p * count + i];
            v = op == ncclSum ? (long long)((unsigned long long)v + (unsigned long long)w) : (w > v ? w : v);
        }
        out[i] = v;
    }
    if (hipMemcpy(recv, out.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return hipDeviceSynchronize() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP error";
        case ncclSystemError: return "fake_rccl: message file missing, short or late";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        case ncclInvalidUsage: return "fake_rccl: invalid usage";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"
