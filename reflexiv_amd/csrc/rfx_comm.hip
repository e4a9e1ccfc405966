// rfx_comm.hip -- the Spark shuffle of the count stage on several GPUs, behind the C ABI.
//
// `reduceByKey` (P/ReflexivMain.java:155, with its map-side combine) / `groupBy("value").count()`
// (P/ReflexivDSMain.java:207-209) repartition the k-mer instances by key hash over the executors.  Here the k-mer space
// is radix-sharded over the GPUs of one node by the owner of each k-mer's minimiser and the shuffle is an
// all-to-all(v) over RCCL (xGMI): one process (or thread) per GPU, one rfx_ctx + one rfx_comm each.
//
//   rfx_dev_sharded_count:  reads in HBM -> super-k-mer records bucketed ONCE by (generation, owner) -- bin g*world + o
//   of an owner function over G*world bins -- one ncclAllGather agrees every count, the G exchanges are queued back to
//   back on the communicator's own stream (ncclGroupStart { ncclSend / ncclRecv per peer } ncclGroupEnd, never more
//   than 512 MiB per peer and call; the rank's own bucket moves by a device copy), and generation g is counted on the
//   context's stream while g+1.. are still travelling.  A k-mer lives in exactly one generation, so the G counts are
//   independent; their ascending outputs are merged by one sort of the survivors.  Scalars by ncclAllReduce.
//
// RCCL is bound at run time (dlopen): a host that already carries an RCCL (a PyTorch process does) keeps using that
// one, a plain C / JNI host gets /opt/rocm/lib/librccl.so.1 -- and the library stays loadable where RCCL is absent.
#include <dlfcn.h>
#include <mutex>
#include "rfx_comm.h"
#include "rfx_device.h"

NcclApi &nccl() {
    static NcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // RFX_RCCL_LIB: an explicit library instead (tests: a stand-in that lets several ranks share one GPU)
        if (const char *e = getenv("RFX_RCCL_LIB")) {
            api.h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
            if (!api.h) { api.error = std::string("RFX_RCCL_LIB: ") + (dlerror() ? dlerror() : "?"); return; }
        }
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the host process's own RCCL first
        const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *p : paths)
            if (!api.h) api.h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!api.h) { api.error = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *s) { void *p = dlsym(api.h, s); if (!p) api.error = std::string("RCCL lacks ") + s; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return api;
}

// tuning / test knobs, read at every collective call (so that one communicator serves every case of a test run: RCCL does
// not take kindly to many communicators made and destroyed in one process)
void comm_options(rfx_comm *c) {
    const char *e = getenv("RFX_COMM_LIMIT_BYTES");
    c->limit_bytes = e ? std::max<size_t>(1024, (size_t)atoll(e)) : (size_t)1 << 29;
    e = getenv("RFX_COMM_SELF_VIA_RCCL");
    c->self_via_rccl = e && atoi(e) != 0;
    e = getenv("RFX_COMM_VIRTUAL_WORLD");
    c->virtual_world = e ? std::max(1, std::min(64, atoi(e))) : 1;
}

int comm_grow(rfx_ctx *ctx, void **p, size_t *have, size_t want, hipStream_t s1, hipStream_t s2) {
    if (*p && *have >= want) return RFX_OK;
    if (*p) { RFX_HIP(hipStreamSynchronize(s1)); RFX_HIP(hipStreamSynchronize(s2)); RFX_HIP(hipFree(*p)); *p = nullptr; *have = 0; }
    const size_t bytes = want + (want >> 4) + (1 << 20);
    RFX_HIP(hipMalloc(p, bytes));
    *have = bytes;
    return RFX_OK;
}

extern "C" {

int rfx_comm_unique_id(uint8_t *id128) try {
    if (!id128) return RFX_E_ARG;
    NcclApi &n = nccl();
    if (!n.error.empty() || !n.GetUniqueId) return RFX_E_NOGPU;
    ncclUniqueId id;
    if (n.GetUniqueId(&id) != ncclSuccess) return RFX_E_HIP;
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return RFX_OK;
} RFX_API_CATCH(nullptr)

int rfx_comm_init(rfx_ctx *ctx, const uint8_t *id128, int rank, int world, rfx_comm **out) try {
    if (!ctx || !id128 || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return RFX_E_ARG;
    NcclApi &n = nccl();
    if (!n.error.empty()) { ctx->last_error = n.error; return RFX_E_NOGPU; }
    RFX_HIP(hipSetDevice(ctx->device));
    rfx_comm *c = new rfx_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    comm_options(c);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = n.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank -> ") + (n.GetErrorString ? n.GetErrorString(r) : "error");
        delete c;
        return RFX_E_HIP;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    c->tab_n = (size_t)TABW * (world + 1) + 16;
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tab, c->tab_n * 8);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_tab, c->tab_n * 8, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->last_error = std::string("rfx_comm_init: ") + hipGetErrorString(e); rfx_comm_destroy(c); return RFX_E_HIP; }
    *out = c;
    return RFX_OK;
} RFX_API_CATCH(ctx)

void rfx_comm_destroy(rfx_comm *c) try {
    if (!c) return;
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->comm && nccl().CommDestroy) (void)nccl().CommDestroy(c->comm);
    for (auto e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->d_tab) (void)hipFree(c->d_tab);
    if (c->h_tab) (void)hipHostFree(c->h_tab);
    if (c->d_sh) (void)hipFree(c->d_sh);
    if (c->h_sh) (void)hipHostFree(c->h_sh);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    delete c;
} RFX_API_CATCH_VOID((c ? c->ctx : nullptr))

int rfx_comm_rank(const rfx_comm *c) { return c ? c->rank : -1; }
int rfx_comm_world(const rfx_comm *c) { return c ? c->world : 0; }
int64_t rfx_comm_last_bytes_bucketed(const rfx_comm *c) { return c ? c->bytes_bucketed : 0; }

// sum (op 0) or max (op 1) of n <= 8 host int64 over the ranks, in place (the stop rule's count(), totals, barriers)
int rfx_comm_all_reduce_i64(rfx_comm *c, int64_t *h_vals, int n, int op) try {
    if (!c || !h_vals || n < 1 || n > 8 || (op != 0 && op != 1)) return RFX_E_ARG;
    rfx_ctx *ctx = c->ctx;
    RFX_HIP(hipSetDevice(ctx->device));
    int64_t *d = c->d_tab + c->tab_n - 16, *h = c->h_tab + c->tab_n - 16;
    for (int i = 0; i < n; i++) h[i] = h_vals[i];
    RFX_HIP(hipMemcpyAsync(d, h, (size_t)n * 8, hipMemcpyHostToDevice, c->xs));
    RFX_NCCL(nccl().AllReduce(d, d + 8, (size_t)n, ncclInt64, op == 0 ? ncclSum : ncclMax, c->comm, c->xs));
    RFX_HIP(hipMemcpyAsync(h + 8, d + 8, (size_t)n * 8, hipMemcpyDeviceToHost, c->xs));
    RFX_HIP(hipStreamSynchronize(c->xs));
    note_foreign_hip_error(ctx, "rfx_comm_all_reduce_i64");
    for (int i = 0; i < n; i++) h_vals[i] = h[8 + i];
    return RFX_OK;
} RFX_API_CATCH((c ? c->ctx : nullptr))

}  // extern "C"

// One all-to-all(v) of 8-byte words, queued on the exchange stream: this rank sends send_cnt[p] words from
// d_send + send_off[p] to peer p and receives recv_cnt[p] words from peer p at d_recv + recv_off[p].
// What goes to (comes from) a peer may be S PIECES (the sweep's bins of an owner bucket, rfx::bucket_records_by_owner_sweep):
// piece q of peer p is entry p * S + q of the four arrays, and a pair of ranks meets its pieces in the same order on both sides.
int alltoallv_words(rfx_comm *c, const uint64_t *d_send, const int64_t *send_off, const int64_t *send_cnt,
                    uint64_t *d_recv, const int64_t *recv_off, const int64_t *recv_cnt, int64_t rounds, int S, hipStream_t stream) {
    rfx_ctx *ctx = c->ctx;
    NcclApi &n = nccl();
    const int64_t limit = (int64_t)(c->limit_bytes / 8);
    const int me = c->rank;
    if (!c->self_via_rccl)
        for (int q = 0; q < S; q++)
            if (send_cnt[me * S + q] > 0)
                RFX_HIP(hipMemcpyAsync(d_recv + recv_off[me * S + q], d_send + send_off[me * S + q], (size_t)send_cnt[me * S + q] * 8,
                                       hipMemcpyDeviceToDevice, stream));
    for (int64_t j = 0; j < rounds; j++) {
        bool any = false;
        for (int i = 0; i < c->world * S && !any; i++)
            if ((i / S != me || c->self_via_rccl) && (send_cnt[i] > j * limit || recv_cnt[i] > j * limit)) any = true;
        if (!any) continue;               // (every rank skips the same rounds only when nobody has data left in them: `rounds` is global)
        GroupGuard gg;
        RFX_NCCL(n.GroupStart());
        gg.open = true;
        for (int p = 0; p < c->world; p++) {
            if (p == me && !c->self_via_rccl) continue;
            for (int q = 0; q < S; q++) {
                const int i = p * S + q;
                const int64_t s = std::max<int64_t>(0, std::min(limit, send_cnt[i] - j * limit));
                const int64_t r = std::max<int64_t>(0, std::min(limit, recv_cnt[i] - j * limit));
                if (s > 0) RFX_NCCL(n.Send(d_send + send_off[i] + j * limit, (size_t)s, ncclUint64, p, c->comm, stream));
                if (r > 0) RFX_NCCL(n.Recv(d_recv + recv_off[i] + j * limit, (size_t)r, ncclUint64, p, c->comm, stream));
            }
        }
        gg.open = false;
        RFX_NCCL(n.GroupEnd());
    }
    return RFX_OK;
}

// ---- k outside the record path (k = 3..20 one-word k-mers; k = 65..125 with k % 32 != 0: three- and four-word k-mers in the
// counter's layout): the units of the exchange are the k-mer instances themselves.  Not a BASELINE configuration (both of
// its k's take the record path); here so that the sharded counter takes every k the one-GPU counters take
// (P/ReflexivDataFrameCounter64.java:401-650, klist of U/DefaultParam.java:87: 67, 81, 95).
namespace {
__global__ void k_owner_w(const uint64_t *__restrict__ aos, int64_t n, int W, int world, uint64_t *__restrict__ owner, uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = 0;
    for (int w = 0; w < W; w++) h = rfxd::mix64(h ^ aos[i * W + w]);
    owner[i] = (uint64_t)__umul64hi(h, (uint64_t)world);
    idx[i] = (uint32_t)i;
}
__global__ void k_gather_w(const uint64_t *__restrict__ aos, const uint32_t *__restrict__ idx, int64_t n, int W, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *s = aos + (int64_t)idx[i] * W;
    for (int w = 0; w < W; w++) out[i * W + w] = s[w];
}
__global__ void k_owner_counts(const uint64_t *__restrict__ sorted_owner, int64_t n, int world, int64_t *__restrict__ off) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // off[t] = first index with owner >= t
    if (t > world) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_owner[mid] < (uint64_t)t) lo = mid + 1; else hi = mid; }
    off[t] = lo;
}
}  // namespace

static int sharded_count_kmers(rfx_ctx *ctx, rfx_comm *c, const uint64_t *d_words, int64_t n_reads, int wpr, int read_len, int k, int fc, int ec,
                               int min_cov, int max_cov, int twin, uint64_t *d_out_keys, void *d_out_counts, int64_t cap, int64_t *out_n,
                               int64_t *out_totals) {
    const int world = c->world, me = c->rank;
    const bool wide = k > 32;
    const int W = wide ? k / 32 + 1 : 1;
    NcclApi &n = nccl();
    const int64_t nk = wide ? rfx::kmers_per_read_w(read_len, k, fc, ec) : rfx::kmers_per_read(read_len, k, fc, ec);
    const int64_t N = nk * n_reads;
    int64_t hoff[65];
    for (int i = 0; i <= 64; i++) hoff[i] = 0;
    DevBuf units;                                             // this rank's k-mers grouped by owner (AoS, W words each)
    // this rank's own part (its failure travels in the matrix)
    auto bucket = [&]() -> int {
        if (N >= ((int64_t)1 << 32)) { ctx->last_error = "rfx_dev_sharded_count (k-mer units): at most 2^32 - 1 instances per rank and call"; return RFX_E_LIMIT; }
        RFX_HIP(units.alloc((size_t)std::max<int64_t>(1, N) * W * 8, ctx->stream));
        if (N == 0) return RFX_OK;
        if (!wide) {
            rfx::ReadStore rs{d_words, n_reads, wpr, read_len, k, fc, ec};
            DevBuf doff;
            RFX_HIP(doff.alloc((size_t)(world + 1) * 8, ctx->stream));
            RFX_TRY(rfx::bucket_by_owner(ctx, &rs, world, units.as<uint64_t>(), N, doff.as<int64_t>(), hoff));
            return RFX_OK;
        }
        DevBuf aos, owner, idx, tk, tv, doff;
        RFX_HIP(aos.alloc((size_t)N * W * 8, ctx->stream));
        RFX_HIP(owner.alloc((size_t)N * 8, ctx->stream));
        RFX_HIP(idx.alloc((size_t)N * 4, ctx->stream));
        RFX_HIP(tk.alloc((size_t)N * 8, ctx->stream));
        RFX_HIP(tv.alloc((size_t)N * 4, ctx->stream));
        RFX_HIP(doff.alloc((size_t)(world + 1) * 8, ctx->stream));
        RFX_TRY(rfx::extract_w(ctx, d_words, wpr, nullptr, nk, n_reads, k, fc, aos.as<uint64_t>(), N, 1));
        const unsigned g = (unsigned)ceil_div(N, 256);
        hipLaunchKernelGGL(k_owner_w, dim3(g), dim3(256), 0, ctx->stream, (const uint64_t *)aos.as<uint64_t>(), N, W, world, owner.as<uint64_t>(), idx.as<uint32_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(rfx::sort_pairs(ctx, owner.as<uint64_t>(), idx.as<uint32_t>(), N, 8, tk.as<uint64_t>(), tv.as<uint32_t>()));
        hipLaunchKernelGGL(k_gather_w, dim3(g), dim3(256), 0, ctx->stream, (const uint64_t *)aos.as<uint64_t>(), (const uint32_t *)idx.as<uint32_t>(), N, W, units.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_owner_counts, dim3(1), dim3(128), 0, ctx->stream, (const uint64_t *)owner.as<uint64_t>(), N, world, doff.as<int64_t>());
        RFX_HIP(hipGetLastError());
        RFX_HIP(hipMemcpyAsync(hoff, doff.p, (size_t)(world + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        return RFX_OK;
    };
    int st_local = bucket();
    if (st_local != RFX_OK) { (void)hipStreamSynchronize(ctx->stream); for (int i = 0; i <= 64; i++) hoff[i] = 0; }
    c->bytes_bucketed = (st_local == RFX_OK ? N : 0) * W * 8;
    // the count matrix: row = [world] k-mers for each owner, then this rank's status
    const int row = world + 1;
    int64_t *h_mine = c->h_tab, *h_all = c->h_tab + TABW;
    for (int p = 0; p < world; p++) h_mine[p] = hoff[p + 1] - hoff[p];
    h_mine[world] = st_local;
    RFX_HIP(hipMemcpyAsync(c->d_tab, h_mine, (size_t)row * 8, hipMemcpyHostToDevice, c->xs));
    RFX_NCCL(n.AllGather(c->d_tab, c->d_tab + TABW, (size_t)row, ncclInt64, c->comm, c->xs));
    RFX_HIP(hipMemcpyAsync(h_all, c->d_tab + TABW, (size_t)row * world * 8, hipMemcpyDeviceToHost, c->xs));
    RFX_HIP(hipStreamSynchronize(c->xs));
    note_foreign_hip_error(ctx, "rfx_dev_sharded_count (k-mer units: count matrix)");
    for (int r = 0; r < world; r++)
        if (h_all[(size_t)r * row + world] != RFX_OK) {
            if (st_local != RFX_OK) return st_local;
            ctx->last_error = "rfx_dev_sharded_count: a peer failed before the exchange; see its rfx_last_error()";
            return RFX_E_STATE;
        }
    std::vector<int64_t> soff(world), scnt(world), roff(world), rcnt(world);
    int64_t n_in = 0, mx = 0;
    for (int s_ = 0; s_ < world; s_++) {
        const int64_t u = h_all[(size_t)s_ * row + me];
        roff[s_] = n_in * W; rcnt[s_] = u * W; n_in += u;
        soff[s_] = hoff[s_] * W; scnt[s_] = (hoff[s_ + 1] - hoff[s_]) * W;
    }
    for (size_t i = 0; i < (size_t)world * row; i++) if ((int)(i % row) < world) mx = std::max(mx, h_all[i] * W);
    const int64_t lim = (int64_t)(c->limit_bytes / 8);
    const int64_t rounds = std::max<int64_t>(1, (mx + lim - 1) / lim);
    // what arrives, and whether every rank could make room for it (the all-reduce below is every rank's, always)
    DevBuf recv, soa;
    int st_recv = RFX_OK;
    if (recv.alloc((size_t)std::max<int64_t>(1, n_in) * W * 8, ctx->stream) != hipSuccess) { st_recv = RFX_E_HIP; ctx->last_error = "rfx_dev_sharded_count: no room for what arrives"; }
    if (n_in >= ((int64_t)1 << 32)) { st_recv = RFX_E_LIMIT; ctx->last_error = "rfx_dev_sharded_count (k-mer units): a shard of 2^32 instances or more"; }
    {
        int64_t bad[1] = {st_recv != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, bad, 1, 1));
        if (st_recv != RFX_OK) return st_recv;
        if (bad[0]) { ctx->last_error = "rfx_dev_sharded_count: a peer could not take its shard"; return RFX_E_STATE; }
    }
    RFX_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    RFX_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
    RFX_TRY(alltoallv_words(c, units.as<uint64_t>(), soff.data(), scnt.data(), recv.as<uint64_t>(), roff.data(), rcnt.data(), rounds, 1, c->xs));
    RFX_HIP(hipStreamSynchronize(c->xs));
    note_foreign_hip_error(ctx, "rfx_dev_sharded_count (k-mer units: exchange)");
    units.release();
    // this rank's shard: count + filter (a failure from here on is carried into the closing all-reduce)
    int64_t m = 0, distinct = 0;
    int st_cnt = RFX_OK;
    ctx->timing.clear();
    if (!wide) {
        st_cnt = rfx::count_filter(ctx, nullptr, recv.as<uint64_t>(), n_in, min_cov, max_cov, twin, nullptr, 0, d_out_keys, (int32_t *)d_out_counts, cap, &m, &distinct);
    } else {
        auto cnt = [&]() -> int {
            RFX_HIP(soa.alloc((size_t)std::max<int64_t>(1, n_in) * W * 8, ctx->stream));
            if (n_in > 0) RFX_TRY(rfx::aos_to_soa(ctx, recv.as<uint64_t>(), n_in, W, soa.as<uint64_t>()));
            return rfx::count_filter_w(ctx, soa.as<uint64_t>(), n_in, k, min_cov, max_cov, d_out_keys, (int64_t *)d_out_counts, cap, &m, &distinct);
        };
        st_cnt = cnt();
    }
    { const int ss = sync_checked(ctx); if (st_cnt == RFX_OK || st_cnt == RFX_E_CAP) { if (ss != RFX_OK) st_cnt = ss; } }
    ScopedTimer::collect(ctx);
    if (out_n) *out_n = m;
    int64_t tot[5] = {N, distinct, m, st_cnt == RFX_E_CAP ? 1 : 0, (st_cnt != RFX_OK && st_cnt != RFX_E_CAP) ? 1 : 0};
    RFX_TRY(rfx_comm_all_reduce_i64(c, tot, 5, 0));
    if (st_cnt != RFX_OK && st_cnt != RFX_E_CAP) return st_cnt;
    if (tot[4] > 0) { ctx->last_error = "rfx_dev_sharded_count: a peer failed while counting its shard; see its rfx_last_error()"; return RFX_E_STATE; }
    if (out_totals) { out_totals[0] = tot[0]; out_totals[1] = tot[1]; out_totals[2] = tot[2]; }
    return tot[3] > 0 ? RFX_E_CAP : RFX_OK;
}

extern "C" {

static void add_timing(std::map<std::string, rfx_timing_slot> &acc, const std::map<std::string, rfx_timing_slot> &t) {
    for (auto &kv : t) { acc[kv.first].ms += kv.second.ms; acc[kv.first].launches += kv.second.launches; }
}

int rfx_dev_sharded_count(rfx_ctx *ctx, rfx_comm *c, const uint64_t *d_words, const uint32_t *d_read_len, int64_t n_reads,
                          int words_per_read, int read_len, int k, int front_clip, int end_clip, int generations, int min_cov,
                          int max_cov, int twin, uint64_t *d_out_keys, void *d_out_counts, int64_t cap, int64_t *out_n,
                          int64_t *out_totals) try {
    if (!ctx || !c || c->ctx != ctx || !d_words || n_reads < 0 || cap < 0 || generations < 1 || words_per_read * 32 < read_len)
        return RFX_E_ARG;
    comm_options(c);
    const bool wide = k > 32;
    if (wide && d_read_len) { ctx->last_error = "ragged reads: k <= 31 only on the device path"; return RFX_E_ARG; }
    if (wide ? (k > 63) : (k < 21 || k > 31)) {
        // outside the record path the k-mer instances themselves travel (one-word k-mers up to k = 31, the counter's three- and
        // four-word k-mers up to k = 125); k a multiple of 32 is no k of the reference's counters either
        if (d_read_len) { ctx->last_error = "ragged reads: k = 21..31 only on the sharded device path"; return RFX_E_ARG; }
        if (k < 3 || k > 125 || k % 32 == 0) { ctx->last_error = "rfx_dev_sharded_count: k = 3..125, not a multiple of 32"; return RFX_E_ARG; }
        RFX_HIP(hipSetDevice(ctx->device));
        return sharded_count_kmers(ctx, c, d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip, min_cov, max_cov, twin,
                                   d_out_keys, d_out_counts, cap, out_n, out_totals);
    }
    const int world = c->world, me = c->rank;
    int G = generations;
    // one-rank REHEARSAL of a larger node (RFX_COMM_VIRTUAL_WORLD=8 on a world of 1; bench.py --force-dist): the reads are
    // bucketed into G x 8 bins exactly as a rank of 8 would, and a generation's 8 owner bins -- contiguous in the send
    // buffer -- stand for what that rank would receive from its 8 peers (the same volume, statistically); nothing crosses a
    // link.  Results are unchanged (a k-mer still lives in exactly one generation).
    const int vworld = (world == 1 && c->virtual_world > 1) ? c->virtual_world : 1;
    while (G > 1 && G * world * vworld > 64) G /= 2;
    const int bins = G * world * vworld;
    const int uw = wide ? 4 : 2;                                  // 8-byte words per record
    RFX_HIP(hipSetDevice(ctx->device));
    NcclApi &n = nccl();
    std::map<std::string, rfx_timing_slot> acc;
    while ((int)c->ev.size() < G) {
        hipEvent_t e = nullptr;
        RFX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev.push_back(e);
    }

    // 1. records of this rank's reads, grouped by (generation, owner).  A failure HERE is this rank's alone (a bucket that
    // ran out of room three times, an allocation, a kernel): it is carried into the count matrix below, so that every
    // rank leaves the call together instead of the peers waiting in ncclAllGather for a rank that has gone.
    const int64_t nk = wide ? rfx::kmers_per_read_w(read_len, k, front_clip, end_clip) : rfx::kmers_per_read(read_len, k, front_clip, end_clip);
    int64_t inst = nk * n_reads;
    rfx::ReadStore rs{d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip};
    int64_t pb[TABW + 1], pe[TABW];
    for (int b = 0; b <= TABW; b++) pb[b] = 0;
    for (int b = 0; b < TABW; b++) pe[b] = 0;
    constexpr int S = 1;                                   // pieces per bin (alltoallv_words would take more)
    int64_t nrec = 0;
    auto bucket = [&]() -> int {
        if (d_read_len) {
            rs.read_len_arr = d_read_len;
            RFX_TRY(rfx::ragged_instances(ctx, d_read_len, n_reads, k, front_clip, end_clip, &rs.n_instances));
            inst = rs.n_instances;
        }
        if (c->units_per_read <= 0) c->units_per_read = (double)nk / 5.0 + 1.0;
        // the send buffer's layout: bin b = records [pb[b], pe[b]) (the two-pass form packs the bins back to back; level 1's
        // one sweep leaves the slack of its regions between them)
        const bool try_sweep = !(getenv("RFX_COMM_SWEEP") && atoi(getenv("RFX_COMM_SWEEP")) == 0);
        for (int attempt = 0;; attempt++) {
            const int64_t cap_rec = (int64_t)(c->units_per_read * (double)n_reads) + 4096;
            RFX_TRY(comm_grow(ctx, &c->send, &c->send_bytes, (size_t)cap_rec * uw * 8, ctx->stream, c->xs));
            ctx->timing.clear();
            int st;
            bool swept = false;
            if (wide) {
                st = try_sweep ? rfx::bucket_wide_records_by_owner_sweep(ctx, d_words, n_reads, words_per_read, nk, k, front_clip, bins, c->send,
                                                                         cap_rec, pb, pe, &nrec, &swept)
                               : RFX_OK;
                if (st == RFX_OK && !swept)
                    st = rfx_dev_bucket_wide_records_by_owner(ctx, d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip, bins,
                                                              c->send, cap_rec, c->d_tab, pb, &nrec);
                else
                    ScopedTimer::collect(ctx);
            } else {
                st = try_sweep ? rfx::bucket_records_by_owner_sweep(ctx, &rs, bins, c->send, cap_rec, pb, pe, &nrec, &swept) : RFX_OK;
                if (st == RFX_OK && !swept) st = rfx::bucket_records_by_owner(ctx, &rs, bins, c->send, cap_rec, c->d_tab, pb, &nrec);
                ScopedTimer::collect(ctx);
            }
            add_timing(acc, ctx->timing);
            if (st == RFX_OK) { if (!swept) for (int b = 0; b < bins; b++) pe[b] = pb[b + 1]; return RFX_OK; }
            if (st == RFX_E_CAP && attempt >= 2) {                                               // (RFX_E_CAP is the OUTPUT's word)
                ctx->last_error = "rfx_dev_sharded_count: the send buffer was too small three times running";
                return RFX_E_LIMIT;
            }
            if (st != RFX_E_CAP) return st;
            c->units_per_read = 1.03 * (double)nrec / (double)std::max<int64_t>(1, n_reads);      // only ever grows
        }
    };
    const int SP = S * vworld;                            // pieces per (generation, real rank)
    const int rbins = G * world;                          // bins of the real exchange
    const int np = rbins * SP;                            // pieces a rank holds
    const int row = np + 2;                               // + this rank's status, + the records its receive buffer takes
    if (row > TABW) { ctx->last_error = "rfx_dev_sharded_count: more pieces than the count matrix takes"; return RFX_E_STATE; }   // (same on every rank)
    int st_local = bucket();
    if (st_local != RFX_OK) { (void)hipStreamSynchronize(ctx->stream); for (int b = 0; b < TABW; b++) pb[b] = pe[b] = 0; nrec = 0; }
    c->bytes_bucketed = nrec * uw * 8;
    // the receive buffer is sized BEFORE the matrix travels, from what this rank sends (owners are balanced, so what
    // arrives is about what leaves), and its capacity travels with the counts: every rank can then see whether ANY
    // rank's buffer is short, and only then is there a second agreement (below) -- none in the steady state
    // ONE rank (a one-GPU run in G sequential generations: the strong-scaling denominator of bench.py, a set too large for
    // one fused count): nothing travels, a generation is counted where the bucketing left it -- no receive buffer, no copy
    const bool direct = world == 1 && vworld == 1 && !c->self_via_rccl;
    if (st_local == RFX_OK && !direct) {
        const int sg = comm_grow(ctx, &c->recv, &c->recv_bytes, (size_t)std::max<int64_t>(1, nrec + nrec / 8 + 4096) * uw * 8, ctx->stream, c->xs);
        if (sg != RFX_OK) st_local = sg;
    }

    // 2. every rank's counts to every rank: row r = what rank r holds in each piece of each (generation, owner) bin,
    // its status and its receive capacity.  (one-rank rehearsal: a generation's vworld owner bins -- vworld * S pieces --
    // all belong to the one real rank)
    int64_t *h_mine = c->h_tab, *h_all = c->h_tab + TABW;
    for (int q = 0; q < np; q++) h_mine[q] = pe[q] - pb[q];
    h_mine[np] = st_local;
    h_mine[np + 1] = (int64_t)(c->recv_bytes / ((size_t)uw * 8));
    RFX_HIP(hipMemcpyAsync(c->d_tab, h_mine, (size_t)row * 8, hipMemcpyHostToDevice, c->xs));
    RFX_NCCL(n.AllGather(c->d_tab, c->d_tab + TABW, (size_t)row, ncclInt64, c->comm, c->xs));
    RFX_HIP(hipMemcpyAsync(h_all, c->d_tab + TABW, (size_t)row * world * 8, hipMemcpyDeviceToHost, c->xs));
    RFX_HIP(hipStreamSynchronize(c->xs));
    note_foreign_hip_error(ctx, "rfx_dev_sharded_count (count matrix)");
    for (int r = 0; r < world; r++)
        if (h_all[(size_t)r * row + np] != RFX_OK) {                  // somebody could not bucket: everybody leaves, now
            if (st_local != RFX_OK) return st_local;
            char buf[160];
            snprintf(buf, sizeof buf, "rfx_dev_sharded_count: rank %d failed before the exchange (status %lld); see its rfx_last_error()", r,
                     (long long)h_all[(size_t)r * row + np]);
            ctx->last_error = buf;
            return RFX_E_STATE;
        }
    // receive layout: generation after generation, inside a generation source after source, piece after piece
    const size_t ne = (size_t)G * world * SP;
    std::vector<int64_t> gen_off(G + 1, 0), roff(ne), rcnt(ne), soff(ne), scnt(ne);
    int64_t mx = 0;
    for (int g = 0; g < G; g++) {
        int64_t pos = gen_off[g];
        for (int s = 0; s < world; s++)
            for (int q = 0; q < SP; q++) {
                const size_t e = ((size_t)g * world + s) * SP + q;
                const int64_t u = h_all[(size_t)s * row + ((size_t)g * world + me) * SP + q];
                roff[e] = pos * uw; rcnt[e] = u * uw;
                pos += u;
            }
        gen_off[g + 1] = pos;
        for (int p = 0; p < world; p++)
            for (int q = 0; q < SP; q++) {
                const size_t e = ((size_t)g * world + p) * SP + q;
                soff[e] = pb[e] * uw;
                scnt[e] = (pe[e] - pb[e]) * uw;
            }
    }
    bool any_short = false;                                         // (computed identically on every rank)
    for (int r = 0; r < world; r++) {
        int64_t need = 0;
        for (int s = 0; s < world; s++)
            for (int g = 0; g < G; g++)
                for (int q = 0; q < SP; q++) {
                    const int64_t u = h_all[(size_t)s * row + ((size_t)g * world + r) * SP + q];
                    need += u;
                    mx = std::max(mx, u * uw);
                }
        if (need > h_all[(size_t)r * row + np + 1] && !direct) any_short = true;
    }
    const int64_t rounds = std::max<int64_t>(1, (mx + (int64_t)(c->limit_bytes / 8) - 1) / (int64_t)(c->limit_bytes / 8));
    if (any_short) {                                                // the short ranks grow; everybody learns how that went
        int sg = RFX_OK;
        if (gen_off[G] > (int64_t)(c->recv_bytes / ((size_t)uw * 8)))
            sg = comm_grow(ctx, &c->recv, &c->recv_bytes, (size_t)std::max<int64_t>(1, gen_off[G]) * uw * 8, ctx->stream, c->xs);
        int64_t bad[1] = {sg != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, bad, 1, 1));
        if (sg != RFX_OK) return sg;
        if (bad[0]) { ctx->last_error = "rfx_dev_sharded_count: a peer could not allocate its receive buffer"; return RFX_E_STATE; }
    }

    // 3. all G exchanges queued back to back on the exchange stream
    RFX_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    RFX_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
    for (int g = 0; g < G && !direct; g++) {
        RFX_TRY(alltoallv_words(c, (const uint64_t *)c->send, &soff[(size_t)g * world * SP], &scnt[(size_t)g * world * SP], (uint64_t *)c->recv,
                                &roff[(size_t)g * world * SP], &rcnt[(size_t)g * world * SP], rounds, SP, c->xs));
        RFX_HIP(hipEventRecord(c->ev[g], c->xs));
    }

    // 4. count generation g while the later ones travel.  From here to the closing all-reduce a failure is remembered, not
    // returned: the peers are waiting there.
    int64_t m = 0, distinct = 0;
    int st_keep = RFX_OK, st_fail = RFX_OK;
    for (int g = 0; g < G; g++) {
        // generation g has landed: the CONTEXT'S STREAM waits for it, not the host (the count's first kernels queue up behind
        // the event while the host goes on; round 3 held the host here once per generation)
        if (!direct && hipStreamWaitEvent(ctx->stream, c->ev[g], 0) != hipSuccess) { if (st_fail == RFX_OK) { st_fail = RFX_E_HIP; ctx->last_error = "rfx_dev_sharded_count: hipStreamWaitEvent failed"; } continue; }
        note_foreign_hip_error(ctx, "rfx_dev_sharded_count (exchange)");      // (the kernels' launch checks below must not see RCCL's leftovers)
        if (st_fail != RFX_OK) continue;
        const int64_t ng = gen_off[g + 1] - gen_off[g];
        int64_t mg = 0, dg = 0;
        const int64_t room = std::max<int64_t>(0, cap - m);
        const void *src = direct ? (const void *)((const uint64_t *)c->send + pb[g] * uw) : (const void *)((const uint64_t *)c->recv + gen_off[g] * uw);
        ctx->timing.clear();
        int st;
        if (wide)
            st = rfx_dev_count_wide_records(ctx, src, ng, inst / G, k, min_cov, max_cov, d_out_keys + 2 * m,
                                            (int64_t *)d_out_counts + m, room, &mg, &dg);
        else
            st = rfx_dev_count_records(ctx, src, ng, inst / G, k, min_cov, max_cov, twin, d_out_keys + m,
                                       (int32_t *)d_out_counts + m, room, &mg, &dg);
        add_timing(acc, ctx->timing);
        if (st == RFX_E_CAP) { st_keep = RFX_E_CAP; m += mg; distinct += dg; continue; }     // keep draining: report the need
        if (st != RFX_OK) { st_fail = st; continue; }
        m += mg; distinct += dg;
    }
    if (hipStreamSynchronize(c->xs) != hipSuccess && st_fail == RFX_OK) { st_fail = RFX_E_HIP; ctx->last_error = "rfx_dev_sharded_count: the exchange stream failed"; }

    // 5. the generations' ascending shards -> one ascending shard
    if (st_fail == RFX_OK && st_keep == RFX_OK && G > 1 && m > 1) {
        ctx->timing.clear();
        auto merge = [&]() -> int {
            if (wide) {
                RFX_TRY(rfx_dev_order_kmers_w(ctx, d_out_keys, (int64_t *)d_out_counts, m, k));
                RFX_TRY(sync_checked(ctx));
            } else {
                DevBuf tk, tv;
                RFX_HIP(tk.alloc((size_t)m * 8, ctx->stream));
                RFX_HIP(tv.alloc((size_t)m * 4, ctx->stream));
                RFX_TRY(rfx_dev_sort_pairs(ctx, d_out_keys, (uint32_t *)d_out_counts, m, 2 * k, tk.as<uint64_t>(), tv.as<uint32_t>()));
                RFX_TRY(sync_checked(ctx));
            }
            return RFX_OK;
        };
        st_fail = merge();
        ScopedTimer::collect(ctx);                      // (order_wide2's "sort" timer: stopped timers never outlive the call)
        add_timing(acc, ctx->timing);
    }
    ctx->timing = acc;
    if (out_n) *out_n = m;

    // 6. global totals: instances, distinct, survivors, whether any rank ran out of room, whether any rank failed
    int64_t tot[5] = {inst, distinct, m, st_keep == RFX_E_CAP ? 1 : 0, st_fail != RFX_OK ? 1 : 0};
    RFX_TRY(rfx_comm_all_reduce_i64(c, tot, 5, 0));
    if (st_fail != RFX_OK) return st_fail;
    if (tot[4] > 0) { ctx->last_error = "rfx_dev_sharded_count: a peer failed while counting its shard; see its rfx_last_error()"; return RFX_E_STATE; }
    if (out_totals) { out_totals[0] = tot[0]; out_totals[1] = tot[1]; out_totals[2] = tot[2]; }
    // a shard that did not fit on ANY rank fails the call on EVERY rank (*out_n = this rank's own need), so that the
    // callers' retries stay collective
    return tot[3] > 0 ? RFX_E_CAP : RFX_OK;
} RFX_API_CATCH(ctx)

// The shards of every rank -> rank `root`, shard after shard in rank order (D' << N: the filtered list of a bacterial
// genome is a few million k-mers, so the extend stage runs on one GPU).  key_words 8-byte words per
// key, count_bytes 4 or 8.  On root: *out_n = total entries (RFX_E_CAP if cap is short); elsewhere 0.
int rfx_dev_gather_shards(rfx_ctx *ctx, rfx_comm *c, const uint64_t *d_keys, const void *d_counts, int64_t n, int key_words,
                          int count_bytes, int root, uint64_t *d_out_keys, void *d_out_counts, int64_t cap, int64_t *out_n) try {
    if (!ctx || !c || c->ctx != ctx || n < 0 || key_words < 1 || (count_bytes != 4 && count_bytes != 8) || root < 0 ||
        root >= c->world)
        return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    comm_options(c);
    NcclApi &nc = nccl();
    const int world = c->world, me = c->rank;
    // row = {entries, status, room}: a rank whose own stream has failed says so HERE, where the peers are listening, and
    // root's room travels too, so that every rank sees a short buffer on root before anybody posts a send
    const int st_sync = sync_checked(ctx);
    c->h_tab[0] = st_sync == RFX_OK ? n : 0;
    c->h_tab[1] = st_sync;
    c->h_tab[2] = cap;
    RFX_HIP(hipMemcpyAsync(c->d_tab, c->h_tab, 24, hipMemcpyHostToDevice, c->xs));
    RFX_NCCL(nc.AllGather(c->d_tab, c->d_tab + TABW, 3, ncclInt64, c->comm, c->xs));
    RFX_HIP(hipMemcpyAsync(c->h_tab + TABW, c->d_tab + TABW, (size_t)world * 24, hipMemcpyDeviceToHost, c->xs));
    RFX_HIP(hipStreamSynchronize(c->xs));
    const int64_t root_cap = c->h_tab[TABW + 3 * root + 2];
    for (int r = 0; r < world; r++)
        if (c->h_tab[TABW + 3 * r + 1] != RFX_OK) {
            if (st_sync != RFX_OK) return st_sync;
            ctx->last_error = "rfx_dev_gather_shards: a peer failed before the gather; see its rfx_last_error()";
            return RFX_E_STATE;
        }
    for (int r = 0; r < world; r++) c->h_tab[TABW + r] = c->h_tab[TABW + 3 * r];       // (compact: entries per rank)
    std::vector<int64_t> off(world + 1, 0);
    for (int r = 0; r < world; r++) off[r + 1] = off[r] + c->h_tab[TABW + r];
    if (out_n) *out_n = me == root ? off[world] : 0;
    if (off[world] > root_cap) return me == root ? RFX_E_CAP : RFX_OK;          // (the same verdict on every rank)
    const int64_t limit_k = std::max<int64_t>(1, (int64_t)(c->limit_bytes / 8) / key_words) , limit_c = (int64_t)(c->limit_bytes / count_bytes);
    const int64_t lim = std::min(limit_k, limit_c);
    int64_t mx = 0;
    for (int r = 0; r < world; r++) mx = std::max(mx, c->h_tab[TABW + r]);
    const int64_t rounds = std::max<int64_t>(1, (mx + lim - 1) / lim);
    if (me == root && n > 0) {
        RFX_HIP(hipMemcpyAsync(d_out_keys + off[me] * key_words, d_keys, (size_t)n * key_words * 8, hipMemcpyDeviceToDevice, c->xs));
        RFX_HIP(hipMemcpyAsync((char *)d_out_counts + off[me] * count_bytes, d_counts, (size_t)n * count_bytes, hipMemcpyDeviceToDevice, c->xs));
    }
    for (int64_t j = 0; j < rounds; j++) {
        GroupGuard gg;
        RFX_NCCL(nc.GroupStart());
        gg.open = true;
        if (me != root) {
            const int64_t s = std::max<int64_t>(0, std::min(lim, n - j * lim));
            if (s > 0) {
                RFX_NCCL(nc.Send(d_keys + j * lim * key_words, (size_t)s * key_words, ncclUint64, root, c->comm, c->xs));
                RFX_NCCL(nc.Send((const char *)d_counts + j * lim * count_bytes, (size_t)s * count_bytes, ncclUint8, root, c->comm, c->xs));
            }
        } else {
            for (int r = 0; r < world; r++) {
                if (r == root) continue;
                const int64_t s = std::max<int64_t>(0, std::min(lim, c->h_tab[TABW + r] - j * lim));
                if (s > 0) {
                    RFX_NCCL(nc.Recv(d_out_keys + (off[r] + j * lim) * key_words, (size_t)s * key_words, ncclUint64, r, c->comm, c->xs));
                    RFX_NCCL(nc.Recv((char *)d_out_counts + (off[r] + j * lim) * count_bytes, (size_t)s * count_bytes, ncclUint8, r, c->comm, c->xs));
                }
            }
        }
        gg.open = false;
        RFX_NCCL(nc.GroupEnd());
    }
    RFX_HIP(hipStreamSynchronize(c->xs));
    note_foreign_hip_error(ctx, "rfx_dev_gather_shards");
    return RFX_OK;
} RFX_API_CATCH(ctx)

// The whole resident path on several GPUs from ASCII reads in host memory: every rank uploads and encodes ITS reads
// (any lengths), rfx_dev_sharded_count, the shards gathered on rank 0, the driver there (rfx_dev_assemble) -> the
// contig text on rank 0 (*out_len = 0 elsewhere).  k = 21..31.  Collective.  out_totals[3] as rfx_dev_sharded_count.
int rfx_sharded_assemble_reads(rfx_ctx *ctx, rfx_comm *c, const uint8_t *bases, const int64_t *read_off, int64_t n_reads,
                               const rfx_params *prm, int generations, int64_t gather_below, char *out, int64_t cap, int64_t *out_len,
                               int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace, int64_t *out_totals) try {
    if (!ctx || !c || c->ctx != ctx || !read_off || !prm || !out_len || n_reads < 0) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    const int k = prm->k;
    const int64_t nb = n_reads ? read_off[n_reads] - read_off[0] : 0;
    int64_t maxlen = 1;
    for (int64_t r = 0; r < n_reads; r++) maxlen = std::max(maxlen, read_off[r + 1] - read_off[r]);
    const int wpr = (int)((maxlen + 31) / 32);
    *out_len = 0;
    if (out_contigs) *out_contigs = 0;
    if (n_trace) *n_trace = 0;
    DevBuf d_bases, d_off, d_words, d_len, d_keys, d_counts, g_keys, g_counts;
    // this rank's own preparation (allocations, upload, encode); how it went is agreed before the first data collective,
    // so that a rank that cannot go on does not leave its peers waiting inside one
    auto prepare = [&]() -> int {
        RFX_HIP(d_bases.alloc((size_t)std::max<int64_t>(nb, 1), ctx->stream));
        RFX_HIP(d_off.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
        RFX_HIP(d_words.alloc((size_t)std::max<int64_t>(n_reads, 1) * wpr * 8, ctx->stream));
        RFX_HIP(d_len.alloc((size_t)std::max<int64_t>(n_reads, 1) * 4, ctx->stream));
        std::vector<int64_t> off((size_t)n_reads + 1, 0);
        for (int64_t r = 0; r <= n_reads && n_reads > 0; r++) off[(size_t)r] = read_off[r] - read_off[0];
        if (nb > 0) RFX_HIP(hipMemcpyAsync(d_bases.p, bases + read_off[0], (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_TRY(rfx::encode_reads(ctx, d_bases.as<uint8_t>(), d_off.as<int64_t>(), n_reads, wpr, d_words.as<uint64_t>(), d_len.as<uint32_t>()));
        RFX_TRY(sync_checked(ctx));                                     // (`off` is read by the copy until here)
        d_bases.release(); d_off.release();
        return RFX_OK;
    };
    int st_local;
    try { st_local = prepare(); } catch (...) { st_local = rfx_api_exception(ctx, "rfx_sharded_assemble_reads"); }
    if (st_local != RFX_OK) (void)hipStreamSynchronize(ctx->stream);
    auto agree = [&](int st_mine, const char *what) -> int {           // collective: RFX_OK only if every rank says so
        int64_t bad[1] = {st_mine != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, bad, 1, 1));
        if (st_mine != RFX_OK) return st_mine;
        if (bad[0]) { ctx->last_error = std::string("rfx_sharded_assemble_reads: a peer failed (") + what + "); see its rfx_last_error()"; return RFX_E_STATE; }
        return RFX_OK;
    };
    RFX_TRY(agree(st_local, "upload / encode"));
    int64_t kcap = std::max<int64_t>(1 << 20, nb / 8), m = 0, tot[3] = {0, 0, 0};
    for (;;) {                                                        // survivors are few; every rank grows together
        int st = RFX_OK;
        if (d_keys.alloc((size_t)kcap * 8, ctx->stream) != hipSuccess || d_counts.alloc((size_t)kcap * 4, ctx->stream) != hipSuccess) {
            ctx->last_error = "rfx_sharded_assemble_reads: no room for the shard";
            st = RFX_E_HIP;
        }
        RFX_TRY(agree(st, "shard allocation"));
        st = rfx_dev_sharded_count(ctx, c, d_words.as<uint64_t>(), d_len.as<uint32_t>(), n_reads, wpr, (int)maxlen, k, prm->front_clip,
                                   prm->end_clip, generations, prm->min_cov, prm->max_cov, prm->twin, d_keys.as<uint64_t>(),
                                   d_counts.p, kcap, &m, tot);
        if (st == RFX_E_CAP) {                                        // (on every rank at once)
            int64_t need[1] = {m};
            RFX_TRY(rfx_comm_all_reduce_i64(c, need, 1, 1));
            kcap = std::max(kcap * 2, need[0]);
            continue;
        }
        RFX_TRY(st);                                                  // (any other failure is every rank's, see there)
        break;
    }
    if (out_totals) { out_totals[0] = tot[0]; out_totals[1] = tot[1]; out_totals[2] = tot[2]; }
    d_words.release(); d_len.release();
    // the extend stage: the range shuffle of sortByKey over the ranks while the record set is larger than `gather_below`,
    // the rest on rank 0 (rfx_shard.hip; a bacterial genome's few million survivors go to rank 0 at once).  Its outcome is
    // every rank's: a text buffer that is too short on rank 0 is RFX_E_CAP with *out_len = the length needed on EVERY rank,
    // so that the callers' "grow and call again" re-enters the collective together.
    return rfx_dev_sharded_assemble(ctx, c, d_keys.as<uint64_t>(), d_counts.as<int32_t>(), m, prm, gather_below, out, cap, out_len, out_contigs,
                                    trace, trace_cap, n_trace);
} RFX_API_CATCH(ctx)

}  // extern "C"
