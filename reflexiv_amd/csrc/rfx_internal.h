// rfx_internal.h -- host-side plumbing shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/reflexiv_hip.h"

struct rfx_timing_slot { float ms = 0.f; int64_t launches = 0; };
// a kernel timer that has been stopped and not yet read (ScopedTimer below): events of the context's own pool
struct rfx_timer_pending { const char *name; hipEvent_t a, b; int64_t launches; };

// RFX_POISON=1 (debugging): every fresh workspace slot and every scratch allocation is filled with 0xA5 before it is handed
// out, so a kernel that reads what nobody wrote sees the same junk in a fresh process as deep into a test session.
// A bit mask: 1 = fresh workspace slots, 2 = scratch allocations (DevBuf), 4 = workspace slots 0 and 1 each time they are
// handed out again.
inline int rfx_poison() {
    static const int on = [] { const char *e = getenv("RFX_POISON"); return e && *e ? atoi(e) : 0; }();
    return on;
}

struct rfx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    std::string foreign_hip_error;        // a HIP error RCCL left on this thread (rfx_comm.hip note_foreign_hip_error): reported, never fatal
    std::string last_error_text;          // what rfx_last_error() hands out (last_error + the note above)
    int num_cu = 256;
    // per-kernel-family timing of the last count call (HIP events on `stream`)
    std::map<std::string, rfx_timing_slot> timing;
    bool timing_enabled = true;
    // stopped timers waiting for ScopedTimer::collect.  They belong to the CONTEXT, like the events they name: until round 4
    // this list was a thread_local, so a timer stopped on a path that never collected (order_wide2 through
    // rfx_dev_order_kmers_w, any early error return) outlived its context, and the next context's first collect() handed
    // destroyed events to hipEventElapsedTime -- the host abort of gpurun_out/s3_both.log.
    std::vector<rfx_timer_pending> timers_pending;
    // grow-only workspace slots for the large, reused buffers (instance arrays of the count
    // stage): allocated once with hipMalloc, kept until the context dies
    struct WsSlot { void *p = nullptr; size_t bytes = 0; };
    WsSlot ws[7];          // 0, 1: the count stage's record / instance buffers; 2, 3: the extend stage's arenas; 4: the count stage's arena;
                           // 5, 6: rfx_assemble_reads' ASCII staging and packed reads (multi-gigabyte stream-ordered allocations of
                           // changing sizes cost 100-600 ms a call in the pool; these are allocated once and kept)
    void *ws_get(int slot, size_t bytes) {
        WsSlot &w = ws[slot];
        if (w.bytes >= bytes && w.p) {
            // slots 0 and 1 are only ever asked for as the DESTINATION of the next kernel (rfx_kmer.hip)
            if ((rfx_poison() & 4) && slot < 2) (void)hipMemsetAsync(w.p, 0xA5, bytes, stream);
            return w.p;
        }
        if (w.p) { (void)hipStreamSynchronize(stream); (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
        size_t want = bytes + (bytes >> 4) + (1 << 20);
        if (hipMalloc(&w.p, want) != hipSuccess) { w.p = nullptr; return nullptr; }
        if (rfx_poison() & 1) (void)hipMemsetAsync(w.p, 0xA5, want, stream);
        if (rfx_poison() & 8) { (void)hipMemset(w.p, 0, want); (void)hipDeviceSynchronize(); }      // (8: zeros, the same way)
        w.bytes = want;
        return w.p;
    }
    void ws_free() {
        for (auto &w : ws) { if (w.p) (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr; pinned_bytes = 0;
        if (scan_desc) (void)hipFree(scan_desc);
        if (scan_ticket) (void)hipFree(scan_ticket);
        if (scan_fault) (void)hipHostFree(scan_fault);
        scan_fault = nullptr;
        if (mailbox) (void)hipHostFree((void *)mailbox);
        mailbox = nullptr;
        scan_desc = nullptr; scan_ticket = nullptr; scan_desc_cap = 0;
        timers_pending.clear();
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear(); ev_next = 0;
    }
    // grow-only pinned host staging for result downloads (a pageable copy of a few MB costs
    // milliseconds the first time the runtime stages it)
    // events of the kernel timers: created once, handed out round-robin until the next collect()
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;
    hipEvent_t ev_get() {
        if (ev_next == ev_pool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev_pool.push_back(e); }
        return ev_pool[ev_next++];
    }
    // single-pass scans (rfx_scan.hip): tile descriptors and the ticket counter live in the context; a descriptor is
    // valid only under the epoch of the call that wrote it, so nothing is cleared between calls
    uint64_t *scan_desc = nullptr;
    size_t scan_desc_cap = 0;              // descriptors (two per tile for the dual scan)
    unsigned long long *scan_ticket = nullptr;
    int *scan_fault = nullptr;             // host-mapped: set by a look-back that gave up (never expected)
    unsigned long long scan_tickets_issued = 0;
    uint32_t scan_epoch = 0;
    // A mailbox in host-mapped pinned memory: the last (one-workgroup) kernel of a chain writes up to seven values and then a
    // sequence number into it, the host spins on the number instead of queueing a 24-byte copy and waiting for the stream
    // (mailbox_wait below).  The extend stage waits twice per pass for a handful of bytes; a stream wait + the copy cost 25-40 us
    // each, the mailbox a few.  RFX_MAILBOX=0: the copies again.
    volatile uint64_t *mailbox = nullptr;          // [0] sequence, [1..7] values
    uint64_t mailbox_seq = 0;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    void *pinned_get(size_t bytes) {
        if (pinned && pinned_bytes >= bytes) return pinned;
        if (pinned) { (void)hipStreamSynchronize(stream); (void)hipHostFree(pinned); pinned = nullptr; pinned_bytes = 0; }
        size_t want = bytes + (bytes >> 2) + (1 << 20);
        if (hipHostMalloc(&pinned, want, hipHostMallocDefault) != hipSuccess) { pinned = nullptr; return nullptr; }
        pinned_bytes = want;
        return pinned;
    }
};

#define RFX_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            char buf_[512];                                                               \
            snprintf(buf_, sizeof buf_, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,     \
                     hipGetErrorString(e_));                                              \
            if (ctx) ctx->last_error = buf_;                                              \
            return RFX_E_HIP;                                                             \
        }                                                                                 \
    } while (0)

// The boundary keeps its word ("never aborts", include/reflexiv_hip.h): every extern "C" entry point is a function-try-block
// that ends in one of these, so a C++ exception raised under it (std::bad_alloc from a vector or a map, std::system_error
// from a thread, std::length_error ...) becomes RFX_E_HOST with its what() in rfx_last_error() instead of std::terminate
// inside the JVM / Python process that loaded the library.
int rfx_api_exception(rfx_ctx *ctx, const char *where) noexcept;
#define RFX_API_CATCH(ctxexpr) catch (...) { return rfx_api_exception((ctxexpr), __func__); }
#define RFX_API_CATCH_VOID(ctxexpr) catch (...) { (void)rfx_api_exception((ctxexpr), __func__); }

#define RFX_TRY(call)                              \
    do {                                           \
        int s_ = (call);                           \
        if (s_ != RFX_OK) return s_;               \
    } while (0)

// Every host wait on the context's stream goes through here: a single-pass scan whose look-back gave up (rfx_scan.hip:
// it substitutes prefix 0 and raises the host-mapped flag) fails the call that waited, not some later one.
static inline int sync_checked(rfx_ctx *ctx) {
    hipError_t e_ = hipStreamSynchronize(ctx->stream);
    if (e_ != hipSuccess) {
        ctx->last_error = std::string("hipStreamSynchronize -> ") + hipGetErrorString(e_);
        return RFX_E_HIP;
    }
    if (ctx->scan_fault && *ctx->scan_fault) {
        *ctx->scan_fault = 0;
        ctx->last_error = "scan: look-back gave up waiting for a predecessor tile (offsets of this call are invalid)";
        return RFX_E_HIP;
    }
    return RFX_OK;
}

// The mailbox (see rfx_ctx): mailbox_next() -> the number the kernel must post (0: no mailbox, use a copy + sync_checked);
// mailbox_wait() spins for it -- everything queued before the posting kernel has completed by then -- and falls back to a
// stream wait after two seconds (a faulted kernel never posts) or when the stream already reports an error.
static inline uint64_t mailbox_next(rfx_ctx *ctx) {
    static const bool off = [] { const char *e = getenv("RFX_MAILBOX"); return e && atoi(e) == 0; }();
    if (off) return 0;
    if (!ctx->mailbox) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return 0;
        memset(p, 0, 64);
        ctx->mailbox = (volatile uint64_t *)p;
        ctx->mailbox_seq = 0;
    }
    return ++ctx->mailbox_seq;
}
static inline int mailbox_wait(rfx_ctx *ctx, uint64_t seq, uint64_t *vals, int nvals) {
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t spins = 0;; spins++) {
        if (ctx->mailbox[0] == seq) break;
        if ((spins & 0xFFF) == 0xFFF) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000 > 2000) {
                const int st = sync_checked(ctx);
                if (st != RFX_OK) return st;
                if (ctx->mailbox[0] != seq) { ctx->last_error = "mailbox: the posting kernel finished without posting"; return RFX_E_STATE; }
                break;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < nvals; i++) vals[i] = ctx->mailbox[1 + i];
    if (ctx->scan_fault && *ctx->scan_fault) {
        *ctx->scan_fault = 0;
        ctx->last_error = "scan: look-back gave up waiting for a predecessor tile (offsets of this call are invalid)";
        return RFX_E_HIP;
    }
    return RFX_OK;
}

// up to 56 bytes of device memory to the host through the mailbox (one tiny launch instead of a queued copy + a stream wait);
// anything larger, or with the mailbox off: the copy and sync_checked.  Everything queued before it has completed on return.
int small_readback(rfx_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes);

// Bump arena for the temporaries of one pass of the extend loop: two of them alternate, so a
// pass reads its inputs from the previous pass's arena and nothing is allocated or freed per pass
// (hipMallocAsync with ever-changing sizes cost ~5 ms per pass, far more than the kernels).
struct Arena { char *base = nullptr; size_t cap = 0, off = 0; };
inline thread_local Arena *tl_arena = nullptr;

// Stream-ordered scratch allocation that frees itself (or a slice of the current arena).
struct DevBuf {
    void *p = nullptr;
    hipStream_t s = nullptr;
    bool borrowed = false;        // points into a context workspace slot: never freed here
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; s = o.s; borrowed = o.borrowed; o.p = nullptr; o.borrowed = false; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t alloc(size_t bytes, hipStream_t stream) {
        release();
        s = stream;
        if (bytes == 0) bytes = 16;
        if (tl_arena) {
            const size_t need = (bytes + 255) & ~(size_t)255;
            if (tl_arena->off + need <= tl_arena->cap) {
                p = tl_arena->base + tl_arena->off;
                tl_arena->off += need;
                borrowed = true;
                if (rfx_poison() & 2) (void)hipMemsetAsync(p, 0xA5, bytes, stream);
                return hipSuccess;
            }
        }
        hipError_t e = hipMallocAsync(&p, bytes, stream);
        if (e == hipSuccess && (rfx_poison() & 2)) (void)hipMemsetAsync(p, 0xA5, bytes, stream);
        return e;
    }
    void release() {
        if (p && !borrowed) (void)hipFreeAsync(p, s);
        p = nullptr; borrowed = false;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// The temporaries of one count-stage call (histogram tables, scans, hole lists, sort buffers) from one bump arena the
// context keeps (workspace slot 4): the stream-ordered allocator left the GPU idle for half a millisecond between the
// levels of every step (30.9 -> 28.6 ms per step).  Nothing if an arena is already active; a request the arena cannot
// take goes to the stream-ordered allocator as before.
struct StageArena {
    Arena arena;
    Arena *prev;
    bool on = false;
    StageArena(rfx_ctx *ctx, size_t bytes) : prev(tl_arena) {
        if (tl_arena || (getenv("RFX_COUNT_ARENA") && atoi(getenv("RFX_COUNT_ARENA")) == 0)) return;
        arena.base = (char *)ctx->ws_get(4, bytes);
        if (arena.base) { arena.cap = bytes; tl_arena = &arena; on = true; }
    }
    ~StageArena() { if (on) tl_arena = prev; }
    StageArena(const StageArena &) = delete;
    StageArena &operator=(const StageArena &) = delete;
};

// Event pair that accumulates into ctx->timing[name].
struct ScopedTimer {
    rfx_ctx *ctx; const char *name; hipEvent_t a = nullptr, b = nullptr; bool on;
    ScopedTimer(rfx_ctx *c, const char *n) : ctx(c), name(n), on(c && c->timing_enabled && !getenv("RFX_NO_TIMING")) {
        if (on) { a = ctx->ev_get(); b = ctx->ev_get(); on = a && b; }
        if (on) (void)hipEventRecord(a, ctx->stream);
    }
    void stop(int64_t launches = 1) {
        if (!on) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->timers_pending.push_back({name, a, b, launches});
        on = false;
    }
    ~ScopedTimer() { stop(); }
    // call after a stream sync (a timer whose events have not completed is dropped, not waited for)
    static void collect(rfx_ctx *ctx) {
        for (auto &p : ctx->timers_pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                ctx->timing[p.name].ms += ms;
                ctx->timing[p.name].launches += p.launches;
            }
        }
        ctx->timers_pending.clear();
        ctx->ev_next = 0;                  // the pool's events are free again
    }
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

namespace rfx {

// ---- rfx_scan.hip
int exclusive_scan_u64(rfx_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int64_t n);   // out[n] = total (n+1 entries)
int exclusive_scan_u32_to_u64(rfx_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, int64_t n);
// two arrays in one launch (the emission indices and the word offsets of an extend pass)
int exclusive_scan2_u32_to_u64(rfx_ctx *ctx, const uint32_t *d_in_a, const uint32_t *d_in_b, uint64_t *d_out_a,
                               uint64_t *d_out_b, int64_t n);

// ---- rfx_sort.hip
int sort_pairs(rfx_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int64_t n, int key_bits,
               uint64_t *d_tmp_keys, uint32_t *d_tmp_vals);   // result in d_keys/d_vals

// ---- rfx_kmer.hip
struct ReadStore {
    const uint64_t *words; int64_t n_reads; int words_per_read; int read_len;
    int k, front_clip, end_clip;
    // ragged reads: per-read lengths in HBM (read_len is then the maximum) and the exact number of
    // k-mer instances (-1: uniform reads, kmers_per_read(read_len) each)
    const uint32_t *read_len_arr = nullptr;
    int64_t n_instances = -1;
};
int ragged_instances(rfx_ctx *ctx, const uint32_t *d_read_len, int64_t n_reads, int k, int front_clip,
                         int end_clip, int64_t *out_total);
int64_t kmers_per_read(int read_len, int k, int front_clip, int end_clip);
int encode_reads(rfx_ctx *ctx, const uint8_t *d_bases, const int64_t *d_read_off, int64_t n_reads,
                 int words_per_read, uint64_t *d_words, uint32_t *d_read_len);
int extract_ordered_packed(rfx_ctx *ctx, const uint64_t *d_words, int wpr, const uint64_t *d_kmer_off,
                           int64_t n_reads, int k, int front_clip, uint64_t *d_out);
int kmer_counts_per_read(rfx_ctx *ctx, const int64_t *d_read_off, int64_t n_reads, int k,
                         int front_clip, int end_clip, uint64_t *d_nk);
int64_t count_workspace_bytes(int64_t n_kmers);
int count_filter(rfx_ctx *ctx, const ReadStore *reads, const uint64_t *d_kmers, int64_t n,
                 int min_cov, int max_cov, int twin, void *ws, int64_t ws_bytes,
                 uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                 int64_t *out_n, int64_t *out_distinct, bool pair_out = false);
int bucket_pairs_by_owner(rfx_ctx *ctx, const void *d_pairs, int64_t n, int n_owners, void *d_out,
                          int64_t *d_owner_off, int64_t *h_owner_off);
int merge_pairs(rfx_ctx *ctx, const void *d_pairs, int64_t n, int k, int min_cov, int max_cov, int twin,
                uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct);
int bucket_by_owner(rfx_ctx *ctx, const ReadStore *reads, int n_owners, uint64_t *d_out,
                    int64_t cap, int64_t *d_owner_off, int64_t *h_owner_off);
int bucket_records_by_owner_sweep(rfx_ctx *ctx, const ReadStore *reads, int n_owners, void *d_out, int64_t cap_records,
                                  int64_t *h_begin, int64_t *h_end, int64_t *out_n_records, bool *done);
int bucket_wide_records_by_owner_sweep(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                                       int n_owners, void *d_out, int64_t cap_records, int64_t *h_begin, int64_t *h_end,
                                       int64_t *out_n_records, bool *done);
int bucket_records_by_owner(rfx_ctx *ctx, const ReadStore *reads, int n_owners, void *d_out, int64_t cap_records,
                            int64_t *d_owner_off, int64_t *h_owner_off, int64_t *out_n_records);
int count_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k,
                  int min_cov, int max_cov, int twin, uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                  int64_t *out_n, int64_t *out_distinct);
int bucket_wide_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                                 int n_owners, void *d_out, int64_t cap_records, int64_t *d_owner_off, int64_t *h_owner_off,
                                 int64_t *out_n_records);
int count_wide_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k, int min_cov,
                       int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                       int64_t *out_distinct);
int count_wide2(rfx_ctx *ctx, const void *d_elems, int64_t n, int min_cov, int max_cov, uint64_t *d_out_keys,
                int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct);
int count_wide2_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                      int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                      int64_t *out_distinct);
int bucket_wide_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                         int n_owners, void *d_out, int64_t cap_elems, int64_t *d_owner_off, int64_t *h_owner_off);
int synth_genome(rfx_ctx *ctx, uint64_t seed, int64_t genome_len, uint64_t *d_genome);
int synth_reads(rfx_ctx *ctx, uint64_t seed, const uint64_t *d_genome, int64_t genome_len,
                int64_t first_read, int64_t n_reads, int read_len, uint32_t err, int words_per_read,
                uint64_t *d_words);

// ---- rfx_wide.hip : k > 31, W = k/32+1 words per k-mer (the counter's layout)
int check_k_w(int k);
int64_t kmers_per_read_w(int read_len, int k, int fc, int ec);
int kmer_counts_per_read_w(rfx_ctx *ctx, const int64_t *d_read_off, int64_t n_reads, int k, int fc, int ec,
                           uint64_t *d_nk);
int extract_w(rfx_ctx *ctx, const uint64_t *d_words, int wpr, const uint64_t *d_kmer_off, int64_t nk_uniform,
              int64_t n_reads, int k, int fc, uint64_t *d_soa, int64_t N, int aos = 0);
bool wide_fast_path(int k);
int order_wide2(rfx_ctx *ctx, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t m, int k);
int count_filter_w2(rfx_ctx *ctx, const uint64_t *d_elems, int64_t N, int k, int min_cov, int max_cov,
                    uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct);
int aos_to_soa(rfx_ctx *ctx, const uint64_t *d_aos, int64_t n, int W, uint64_t *d_soa);
int soa_to_aos(rfx_ctx *ctx, const uint64_t *d_soa, int64_t n, int W, uint64_t *d_aos);
int count_filter_w(rfx_ctx *ctx, uint64_t *d_soa, int64_t N, int k, int min_cov, int max_cov, uint64_t *d_out_keys,
                   int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct);

// ---- rfx_graph.hip : device record set (reference layout, in HBM)
struct DevRecords {
    int64_t n = 0, words = 0;
    int kw = 1;                 // words per key (AoS): 1 up to k = 32, (k-2)/31+1 beyond
    DevBuf key, marker, ext_off, ext, left, right;
};
inline int sub_words(int k) { return k <= 32 ? 1 : (k - 2) / 31 + 1; }     // subKmerBinarySlots, U/DefaultParam.java:94
inline int asm_words(int k) { return k <= 31 ? 1 : (k - 1) / 31 + 1; }     // kmerBinarySlotsAssemble, U/DefaultParam.java:85
constexpr int MAX_KEY_WORDS = 4;                                           // k <= 125
int dev_records_alloc(rfx_ctx *ctx, DevRecords &r, int64_t cap_n, int64_t cap_words, int kw = 1);
int dev_records_upload(rfx_ctx *ctx, const rfx_records *h, DevRecords &d);
int dev_records_download(rfx_ctx *ctx, const DevRecords &d, rfx_records *h);

int rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n,
                      int k, DevRecords &out);
// key_bits: significant bits of a one-word key (ignored for kw > 1: k gives the word widths)
int sort_records(rfx_ctx *ctx, const DevRecords &in, int P, int key_bits, DevRecords &out,
                 DevBuf &part_start /* int64[P+1] */, int k = 0);
int counter_to_asm(rfx_ctx *ctx, const uint64_t *d_keys32, const int64_t *d_counts64, int64_t n, int k, int min_cov,
                   int max_cov, uint64_t *d_out31, int32_t *d_out_counts, int64_t *out_n);
int fork_filter(rfx_ctx *ctx, bool reflected, const DevRecords &in, const int64_t *d_part_start,
                int P, int k, int min_error_cov, int twin, DevRecords &out,
                DevBuf &out_part_start);
int reflect_from_forward(rfx_ctx *ctx, const DevRecords &in, int k, DevRecords &out);
// Several GPUs (rfx_shard.hip): a logical partition may begin on an earlier rank.  What it brings from there is the PARITY
// of a count -- records for the random reflection, emissions for an extend pass -- as device int32[2] = {partition, parity}.
// The extend pass learns its count only after its scan, so it calls back between the scan and the emission.
struct PartCarry {
    virtual ~PartCarry() {}
    // d_cum: n + 1 prefix counts of the quantity whose parity decides (nullptr: the record index itself)
    virtual int compute(rfx_ctx *ctx, const int64_t *d_part_start, int P, const uint64_t *d_cum, int64_t n, const int32_t **d_carry) = 0;
};
int random_reflection(rfx_ctx *ctx, const DevRecords &in, const int64_t *d_part_start, int P,
                      int k, DevRecords &out, const int32_t *d_carry = nullptr);
int extend_pass(rfx_ctx *ctx, const DevRecords &in, const int64_t *d_part_start, int P, int k,
                int twin, int stage, DevRecords &out, DevBuf &out_part_start, int start_marker = 2, PartCarry *carry_hook = nullptr);

// the k > 31 from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712): op 0 DSReflexivAndForwardKmer (2n out),
// 1 DSFilterExtendableKmerPairs, 2 DSFilterUnExtendableKmer, 3 DSFilterStillExtendableKmerFromPairs,
// 4 DSFilterStillExtendableKmerEnds, 5 / 6 DSFilterUnExtendableKmerLeftEnds / ...RightEnds
int extras_operator(rfx_ctx *ctx, int op, const DevRecords &in, const int64_t *d_part_start, int P, int k, DevRecords &out,
                    DevBuf &out_part_start);
// the driver (rfx_api.hip), and the state with which the sharded driver (rfx_shard.hip) hands its loop over to it
struct AsmResume {
    DevRecords *recs;            // the record set, in global arrival order (its buffers are taken over)
    int passes_done;             // extend passes behind it (0..4: the passes before the loop are still to come)
    int iterations;
    int64_t contig_number;
    int scramble, P, partition_number;
    int64_t nt;                  // trace entries already written
};
int assemble_impl(rfx_ctx *ctx, bool wide, const uint64_t *d_keys, const int32_t *d_counts, int64_t n, const rfx_params *prm,
                  char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs, int64_t *trace, int64_t trace_cap,
                  int64_t *n_trace, AsmResume *resume);
// the rest of the driver's loop on <= small_pass_limit() records: two launches per pass, state in HBM (rfx_extend.hip)
int small_passes(rfx_ctx *ctx, DevRecords &recs, int k, int twin, bool wide, int coalesce, int min_iter, int max_iter,
                 int *iterations, int64_t *contig_number, int *scramble, int *P, int *partition_number,
                 int64_t *trace, int64_t trace_cap, int64_t *nt);
int small_pass_limit();
int small_pass_max_partitions();

}  // namespace rfx
