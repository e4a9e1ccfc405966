// rfx_api.hip -- the extern "C" boundary of libreflexiv_hip.so (include/reflexiv_hip.h) and
// the driver loop that mirrors ReflexivMain.assembly() (P/ReflexivMain.java:168-310).
#include <system_error>
#include <thread>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "rfx_internal.h"
#include <algorithm>
#include <cstdlib>
#include <ctime>

using namespace rfx;

namespace {

int check_k(int k) { return (k >= 3 && k <= 31) ? RFX_OK : RFX_E_ARG; }
// record operators: any k whose (k-1)-mer key fits MAX_KEY_WORDS words of 31 bases
int check_k_rec(int k) { return (k >= 3 && sub_words(k) <= MAX_KEY_WORDS) ? RFX_OK : RFX_E_ARG; }
int check_kw(const rfx_records *r, int k) { return (r->key_words > 1 ? r->key_words : 1) == sub_words(k) ? RFX_OK : RFX_E_ARG; }

int download_to(rfx_ctx *ctx, const DevRecords &d, rfx_records *out) {
    if (!out) return RFX_E_ARG;
    out->need_n = d.n; out->need_words = d.words;
    if (d.n > out->cap_n || d.words > out->cap_words) return RFX_E_CAP;
    return dev_records_download(ctx, d, out);
}

int upload_part_start(rfx_ctx *ctx, const int64_t *h, int P, DevBuf &d) {
    RFX_HIP(d.alloc((size_t)(P + 1) * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d.p, h, (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}
int download_part_start(rfx_ctx *ctx, const DevBuf &d, int P, int64_t *h) {
    if (!h) return RFX_OK;
    RFX_HIP(hipMemcpyAsync(h, d.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

// a-15 on the host (tiny): BinaryReflexivKmerArrayToString + KmerToContig + TagContigID
// k > 31 (P/ReflexivDSMain64.java:830-866, 1913-1975): the RDD-style header, no skip rule
int64_t contigs_text_host(const rfx_records *r, int k, int min_contig, int twin, char *out, int64_t cap,
                          int64_t *n_contigs) {
    static const char NUC[4] = {'A', 'C', 'G', 'T'};
    // four bases per byte; built once under the language's thread-safe initialisation of a function-local static
    // (several host threads format contigs at once: Spark executor threads through the JNI, one context each)
    struct QuadTable {
        char q[256][4];
        QuadTable() { for (int b = 0; b < 256; b++) for (int j = 0; j < 4; j++) q[b][j] = NUC[(b >> (6 - 2 * j)) & 3]; }
    };
    static const QuadTable quad_table;
    const char (*QUAD)[4] = quad_table.q;
    const int sub = k - 1;
    const int kw = r->key_words > 1 ? r->key_words : 1;
    if (k > 31) twin = RFX_TWIN_RDD;
    // Large outputs that fit `out` (a bacterial genome is 9 MB of text, 1.1 ms on one core): the layout is known before
    // a base is written -- header, then base j of the contig at body + j + j / 100 -- so a few threads decode disjoint
    // ranges of extension words straight to their places.  Anything else (small, or only sized / clipped) below.
    {
        struct Job { int64_t i, body, len, L; int f; };
        std::vector<Job> jobs;
        int64_t at = 0, id2 = 0, words = 0;
        std::vector<std::string> hdrs;
        for (int64_t i = 0; i < r->n; i++) {
            if (twin == RFX_TWIN_DS && r->left[i] <= -10000000 && r->right[i] <= -10000000) continue;
            const uint64_t *w = r->ext + r->ext_off[i];
            const int64_t nw = r->ext_off[i + 1] - r->ext_off[i];
            const int nlz = w[0] ? __builtin_clzll(w[0]) : 64;
            const int f = 32 - (nlz / 2 + 1);
            const int64_t L = (nw - 1) * 31 + f, len = L + sub;
            if (len < min_contig) continue;
            char hdr[96];
            const int hl = twin == RFX_TWIN_DS
                               ? snprintf(hdr, sizeof hdr, ">Contig-%lld-(%d,%d)-%lld\n", (long long)len, r->left[i], r->right[i], (long long)id2)
                               : snprintf(hdr, sizeof hdr, ">Contig-%lld-%lld\n", (long long)len, (long long)id2);
            hdrs.emplace_back(hdr, (size_t)hl);
            jobs.push_back(Job{i, at + hl, len, L, f});
            at += hl + len + (len - 1) / 100 + 1;          // bases, the line breaks between them, the final one
            words += nw;
            id2++;
        }
        if (out && at <= cap && words >= (1 << 16)) {
            // base j0.. of a contig: n characters from p (a line break before every hundredth base but the first)
            auto put = [&](const Job &jb, int64_t j0, const char *p, int n) {
                while (n > 0) {
                    if (j0 > 0 && j0 % 100 == 0) out[jb.body + j0 + j0 / 100 - 1] = '\n';
                    const int room = (int)std::min<int64_t>(n, 100 - j0 % 100);
                    memcpy(out + jb.body + j0 + j0 / 100, p, (size_t)room);
                    p += room; j0 += room; n -= room;
                }
            };
            const int T = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
            const int64_t per = (words + T - 1) / T;
            auto work = [&](int t) {
                int64_t w_lo = (int64_t)t * per, w_hi = std::min(words, w_lo + per), base = 0;
                for (size_t q = 0; q < jobs.size(); q++) {
                    const Job &jb = jobs[q];
                    const uint64_t *w = r->ext + r->ext_off[jb.i];
                    const int64_t nw = r->ext_off[jb.i + 1] - r->ext_off[jb.i];
                    const int64_t a = std::max<int64_t>(w_lo - base, 0), b2 = std::min<int64_t>(w_hi - base, nw);
                    base += nw;
                    if (a >= b2) continue;
                    const int64_t eoff = r->marker[jb.i] == 1 ? sub : 0, koff = r->marker[jb.i] == 1 ? 0 : jb.L;
                    char tmp[32];
                    for (int64_t x = a; x < b2; x++) {
                        if (x == 0) {
                            // the record's thread of word 0 also writes the header, the key and the closing line break
                            memcpy(out + jb.body - (int64_t)hdrs[q].size(), hdrs[q].data(), hdrs[q].size());
                            const uint64_t *kp = r->key + (size_t)jb.i * kw;
                            int64_t o2 = 0;
                            for (int w2 = 0; w2 < kw; w2++) {
                                const int nb = w2 < kw - 1 ? 31 : sub - 31 * (kw - 1);
                                for (int j = 0; j < nb; j++) tmp[j] = NUC[(kp[w2] >> (2 * (nb - 1 - j))) & 3];
                                put(jb, koff + o2, tmp, nb);
                                o2 += nb;
                            }
                            for (int j = 0; j < jb.f; j++) tmp[j] = NUC[(w[0] >> (2 * (jb.f - 1 - j))) & 3];
                            put(jb, eoff, tmp, jb.f);
                            out[jb.body + jb.len + (jb.len - 1) / 100] = '\n';
                        } else {
                            const uint64_t v = w[x] << 2;
                            for (int j = 0; j < 7; j++) memcpy(tmp + 4 * j, QUAD[(v >> (56 - 8 * j)) & 255], 4);
                            memcpy(tmp + 28, QUAD[v & 255], 3);
                            put(jb, eoff + jb.f + (x - 1) * 31, tmp, 31);
                        }
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) {
                try { th.emplace_back(work, t); }
                catch (const std::system_error &) { work(t); }       // (no thread to be had: this one does the share)
            }
            work(0);
            for (auto &x : th) x.join();
            if (n_contigs) *n_contigs = id2;
            return at;
        }
    }
    int64_t pos = 0, idx = 0;
    std::vector<char> b;
    auto putc_ = [&](char c) { if (pos < cap) out[pos] = c; pos++; };
    auto puts_ = [&](const char *p, int64_t n) {                           // bulk copy when it fits, else count / clip
        if (pos + n <= cap) memcpy(out + pos, p, (size_t)n);
        else for (int64_t j = 0; j < n; j++) if (pos + j < cap) out[pos + j] = p[j];
        pos += n;
    };
    for (int64_t i = 0; i < r->n; i++) {
        // DS drops records whose markers are both <= -10,000,000  P/ReflexivDSMain.java:749
        if (twin == RFX_TWIN_DS && r->left[i] <= -10000000 && r->right[i] <= -10000000) continue;
        const uint64_t *w = r->ext + r->ext_off[i];
        const int64_t nw = r->ext_off[i + 1] - r->ext_off[i];
        const int nlz = w[0] ? __builtin_clzll(w[0]) : 64;
        const int f = 32 - (nlz / 2 + 1);                                  // :702
        const int64_t L = (nw - 1) * 31 + f;
        const int64_t len = L + sub;
        if (len < min_contig) continue;                                    // :596, :606
        b.resize((size_t)len);
        char *e = r->marker[i] == 1 ? b.data() + sub : b.data();           // :593-594 / :603-604
        char *kb = r->marker[i] == 1 ? b.data() : b.data() + L;
        {                                                                                          // :704-709 / 64 :1928-1940
            const uint64_t *kp = r->key + (size_t)i * kw;
            int o2 = 0;
            for (int w2 = 0; w2 < kw; w2++) {
                const int nb = w2 < kw - 1 ? 31 : sub - 31 * (kw - 1);
                for (int j = 0; j < nb; j++) kb[o2++] = NUC[(kp[w2] >> (2 * (nb - 1 - j))) & 3];
            }
        }
        int64_t o = 0;
        for (int j = 0; j < f; j++) e[o++] = NUC[(w[0] >> (2 * (f - 1 - j))) & 3];                 // :713-718
        for (int64_t x = 1; x < nw; x++) {                                                         // :720-729
            // 31 bases per word, four at a time through a byte table (a 2.6 Mbp contig is 84 K words)
            const uint64_t v = w[x] << 2;                                  // first base in the top pair
            char *q = e + o;
            for (int j = 0; j < 7; j++) memcpy(q + 4 * j, QUAD[(v >> (56 - 8 * j)) & 255], 4);
            memcpy(q + 28, QUAD[v & 255], 3);
            o += 31;
        }
        char hdr[96];
        int hl = twin == RFX_TWIN_DS
                     ? snprintf(hdr, sizeof hdr, ">Contig-%lld-(%d,%d)-%lld\n", (long long)len, r->left[i],
                                r->right[i], (long long)idx)                // DS :755, :722
                     : snprintf(hdr, sizeof hdr, ">Contig-%lld-%lld\n", (long long)len, (long long)idx);   // :597, :578
        puts_(hdr, hl);
        for (int64_t j = 0; j < len; j += 100) {                            // changeLine :616-637
            if (j > 0) putc_('\n');
            puts_(b.data() + j, std::min<int64_t>(100, len - j));
        }
        putc_('\n');                                                        // saveAsTextFile
        idx++;
    }
    if (n_contigs) *n_contigs = idx;
    return pos;
}

struct HostRecords {
    std::vector<uint64_t> key, ext; std::vector<int32_t> marker, left, right; std::vector<int64_t> ext_off;
    rfx_records view;
    void resize(int64_t n, int64_t words) {
        key.resize((size_t)std::max<int64_t>(n, 1)); marker.resize(key.size()); left.resize(key.size());
        right.resize(key.size()); ext_off.resize((size_t)n + 1); ext.resize((size_t)std::max<int64_t>(words, 1));
        view = rfx_records{};
        view.n = 0; view.key = key.data(); view.marker = marker.data(); view.ext_off = ext_off.data();
        view.ext = ext.data(); view.left = left.data(); view.right = right.data(); view.cap_n = n;
        view.cap_words = words;
    }
};

}  // namespace

// a piece of a workspace slot (not owned)
struct DevSpan { void *p = nullptr; template <class T> T *as() const { return (T *)p; } };

// What a C++ exception under an entry point becomes (RFX_API_CATCH, rfx_internal.h).
int rfx_api_exception(rfx_ctx *ctx, const char *where) noexcept {
    const char *what = "unknown C++ exception";
    char buf[400];
    try { throw; }
    catch (const std::bad_alloc &) { what = "std::bad_alloc (out of host memory)"; }
    catch (const std::exception &e) { snprintf(buf, sizeof buf, "%s", e.what()); what = buf; }
    catch (...) {}
    if (ctx) {
        try { ctx->last_error = std::string(where) + ": C++ exception caught at the C ABI: " + what; } catch (...) {}
    }
    return RFX_E_HOST;
}

extern "C" {

int rfx_version(void) { return 100; }

void rfx_default_params(rfx_params *p) try {
    p->k = 31; p->min_cov = 2; p->max_cov = 10000000; p->min_error_cov = 8; p->min_contig = 500;
    p->min_iter = 15; p->max_iter = 150; p->front_clip = 0; p->end_clip = 0; p->partitions = 8;
    p->twin = RFX_TWIN_DS; p->coalesce = 0; p->extras = 1;
} RFX_API_CATCH_VOID(nullptr)

// RFX_BACKTRACE=1 (debugging aid): the native frames of a host crash inside the library on stderr -- offsets into the
// shared objects, for llvm-symbolizer -- before the signal takes its course
static void rfx_crash_backtrace(int sig) {
    void *fr[64];
    const int n = backtrace(fr, 64);
    const char msg[] = "\n[reflexiv] fatal signal, native frames:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(fr, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int rfx_ctx_create(int device, rfx_ctx **out) try {
    if (!out) return RFX_E_ARG;
    *out = nullptr;
    if (const char *e = getenv("RFX_BACKTRACE")) {
        if (atoi(e)) { signal(SIGSEGV, rfx_crash_backtrace); signal(SIGABRT, rfx_crash_backtrace); signal(SIGBUS, rfx_crash_backtrace); }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RFX_E_NOGPU;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return RFX_E_NOGPU; }
    if (device >= ndev) return RFX_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return RFX_E_NOGPU;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RFX_E_NOGPU;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RFX_E_NOGPU;   // the code object is gfx950-only
    rfx_ctx *ctx = new rfx_ctx();
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return RFX_E_HIP; }
    ctx->own_stream = true;
    // keep freed scratch in the stream-ordered pool instead of returning it to the driver
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
        uint64_t thr = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    }
    // result staging, part of the context like the workspaces; one device-to-host copy through it now,
    // so that the runtime sets up its copy path here and not inside the first timed download (measured:
    // the first multi-megabyte hipMemcpyAsync D2H of a process blocks ~7-18 ms on the host)
    if (void *pin = ctx->pinned_get((size_t)16 << 20)) {
        void *tmp = nullptr;
        if (hipMalloc(&tmp, (size_t)4 << 20) == hipSuccess) {
            (void)hipMemsetAsync(tmp, 0, (size_t)4 << 20, ctx->stream);
            (void)hipMemcpyAsync(pin, tmp, (size_t)4 << 20, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(tmp);
        }
    }
    *out = ctx;
    return RFX_OK;
} RFX_API_CATCH(nullptr)

void rfx_ctx_destroy(rfx_ctx *ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->ws_free();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
} RFX_API_CATCH_VOID(ctx)

int rfx_ctx_sync(rfx_ctx *ctx) try {
    if (!ctx) return RFX_E_ARG;
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

int rfx_ctx_set_stream(rfx_ctx *ctx, void *hip_stream) try {
    if (!ctx) return RFX_E_ARG;
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return RFX_OK;
} RFX_API_CATCH(ctx)

// Hands the context's grow-only workspaces (count-stage record buffers, extend-stage arenas, the staging and packed reads
// of rfx_assemble_reads, pinned staging) back to the driver; the next call allocates what it needs again.
int rfx_ctx_trim(rfx_ctx *ctx) try {
    if (!ctx) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    RFX_TRY(sync_checked(ctx));
    for (auto &w : ctx->ws) { if (w.p) (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
    return RFX_OK;
} RFX_API_CATCH(ctx)

int64_t rfx_ctx_workspace_bytes(rfx_ctx *ctx) try {
    int64_t t = 0;
    if (ctx) for (auto &w : ctx->ws) t += (int64_t)w.bytes;
    return t;
} RFX_API_CATCH(ctx)

void *rfx_ctx_stream(rfx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
const char *rfx_last_error(rfx_ctx *ctx) {
    if (!ctx) return "no context";
    if (ctx->foreign_hip_error.empty()) return ctx->last_error.c_str();
    try { ctx->last_error_text = ctx->last_error + " [after RCCL: " + ctx->foreign_hip_error + "]"; } catch (...) { return ctx->last_error.c_str(); }
    return ctx->last_error_text.c_str();
}

// ------------------------------------------------------------- host operators

int rfx_extract_canon(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off, int64_t n_reads, int k,
                      int front_clip, int end_clip, uint64_t *out_kmers, int64_t cap, int64_t *out_n) try {
    if (!ctx || !read_off || !out_n || n_reads < 0) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    if (front_clip < 0 || end_clip < 0) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    *out_n = 0;
    if (n_reads == 0) return RFX_OK;
    const int64_t nb = read_off[n_reads] - read_off[0];
    int64_t maxlen = 0;
    for (int64_t r = 0; r < n_reads; r++) maxlen = std::max(maxlen, read_off[r + 1] - read_off[r]);
    const int wpr = (int)std::max<int64_t>(1, (maxlen + 31) / 32);
    DevBuf d_bases, d_off, d_words, d_nk, d_koff, d_out;
    RFX_HIP(d_bases.alloc((size_t)nb, ctx->stream));
    RFX_HIP(d_off.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
    RFX_HIP(d_words.alloc((size_t)n_reads * wpr * 8, ctx->stream));
    RFX_HIP(d_nk.alloc((size_t)n_reads * 8, ctx->stream));
    RFX_HIP(d_koff.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
    // offsets are rebased to the first read
    std::vector<int64_t> off((size_t)n_reads + 1);
    for (int64_t r = 0; r <= n_reads; r++) off[(size_t)r] = read_off[r] - read_off[0];
    if (nb > 0) RFX_HIP(hipMemcpyAsync(d_bases.p, bases + read_off[0], (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(encode_reads(ctx, d_bases.as<uint8_t>(), d_off.as<int64_t>(), n_reads, wpr, d_words.as<uint64_t>(), nullptr));
    RFX_TRY(kmer_counts_per_read(ctx, d_off.as<int64_t>(), n_reads, k, front_clip, end_clip, d_nk.as<uint64_t>()));
    RFX_TRY(exclusive_scan_u64(ctx, d_nk.as<uint64_t>(), d_koff.as<uint64_t>(), n_reads));
    uint64_t total = 0;
    RFX_HIP(hipMemcpyAsync(&total, d_koff.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    *out_n = (int64_t)total;
    if ((int64_t)total > cap) return RFX_E_CAP;
    if (total == 0) return RFX_OK;
    if (!out_kmers) return RFX_E_ARG;
    RFX_HIP(d_out.alloc((size_t)total * 8, ctx->stream));
    RFX_TRY(extract_ordered_packed(ctx, d_words.as<uint64_t>(), wpr, d_koff.as<uint64_t>(), n_reads, k, front_clip,
                                   d_out.as<uint64_t>()));
    RFX_HIP(hipMemcpyAsync(out_kmers, d_out.p, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

int rfx_extract_canon_w(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off, int64_t n_reads, int k,
                        int front_clip, int end_clip, uint64_t *out_kmers, int64_t cap, int64_t *out_n) try {
    if (!ctx || !read_off || !out_n || n_reads < 0) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (front_clip < 0 || end_clip < 0) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    *out_n = 0;
    if (n_reads == 0) return RFX_OK;
    const int W = k / 32 + 1;
    const int64_t nb = read_off[n_reads] - read_off[0];
    int64_t maxlen = 0;
    for (int64_t r = 0; r < n_reads; r++) maxlen = std::max(maxlen, read_off[r + 1] - read_off[r]);
    const int wpr = (int)std::max<int64_t>(1, (maxlen + 31) / 32);
    DevBuf d_bases, d_off, d_words, d_nk, d_koff, d_soa, d_out;
    RFX_HIP(d_bases.alloc((size_t)nb, ctx->stream));
    RFX_HIP(d_off.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
    RFX_HIP(d_words.alloc((size_t)n_reads * wpr * 8, ctx->stream));
    RFX_HIP(d_nk.alloc((size_t)n_reads * 8, ctx->stream));
    RFX_HIP(d_koff.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
    std::vector<int64_t> off((size_t)n_reads + 1);
    for (int64_t r = 0; r <= n_reads; r++) off[(size_t)r] = read_off[r] - read_off[0];
    if (nb > 0) RFX_HIP(hipMemcpyAsync(d_bases.p, bases + read_off[0], (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(encode_reads(ctx, d_bases.as<uint8_t>(), d_off.as<int64_t>(), n_reads, wpr, d_words.as<uint64_t>(), nullptr));
    RFX_TRY(kmer_counts_per_read_w(ctx, d_off.as<int64_t>(), n_reads, k, front_clip, end_clip, d_nk.as<uint64_t>()));
    RFX_TRY(exclusive_scan_u64(ctx, d_nk.as<uint64_t>(), d_koff.as<uint64_t>(), n_reads));
    uint64_t total = 0;
    RFX_HIP(hipMemcpyAsync(&total, d_koff.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    *out_n = (int64_t)total;
    if ((int64_t)total > cap) return RFX_E_CAP;
    if (total == 0) return RFX_OK;
    if (!out_kmers) return RFX_E_ARG;
    RFX_HIP(d_soa.alloc((size_t)total * W * 8, ctx->stream));
    RFX_HIP(d_out.alloc((size_t)total * W * 8, ctx->stream));
    RFX_TRY(extract_w(ctx, d_words.as<uint64_t>(), wpr, d_koff.as<uint64_t>(), 0, n_reads, k, front_clip,
                      d_soa.as<uint64_t>(), (int64_t)total));
    RFX_TRY(soa_to_aos(ctx, d_soa.as<uint64_t>(), (int64_t)total, W, d_out.as<uint64_t>()));
    RFX_HIP(hipMemcpyAsync(out_kmers, d_out.p, (size_t)total * W * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

int rfx_count_filter_w(rfx_ctx *ctx, const uint64_t *kmers, int64_t n, int k, int min_cov, int max_cov,
                       uint64_t *out_keys, int64_t *out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) try {
    if (!ctx || n < 0 || !out_n || (n > 0 && !kmers)) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    RFX_HIP(hipSetDevice(ctx->device));
    *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n == 0) return RFX_OK;
    const int W = k / 32 + 1;
    DevBuf d_in, d_soa, d_keys, d_counts;
    RFX_HIP(d_in.alloc((size_t)n * W * 8, ctx->stream));
    RFX_HIP(d_soa.alloc((size_t)n * W * 8, ctx->stream));
    RFX_HIP(d_keys.alloc((size_t)n * W * 8, ctx->stream));
    RFX_HIP(d_counts.alloc((size_t)n * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_in.p, kmers, (size_t)n * W * 8, hipMemcpyHostToDevice, ctx->stream));
    int64_t m = 0, dist = 0;
    if (wide_fast_path(k)) {
        RFX_TRY(count_filter_w2(ctx, d_in.as<uint64_t>(), n, k, min_cov, max_cov, d_keys.as<uint64_t>(), d_counts.as<int64_t>(),
                                n, &m, &dist));
    } else {
        RFX_TRY(aos_to_soa(ctx, d_in.as<uint64_t>(), n, W, d_soa.as<uint64_t>()));
        RFX_TRY(count_filter_w(ctx, d_soa.as<uint64_t>(), n, k, min_cov, max_cov, d_keys.as<uint64_t>(), d_counts.as<int64_t>(),
                               n, &m, &dist));
    }
    *out_n = m;
    if (out_distinct) *out_distinct = dist;
    if (m > cap) return RFX_E_CAP;
    if (m > 0) {
        if (!out_keys || !out_counts) return RFX_E_ARG;
        RFX_HIP(hipMemcpyAsync(out_keys, d_keys.p, (size_t)m * W * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(out_counts, d_counts.p, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

int64_t rfx_kmers_per_read_w(int read_len, int k, int front_clip, int end_clip) try {
    return kmers_per_read_w(read_len, k, front_clip, end_clip);
} RFX_API_CATCH(nullptr)

int rfx_dev_count_reads_w(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read, int read_len,
                          int k, int front_clip, int end_clip, int min_cov, int max_cov, uint64_t *d_out_keys,
                          int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct,
                          int64_t *out_instances) try {
    if (!ctx || !out_n || n_reads < 0 || (n_reads > 0 && !d_words)) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (front_clip < 0 || end_clip < 0 || words_per_read * 32 < read_len) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    const int W = k / 32 + 1;
    ctx->timing.clear();
    const int64_t nk = kmers_per_read_w(read_len, k, front_clip, end_clip);
    const int64_t N = nk * n_reads;
    if (out_instances) *out_instances = N;
    if (N == 0) return RFX_OK;
    if (wide_fast_path(k) && !getenv("RFX_WIDE_MATERIALIZE")) {
        // level 1 straight from the packed reads: the 16-byte elements are never written unpartitioned
        int64_t m = 0;
        int st = count_wide2_reads(ctx, d_words, n_reads, words_per_read, nk, k, front_clip, min_cov, max_cov, d_out_keys,
                                   d_out_counts, cap, &m, out_distinct);
        *out_n = m;
        if (st == RFX_OK) st = order_wide2(ctx, d_out_keys, d_out_counts, m, k);
        RFX_TRY(sync_checked(ctx));
        ScopedTimer::collect(ctx);
        return st;
    }
    DevBuf d_soa;
    RFX_HIP(d_soa.alloc((size_t)N * W * 8, ctx->stream));
    const bool fast = wide_fast_path(k);
    {
        ScopedTimer t(ctx, "extract_w");
        RFX_TRY(extract_w(ctx, d_words, words_per_read, nullptr, nk, n_reads, k, front_clip, d_soa.as<uint64_t>(), N,
                          fast ? 1 : 0));
    }
    int st;
    if (fast) {
        st = count_filter_w2(ctx, d_soa.as<uint64_t>(), N, k, min_cov, max_cov, d_out_keys, d_out_counts, cap, out_n,
                             out_distinct);
    } else {
        ScopedTimer t(ctx, "count_w");
        st = count_filter_w(ctx, d_soa.as<uint64_t>(), N, k, min_cov, max_cov, d_out_keys, d_out_counts, cap, out_n,
                            out_distinct);
    }
    RFX_TRY(sync_checked(ctx));
    ScopedTimer::collect(ctx);
    return st;
} RFX_API_CATCH(ctx)

int rfx_count_filter(rfx_ctx *ctx, const uint64_t *kmers, int64_t n, int min_cov, int max_cov, int twin,
                     uint64_t *out_keys, int32_t *out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) try {
    if (!ctx || n < 0 || !out_n || (n > 0 && !kmers)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n == 0) return RFX_OK;
    DevBuf d_in, d_keys, d_counts;
    RFX_HIP(d_in.alloc((size_t)n * 8, ctx->stream));
    // at most n survivors; the device path reports the needed size if cap is smaller
    const int64_t dcap = n;
    RFX_HIP(d_keys.alloc((size_t)dcap * 8, ctx->stream));
    RFX_HIP(d_counts.alloc((size_t)dcap * 4, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_in.p, kmers, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    int64_t m = 0, dist = 0;
    RFX_TRY(count_filter(ctx, nullptr, d_in.as<uint64_t>(), n, min_cov, max_cov, twin, nullptr, 0,
                         d_keys.as<uint64_t>(), d_counts.as<int32_t>(), dcap, &m, &dist));
    *out_n = m;
    if (out_distinct) *out_distinct = dist;
    if (m > cap) return RFX_E_CAP;
    if (m > 0) {
        RFX_HIP(hipMemcpyAsync(out_keys, d_keys.p, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(out_counts, d_counts.p, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

int rfx_rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *kmers, const int32_t *counts, int64_t n, int k,
                          rfx_records *out) try {
    if (!ctx || n < 0 || !out) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevBuf dk, dc;
    const int aw = asm_words(k);
    RFX_HIP(dk.alloc((size_t)n * 8 * aw, ctx->stream));
    RFX_HIP(dc.alloc((size_t)n * 4, ctx->stream));
    if (n > 0) {
        RFX_HIP(hipMemcpyAsync(dk.p, kmers, (size_t)n * 8 * aw, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(dc.p, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    DevRecords r;
    RFX_TRY(rc_expand_subkmer(ctx, dk.as<uint64_t>(), dc.as<int32_t>(), n, k, r));
    return download_to(ctx, r, out);
} RFX_API_CATCH(ctx)

int rfx_sort_records(rfx_ctx *ctx, const rfx_records *in, int P, rfx_records *out, int64_t *part_start) try {
    if (!ctx || !in || !out || P < 1) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, s;
    DevBuf ps;
    RFX_TRY(dev_records_upload(ctx, in, d));
    if (d.kw > MAX_KEY_WORDS) return RFX_E_ARG;
    RFX_TRY(sort_records(ctx, d, P, 64, s, ps, 0));
    RFX_TRY(download_to(ctx, s, out));
    return download_part_start(ctx, ps, P, part_start);
} RFX_API_CATCH(ctx)

static int fork_host(rfx_ctx *ctx, bool reflected, const rfx_records *in, const int64_t *part_start, int P, int k,
                     int min_error_cov, int twin, rfx_records *out, int64_t *out_part_start) {
    if (!ctx || !in || !out || !part_start || P < 1) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, o;
    DevBuf ps, ops;
    RFX_TRY(dev_records_upload(ctx, in, d));
    RFX_TRY(upload_part_start(ctx, part_start, P, ps));
    RFX_TRY(fork_filter(ctx, reflected, d, ps.as<int64_t>(), P, k, min_error_cov, twin, o, ops));
    RFX_TRY(download_to(ctx, o, out));
    return download_part_start(ctx, ops, P, out_part_start);
}

int rfx_fork_filter_forward(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k,
                            int min_error_cov, int twin, rfx_records *out, int64_t *out_part_start) try {
    return fork_host(ctx, false, in, part_start, P, k, min_error_cov, twin, out, out_part_start);
} RFX_API_CATCH(ctx)

int rfx_fork_filter_reflected(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k,
                              int min_error_cov, int twin, rfx_records *out, int64_t *out_part_start) try {
    return fork_host(ctx, true, in, part_start, P, k, min_error_cov, twin, out, out_part_start);
} RFX_API_CATCH(ctx)

int rfx_reflect_from_forward(rfx_ctx *ctx, const rfx_records *in, int k, rfx_records *out) try {
    if (!ctx || !in || !out) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, o;
    RFX_TRY(dev_records_upload(ctx, in, d));
    RFX_TRY(reflect_from_forward(ctx, d, k, o));
    return download_to(ctx, o, out);
} RFX_API_CATCH(ctx)

int rfx_random_reflection(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k,
                          rfx_records *out) try {
    if (!ctx || !in || !out || !part_start || P < 1) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, o;
    DevBuf ps;
    RFX_TRY(dev_records_upload(ctx, in, d));
    RFX_TRY(upload_part_start(ctx, part_start, P, ps));
    RFX_TRY(random_reflection(ctx, d, ps.as<int64_t>(), P, k, o));
    return download_to(ctx, o, out);
} RFX_API_CATCH(ctx)

static int extend_host(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k, int twin,
                       int stage, int start_marker, rfx_records *out, int64_t *out_part_start) {
    if (!ctx || !in || !out || !part_start || P < 1 || stage < 0 || stage > 2) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, o;
    DevBuf ps, ops;
    RFX_TRY(dev_records_upload(ctx, in, d));
    RFX_TRY(upload_part_start(ctx, part_start, P, ps));
    RFX_TRY(extend_pass(ctx, d, ps.as<int64_t>(), P, k, twin, stage, o, ops, start_marker));
    RFX_TRY(download_to(ctx, o, out));
    return download_part_start(ctx, ops, P, out_part_start);
}

int rfx_extend_pass(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k, int twin,
                    int stage, rfx_records *out, int64_t *out_part_start) try {
    return extend_host(ctx, in, part_start, P, k, k > 31 ? RFX_TWIN_DS : twin, stage, 2, out, out_part_start);
} RFX_API_CATCH(ctx)

int rfx_extend_pass_w(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k, int stage,
                      int scramble, rfx_records *out, int64_t *out_part_start) try {
    if (scramble != 2 && scramble != 3) return RFX_E_ARG;
    return extend_host(ctx, in, part_start, P, k, RFX_TWIN_DS, stage, scramble == 3 ? 1 : 2, out, out_part_start);
} RFX_API_CATCH(ctx)

int rfx_extras_operator(rfx_ctx *ctx, int op, const rfx_records *in, const int64_t *part_start, int P, int k,
                        rfx_records *out, int64_t *out_part_start) try {
    if (!ctx || !in || !out || !part_start || P < 1 || op < 0 || op > 6) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords d, o;
    DevBuf ps, ops;
    RFX_TRY(dev_records_upload(ctx, in, d));
    RFX_TRY(upload_part_start(ctx, part_start, P, ps));
    RFX_TRY(extras_operator(ctx, op, d, ps.as<int64_t>(), P, k, o, ops));
    RFX_TRY(download_to(ctx, o, out));
    return download_part_start(ctx, ops, P, out_part_start);
} RFX_API_CATCH(ctx)

int rfx_contigs_text(rfx_ctx *ctx, const rfx_records *in, int k, int min_contig, int twin, char *out, int64_t cap,
                     int64_t *out_len, int64_t *out_contigs) try {
    if (!ctx || !in || !out_len) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(in, k));
    int64_t len = contigs_text_host(in, k, min_contig, twin, out, out ? cap : 0, out_contigs);
    *out_len = len;
    return len > cap ? RFX_E_CAP : RFX_OK;
} RFX_API_CATCH(ctx)

// ------------------------------------------------------------ record operators on device-resident sets

namespace {

void wrap_dev(rfx_ctx *ctx, const rfx_records *d, DevRecords &r) {
    r.kw = d->key_words > 1 ? d->key_words : 1;
    r.n = d->n; r.words = d->need_words;
    DevBuf *bufs[6] = {&r.key, &r.marker, &r.ext_off, &r.ext, &r.left, &r.right};
    void *ptrs[6] = {d->key, d->marker, d->ext_off, d->ext, d->left, d->right};
    for (int i = 0; i < 6; i++) { bufs[i]->release(); bufs[i]->p = ptrs[i]; bufs[i]->borrowed = true; bufs[i]->s = ctx->stream; }
}

int copy_out_dev(rfx_ctx *ctx, const DevRecords &o, rfx_records *d) {
    d->need_n = o.n; d->need_words = o.words; d->key_words = o.kw;
    if (o.n > d->cap_n || o.words > d->cap_words) return RFX_E_CAP;
    const int64_t n = o.n;
    if (n > 0) {
        RFX_HIP(hipMemcpyAsync(d->key, o.key.p, (size_t)n * 8 * o.kw, hipMemcpyDeviceToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d->marker, o.marker.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d->left, o.left.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d->right, o.right.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (o.words > 0) RFX_HIP(hipMemcpyAsync(d->ext, o.ext.p, (size_t)o.words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    RFX_HIP(hipMemcpyAsync(d->ext_off, o.ext_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    d->n = n;
    return RFX_OK;
}

int copy_ps_dev(rfx_ctx *ctx, const DevBuf &ps, int P, int64_t *d_dst) {
    if (!d_dst) return RFX_OK;
    RFX_HIP(hipMemcpyAsync(d_dst, ps.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

__global__ void k_lower_bound(const uint64_t *__restrict__ keys, int64_t n, const uint64_t *__restrict__ values, int64_t m,
                              int upper, int64_t *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t v = values[j];
    int64_t lo = 0, hi = n;                            // first index with key >= v (upper: > v)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const uint64_t kk = keys[mid];
        if (upper ? kk <= v : kk < v) lo = mid + 1; else hi = mid;
    }
    out[j] = lo;
}

}  // namespace

extern "C" {

int rfx_dev_rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n, int k, rfx_records *d_out) try {
    if (!ctx || n < 0 || !d_out) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords o;
    RFX_TRY(rc_expand_subkmer(ctx, d_kmers, d_counts, n, k, o));
    return copy_out_dev(ctx, o, d_out);
} RFX_API_CATCH(ctx)

int rfx_dev_sort_records(rfx_ctx *ctx, const rfx_records *d_in, int P, int k, rfx_records *d_out, int64_t *d_part_start) try {
    if (!ctx || !d_in || !d_out || P < 1) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(d_in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords in, o;
    DevBuf ps;
    wrap_dev(ctx, d_in, in);
    RFX_TRY(sort_records(ctx, in, P, 2 * (k - 1), o, ps, k));
    RFX_TRY(copy_out_dev(ctx, o, d_out));
    return copy_ps_dev(ctx, ps, P, d_part_start);
} RFX_API_CATCH(ctx)

int rfx_dev_fork_filter(rfx_ctx *ctx, int reflected, const rfx_records *d_in, const int64_t *d_part_start, int P, int k,
                        int min_error_cov, int twin, rfx_records *d_out, int64_t *d_out_part_start) try {
    if (!ctx || !d_in || !d_out || !d_part_start || P < 1) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(d_in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords in, o;
    DevBuf ops;
    wrap_dev(ctx, d_in, in);
    RFX_TRY(fork_filter(ctx, reflected != 0, in, d_part_start, P, k, min_error_cov, twin, o, ops));
    RFX_TRY(copy_out_dev(ctx, o, d_out));
    return copy_ps_dev(ctx, ops, P, d_out_part_start);
} RFX_API_CATCH(ctx)

int rfx_dev_reflect_from_forward(rfx_ctx *ctx, const rfx_records *d_in, int k, rfx_records *d_out) try {
    if (!ctx || !d_in || !d_out) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(d_in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords in, o;
    wrap_dev(ctx, d_in, in);
    RFX_TRY(reflect_from_forward(ctx, in, k, o));
    return copy_out_dev(ctx, o, d_out);
} RFX_API_CATCH(ctx)

int rfx_dev_random_reflection(rfx_ctx *ctx, const rfx_records *d_in, const int64_t *d_part_start, int P, int k, rfx_records *d_out) try {
    if (!ctx || !d_in || !d_out || !d_part_start || P < 1) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(d_in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords in, o;
    wrap_dev(ctx, d_in, in);
    RFX_TRY(random_reflection(ctx, in, d_part_start, P, k, o));
    return copy_out_dev(ctx, o, d_out);
} RFX_API_CATCH(ctx)

int rfx_dev_extend_pass(rfx_ctx *ctx, const rfx_records *d_in, const int64_t *d_part_start, int P, int k, int twin, int stage,
                        int scramble, rfx_records *d_out, int64_t *d_out_part_start) try {
    if (!ctx || !d_in || !d_out || !d_part_start || P < 1 || stage < 0 || stage > 2 || (scramble != 2 && scramble != 3)) return RFX_E_ARG;
    RFX_TRY(check_k_rec(k));
    RFX_TRY(check_kw(d_in, k));
    RFX_HIP(hipSetDevice(ctx->device));
    DevRecords in, o;
    DevBuf ops;
    wrap_dev(ctx, d_in, in);
    RFX_TRY(extend_pass(ctx, in, d_part_start, P, k, k > 31 ? RFX_TWIN_DS : twin, stage, o, ops, scramble == 3 ? 1 : 2));
    RFX_TRY(copy_out_dev(ctx, o, d_out));
    return copy_ps_dev(ctx, ops, P, d_out_part_start);
} RFX_API_CATCH(ctx)

int rfx_dev_lower_bound(rfx_ctx *ctx, const uint64_t *d_sorted_keys, int64_t n, const uint64_t *d_values, int64_t m, int upper,
                        int64_t *d_out) try {
    if (!ctx || n < 0 || m < 0 || (m > 0 && (!d_values || !d_out)) || (n > 0 && !d_sorted_keys)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    if (m == 0) return RFX_OK;
    hipLaunchKernelGGL(k_lower_bound, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx->stream, d_sorted_keys, n, d_values, m, upper, d_out);
    RFX_HIP(hipGetLastError());
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
} RFX_API_CATCH(ctx)

}  // extern "C"

// ------------------------------------------------------------ device pipeline

int rfx_dev_encode_reads(rfx_ctx *ctx, const uint8_t *d_bases, const int64_t *d_read_off, int64_t n_reads,
                         int words_per_read, uint64_t *d_words, uint32_t *d_read_len) try {
    if (!ctx || n_reads < 0 || words_per_read < 1) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    return encode_reads(ctx, d_bases, d_read_off, n_reads, words_per_read, d_words, d_read_len);
} RFX_API_CATCH(ctx)

int64_t rfx_kmers_per_read(int read_len, int k, int front_clip, int end_clip) try {
    return kmers_per_read(read_len, k, front_clip, end_clip);
} RFX_API_CATCH(nullptr)

int64_t rfx_count_workspace_bytes(int64_t n_kmers) { return count_workspace_bytes(n_kmers); }

int rfx_dev_count_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read, int read_len,
                        int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                        void *d_workspace, int64_t workspace_bytes, uint64_t *d_out_keys, int32_t *d_out_counts,
                        int64_t cap, int64_t *out_n, int64_t *out_distinct, int64_t *out_instances) try {
    if (!ctx || !d_words || n_reads < 0 || words_per_read * 32 < read_len) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    ReadStore rs{d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip};
    if (out_instances) *out_instances = kmers_per_read(read_len, k, front_clip, end_clip) * n_reads;
    return count_filter(ctx, &rs, nullptr, 0, min_cov, max_cov, twin, d_workspace, workspace_bytes, d_out_keys,
                        d_out_counts, cap, out_n, out_distinct);
} RFX_API_CATCH(ctx)

int rfx_dev_count_reads_ragged(rfx_ctx *ctx, const uint64_t *d_words, const uint32_t *d_read_len, int64_t n_reads,
                               int words_per_read, int max_read_len, int k, int front_clip, int end_clip, int min_cov,
                               int max_cov, int twin, uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                               int64_t *out_n, int64_t *out_distinct, int64_t *out_instances) try {
    if (!ctx || !d_words || !d_read_len || n_reads < 0 || words_per_read * 32 < max_read_len) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    ReadStore rs{d_words, n_reads, words_per_read, max_read_len, k, front_clip, end_clip};
    rs.read_len_arr = d_read_len;
    RFX_TRY(ragged_instances(ctx, d_read_len, n_reads, k, front_clip, end_clip, &rs.n_instances));
    if (out_instances) *out_instances = rs.n_instances;
    return count_filter(ctx, &rs, nullptr, 0, min_cov, max_cov, twin, nullptr, 0, d_out_keys, d_out_counts, cap, out_n,
                        out_distinct);
} RFX_API_CATCH(ctx)

int rfx_dev_bucket_wide_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read, int read_len,
                                 int k, int front_clip, int end_clip, int n_owners, void *d_out_elems, int64_t cap_elems,
                                 int64_t *d_owner_off, int64_t *h_owner_off) try {
    if (!ctx || !d_words || !d_owner_off || n_reads < 0 || words_per_read * 32 < read_len) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (!wide_fast_path(k) || front_clip < 0 || end_clip < 0) return RFX_E_ARG;      // two-word k-mers only
    RFX_HIP(hipSetDevice(ctx->device));
    const int64_t nk = kmers_per_read_w(read_len, k, front_clip, end_clip);
    return bucket_wide_by_owner(ctx, d_words, n_reads, words_per_read, nk, k, front_clip, n_owners, d_out_elems, cap_elems,
                                d_owner_off, h_owner_off);
} RFX_API_CATCH(ctx)

int rfx_dev_bucket_wide_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read,
                                         int read_len, int k, int front_clip, int end_clip, int n_owners,
                                         void *d_out_records, int64_t cap_records, int64_t *d_owner_off,
                                         int64_t *h_owner_off, int64_t *out_n_records) try {
    if (!ctx || !d_words || !d_owner_off || n_reads < 0 || words_per_read * 32 < read_len) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (!wide_fast_path(k) || front_clip < 0 || end_clip < 0) return RFX_E_ARG;      // two-word k-mers only
    RFX_HIP(hipSetDevice(ctx->device));
    ctx->timing.clear();
    const int64_t nk = kmers_per_read_w(read_len, k, front_clip, end_clip);
    const int st = bucket_wide_records_by_owner(ctx, d_words, n_reads, words_per_read, nk, k, front_clip, n_owners,
                                                d_out_records, cap_records, d_owner_off, h_owner_off, out_n_records);
    if (h_owner_off) ScopedTimer::collect(ctx);
    return st;
} RFX_API_CATCH(ctx)

int rfx_dev_count_wide_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k,
                               int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap,
                               int64_t *out_n, int64_t *out_distinct) try {
    if (!ctx || !out_n || n_records < 0 || (n_records > 0 && !d_records)) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (!wide_fast_path(k)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    ctx->timing.clear();
    int64_t m = 0;
    int st = count_wide_records(ctx, d_records, n_records, n_instances_hint, k, min_cov, max_cov, d_out_keys, d_out_counts,
                                cap, &m, out_distinct);
    *out_n = m;
    if (st == RFX_OK) st = order_wide2(ctx, d_out_keys, d_out_counts, m, k);
    RFX_TRY(sync_checked(ctx));
    ScopedTimer::collect(ctx);
    return st;
} RFX_API_CATCH(ctx)

int rfx_dev_count_wide_elems(rfx_ctx *ctx, const void *d_elems, int64_t n_elems, int k, int min_cov, int max_cov,
                             uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                             int64_t *out_distinct) try {
    if (!ctx || !out_n || n_elems < 0 || (n_elems > 0 && !d_elems)) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (!wide_fast_path(k)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    ctx->timing.clear();
    const int st = count_filter_w2(ctx, (const uint64_t *)d_elems, n_elems, k, min_cov, max_cov, d_out_keys, d_out_counts,
                                   cap, out_n, out_distinct);
    RFX_TRY(sync_checked(ctx));
    ScopedTimer::collect(ctx);
    return st;
} RFX_API_CATCH(ctx)

int rfx_dev_count_kmers(rfx_ctx *ctx, const uint64_t *d_kmers, int64_t n, int min_cov, int max_cov, int twin,
                        void *d_workspace, int64_t workspace_bytes, uint64_t *d_out_keys, int32_t *d_out_counts,
                        int64_t cap, int64_t *out_n, int64_t *out_distinct) try {
    if (!ctx || n < 0) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    return count_filter(ctx, nullptr, d_kmers, n, min_cov, max_cov, twin, d_workspace, workspace_bytes, d_out_keys,
                        d_out_counts, cap, out_n, out_distinct);
} RFX_API_CATCH(ctx)

int rfx_dev_bucket_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read,
                            int read_len, int k, int front_clip, int end_clip, int n_owners, uint64_t *d_out,
                            int64_t cap, int64_t *d_owner_off, int64_t *h_owner_off) try {
    if (!ctx || !d_words || !d_owner_off) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    ReadStore rs{d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip};
    return bucket_by_owner(ctx, &rs, n_owners, d_out, cap, d_owner_off, h_owner_off);
} RFX_API_CATCH(ctx)

int rfx_dev_bucket_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read,
                                    int read_len, int k, int front_clip, int end_clip, int n_owners,
                                    void *d_out_records, int64_t cap_records, int64_t *d_owner_off,
                                    int64_t *h_owner_off, int64_t *out_n_records) try {
    if (!ctx || !d_words || !d_owner_off) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    ReadStore rs{d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip};
    ctx->timing.clear();
    const int st = bucket_records_by_owner(ctx, &rs, n_owners, d_out_records, cap_records, d_owner_off, h_owner_off,
                                           out_n_records);
    if (h_owner_off) ScopedTimer::collect(ctx);        // (the call has synchronised)
    return st;
} RFX_API_CATCH(ctx)

int rfx_dev_count_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k,
                          int min_cov, int max_cov, int twin, uint64_t *d_out_keys, int32_t *d_out_counts,
                          int64_t cap, int64_t *out_n, int64_t *out_distinct) try {
    if (!ctx || n_records < 0) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    return count_records(ctx, d_records, n_records, n_instances_hint, k, min_cov, max_cov, twin, d_out_keys,
                         d_out_counts, cap, out_n, out_distinct);
} RFX_API_CATCH(ctx)

int rfx_dev_combine_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read, int read_len,
                          int k, int front_clip, int end_clip, int n_owners, void *d_scratch_pairs, void *d_out_pairs,
                          int64_t cap_pairs, int64_t *d_owner_off, int64_t *h_owner_off, int64_t *out_n,
                          int64_t *out_instances) try {
    if (!ctx || !d_words || !d_owner_off || n_reads < 0 || words_per_read * 32 < read_len || cap_pairs < 0) return RFX_E_ARG;
    if (n_owners < 1 || n_owners > 64) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    ReadStore rs{d_words, n_reads, words_per_read, read_len, k, front_clip, end_clip};
    if (out_instances) *out_instances = kmers_per_read(read_len, k, front_clip, end_clip) * n_reads;
    // local count: every distinct k-mer as a {k-mer, count} pair, block-allocated in the scratch buffer
    // (holes have count 0); then one pass groups the pairs by owner and drops the holes
    int64_t extent = 0;
    int st = count_filter(ctx, &rs, nullptr, 0, 1, 0x7fffffff, RFX_TWIN_DS, nullptr, 0, (uint64_t *)d_scratch_pairs, nullptr,
                          cap_pairs, &extent, nullptr, true);
    if (out_n) *out_n = extent;
    if (st != RFX_OK) return st;
    int64_t h_local[65];
    st = bucket_pairs_by_owner(ctx, d_scratch_pairs, extent, n_owners, d_out_pairs, d_owner_off, h_owner_off ? h_owner_off : h_local);
    if (st != RFX_OK) return st;
    if (out_n) *out_n = (h_owner_off ? h_owner_off : h_local)[n_owners];
    return RFX_OK;
} RFX_API_CATCH(ctx)

int rfx_dev_bucket_pairs_by_owner(rfx_ctx *ctx, const void *d_pairs, int64_t n_pairs, int n_owners, void *d_out_pairs,
                                  int64_t *d_owner_off, int64_t *h_owner_off) try {
    if (!ctx || !d_owner_off || n_pairs < 0 || (n_pairs > 0 && (!d_pairs || !d_out_pairs))) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    ctx->timing.clear();
    return bucket_pairs_by_owner(ctx, d_pairs, n_pairs, n_owners, d_out_pairs, d_owner_off, h_owner_off);
} RFX_API_CATCH(ctx)

int rfx_dev_merge_pairs(rfx_ctx *ctx, const void *d_pairs, int64_t n_pairs, int k, int min_cov, int max_cov, int twin,
                        uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n,
                        int64_t *out_distinct) try {
    if (!ctx || n_pairs < 0) return RFX_E_ARG;
    RFX_TRY(check_k(k));
    RFX_HIP(hipSetDevice(ctx->device));
    return merge_pairs(ctx, d_pairs, n_pairs, k, min_cov, max_cov, twin, d_out_keys, d_out_counts, cap, out_n, out_distinct);
} RFX_API_CATCH(ctx)

int rfx_dev_sort_pairs(rfx_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int64_t n, int key_bits,
                       uint64_t *d_tmp_keys, uint32_t *d_tmp_vals) try {
    if (!ctx || n < 0) return RFX_E_ARG;
    if (n > (int64_t)0xFFFFFFFFLL) return RFX_E_LIMIT;
    RFX_HIP(hipSetDevice(ctx->device));
    return sort_pairs(ctx, d_keys, d_vals, n, key_bits, d_tmp_keys, d_tmp_vals);
} RFX_API_CATCH(ctx)

int rfx_dev_synth_genome(rfx_ctx *ctx, uint64_t seed, int64_t genome_len, uint64_t *d_genome) try {
    if (!ctx || !d_genome) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    return synth_genome(ctx, seed, genome_len, d_genome);
} RFX_API_CATCH(ctx)

int rfx_dev_synth_reads(rfx_ctx *ctx, uint64_t seed, const uint64_t *d_genome, int64_t genome_len,
                        int64_t first_read, int64_t n_reads, int read_len, uint32_t err_per_2_32,
                        int words_per_read, uint64_t *d_words) try {
    if (!ctx || !d_genome || !d_words) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    return synth_reads(ctx, seed, d_genome, genome_len, first_read, n_reads, read_len, err_per_2_32,
                       words_per_read, d_words);
} RFX_API_CATCH(ctx)

int rfx_last_count_timing(rfx_ctx *ctx, const char *name, float *ms, int64_t *launches) try {
    if (!ctx || !name) return RFX_E_ARG;
    auto it = ctx->timing.find(name);
    if (it == ctx->timing.end()) { if (ms) *ms = 0.f; if (launches) *launches = 0; return RFX_E_ARG; }
    if (ms) *ms = it->second.ms;
    if (launches) *launches = it->second.launches;
    return RFX_OK;
} RFX_API_CATCH(ctx)

}  // extern "C"  (the driver below is C++: rfx_shard.hip calls it too)

// Driver: P/ReflexivMain.java:168-310 (DS P/ReflexivDSMain.java:221-352); wide = the k > 31 driver
// ReflexivDSMain64.assemblyFromKmer (P/ReflexivDSMain64.java:374-826) without the extras of :584-619 and :672-712
// (SURVEY.md 8f-3): its loop :621-661 iterates all records, checks the count from minimumIteration + 3 on, the
// first repeat of the count flips param.scramble 2 -> 3 (every later pass starts its marker at 1, :7484-7486) and
// only the second stops; the survivors are sorted once more before they become text (:714).
// resume: the loop taken up where the sharded driver (rfx_shard.hip) left it -- its record set gathered on this rank in
// global arrival order, `passes_done` passes behind it, the driver's variables as they stood.  d_keys / d_counts / n unused.
int rfx::assemble_impl(rfx_ctx *ctx, bool wide, const uint64_t *d_keys, const int32_t *d_counts, int64_t n,
                       const rfx_params *prm, char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs,
                       int64_t *trace, int64_t trace_cap, int64_t *n_trace, rfx::AsmResume *resume) {
    if (!ctx || !prm || !out_len || n < 0) return RFX_E_ARG;
    if (wide) { RFX_TRY(check_k_rec(prm->k)); if (prm->k <= 31) return RFX_E_ARG; }
    else RFX_TRY(check_k(prm->k));
    RFX_HIP(hipSetDevice(ctx->device));
    const int k = prm->k, twin = wide ? RFX_TWIN_DS : prm->twin;
    int P = prm->partitions > 0 ? prm->partitions : 1;
    const int key_bits = 2 * (k - 1);
    const int kw = sub_words(k);
    int64_t nt = resume ? resume->nt : 0;
    DevRecords a, b;
    DevBuf ps, ops;
    // two alternating bump arenas hold every temporary and record set of a pass / stage
    Arena arena[2];
    {
        const size_t per = (resume ? (size_t)(resume->recs->n + resume->recs->words) : (size_t)(2 * n)) * (176 + 24 * (size_t)(kw - 1)) + ((size_t)64 << 20);
        for (int i = 0; i < 2; i++) {
            arena[i].base = (char *)ctx->ws_get(2 + i, per);
            if (!arena[i].base) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
            arena[i].cap = per;
        }
    }
    timespec ts_enter; clock_gettime(CLOCK_MONOTONIC, &ts_enter);
    const double t_enter = ts_enter.tv_sec * 1e3 + ts_enter.tv_nsec * 1e-6;
    struct ArenaGuard { ~ArenaGuard() { tl_arena = nullptr; } } arena_guard;
    int arena_turn = 0;
    auto next_arena = [&]() { Arena *ar = &arena[arena_turn++ & 1]; ar->off = 0; tl_arena = ar; };
    next_arena();
    if (!resume) {
    // KmerReverseComplement + ForwardSubKmerExtraction  :168-176
    RFX_TRY(rc_expand_subkmer(ctx, d_keys, d_counts, n, k, a));
    // sortByKey + FilterForkSubKmer[WithErrorCorrection]  :179-186
    next_arena();
    RFX_TRY(sort_records(ctx, a, P, key_bits, b, ps, k));
    next_arena();
    RFX_TRY(fork_filter(ctx, false, b, ps.as<int64_t>(), P, k, prm->min_error_cov, twin, a, ops));
    // ReflectedSubKmerExtractionFromForward  :188-189
    next_arena();
    RFX_TRY(reflect_from_forward(ctx, a, k, b));
    // sortByKey + FilterForkReflectedSubKmer[WithErrorCorrection]  :191-198
    next_arena();
    RFX_TRY(sort_records(ctx, b, P, key_bits, a, ps, k));
    next_arena();
    RFX_TRY(fork_filter(ctx, true, a, ps.as<int64_t>(), P, k, prm->min_error_cov, twin, b, ops));
    // kmerRandomReflection on the filter's output partitions  :204-205
    next_arena();
    RFX_TRY(random_reflection(ctx, b, ops.as<int64_t>(), P, k, a));
    } else {
        // (the gathered set lives in plain stream-ordered allocations, outside the arenas: the first sort reads it from there)
        a.n = resume->recs->n; a.words = resume->recs->words; a.kw = resume->recs->kw;
        a.key = std::move(resume->recs->key); a.marker = std::move(resume->recs->marker); a.ext_off = std::move(resume->recs->ext_off);
        a.ext = std::move(resume->recs->ext); a.left = std::move(resume->recs->left); a.right = std::move(resume->recs->right);
        P = resume->P;
    }

    const bool verbose = getenv("RFX_TRACE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_loop0 = verbose ? ((void)hipStreamSynchronize(ctx->stream), now_ms()) : 0;
    if (verbose) fprintf(stderr, "assemble: stages before the loop %.3f ms\n", t_loop0 - t_enter);
    auto one_pass = [&](int stage, int start_marker = 2) -> int {
        const double t0 = verbose ? now_ms() : 0;
        next_arena();                     // `a` (this pass's input) stays valid in the other arena
        RFX_TRY(sort_records(ctx, a, P, key_bits, b, ps, k));                     // sortByKey :211,:235,:247,:286
        if (verbose) (void)hipStreamSynchronize(ctx->stream);
        const double t1 = verbose ? now_ms() : 0;
        RFX_TRY(extend_pass(ctx, b, ps.as<int64_t>(), P, k, twin, stage, a, ops, start_marker));
        if (verbose) fprintf(stderr, "pass %lld: n_in %lld words %lld sort %.3f ms extend %.3f ms -> n %lld\n", (long long)nt,
                             (long long)b.n, (long long)b.words, t1 - t0, now_ms() - t1, (long long)a.n);
        if (trace && nt < trace_cap) trace[nt] = a.n;
        nt++;
        return RFX_OK;
    };
    int iterations = resume ? resume->iterations : 0;
    // the five passes before the loop: :221-222 (stage 0), :233-241 (three more, iterations 1..3), :247-254 (first-array)
    for (int np = resume ? resume->passes_done : 0; np < 5; np++) {
        if (np >= 1) iterations++;
        RFX_TRY(one_pass(np < 4 ? 0 : 1));
    }
    int partitionNumber = resume ? resume->partition_number : P;
    int64_t contigNumber = resume ? resume->contig_number : 0;
    int scramble = resume ? resume->scramble : 2;                                 // U/DefaultParam.java:131
    // once the record set is small the rest of the loop runs as two launches per pass with the loop state in HBM
    static const bool small_off = getenv("RFX_NO_SMALL_PASSES") != nullptr;
    const bool extras = wide && prm->extras != 0;
    bool split_done = false;
    DevRecords unext;                       // UnExtendableReflexivKmer (64 :605); lives outside the alternating arenas
    auto small_tail = [&](bool *took) -> int {
        *took = false;
        if (small_off || a.n > small_pass_limit() || P > small_pass_max_partitions()) return RFX_OK;
        if (extras && !split_done && iterations + 1 <= prm->min_iter + 3) return RFX_OK;   // the split of 64 :584-619 comes first
        Arena *saved = tl_arena;
        tl_arena = nullptr;                 // its second record set and scratch outlive the alternating arenas
        const double t0 = verbose ? now_ms() : 0;
        const int64_t n0 = a.n;
        const int st = small_passes(ctx, a, k, twin, wide, prm->coalesce, prm->min_iter, prm->max_iter, &iterations, &contigNumber,
                                    &scramble, &P, &partitionNumber, trace, trace_cap, &nt);
        tl_arena = saved;
        if (verbose) fprintf(stderr, "small passes from n %lld: %.3f ms -> n %lld after pass %lld\n", (long long)n0, now_ms() - t0,
                             (long long)a.n, (long long)nt);
        *took = st == RFX_OK;
        return st;
    };
    // a record set copied out of the arenas (plain stream-ordered allocations)
    auto detach = [&](const DevRecords &src, DevRecords &dst) -> int {
        Arena *saved = tl_arena;
        tl_arena = nullptr;
        int st = dev_records_alloc(ctx, dst, src.n, src.words, src.kw);
        tl_arena = saved;
        RFX_TRY(st);
        if (src.n > 0) {
            RFX_HIP(hipMemcpyAsync(dst.key.p, src.key.p, (size_t)src.n * 8 * src.kw, hipMemcpyDeviceToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dst.marker.p, src.marker.p, (size_t)src.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dst.left.p, src.left.p, (size_t)src.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dst.right.p, src.right.p, (size_t)src.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            if (src.words > 0) RFX_HIP(hipMemcpyAsync(dst.ext.p, src.ext.p, (size_t)src.words * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
        RFX_HIP(hipMemcpyAsync(dst.ext_off.p, src.ext_off.p, (size_t)(src.n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
        dst.n = src.n; dst.words = src.words;
        return RFX_OK;
    };
    while (wide && iterations <= prm->max_iter) {                                 // 64 :582
        { bool took; RFX_TRY(small_tail(&took)); if (took) break; }
        iterations++;
        if (extras && iterations == prm->min_iter + 3) {                          // 64 :584-619
            DevRecords dbl, pe, pu, tmp;
            // sort, DSReflexivAndForwardKmer, sort  (:587-590)
            next_arena();
            RFX_TRY(sort_records(ctx, a, P, key_bits, b, ps, k));
            next_arena();
            RFX_TRY(extras_operator(ctx, RFX_OP_DOUBLE, b, ps.as<int64_t>(), P, k, a, ops));
            RFX_TRY(detach(a, tmp));                                              // (the doubled set is read twice below)
            next_arena();
            RFX_TRY(sort_records(ctx, tmp, P, key_bits, b, ps, k));
            RFX_TRY(detach(b, dbl));
            DevBuf dps;
            { Arena *sv = tl_arena; tl_arena = nullptr; hipError_t e = dps.alloc((size_t)(P + 1) * 8, ctx->stream); tl_arena = sv; RFX_HIP(e); }
            RFX_HIP(hipMemcpyAsync(dps.p, ps.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
            // the two filters on the doubled set, each followed by sort + first-of-key  (:593-605)
            for (int t = 0; t < 2; t++) {
                next_arena();
                RFX_TRY(extras_operator(ctx, t == 0 ? RFX_OP_EXTENDABLE_PAIRS : RFX_OP_UNEXTENDABLE, dbl, dps.as<int64_t>(), P, k, a, ops));
                next_arena();
                RFX_TRY(sort_records(ctx, a, P, key_bits, b, ps, k));
                next_arena();
                RFX_TRY(extras_operator(ctx, RFX_OP_FIRST_OF_KEY, b, ps.as<int64_t>(), P, k, a, ops));
                RFX_TRY(detach(a, t == 0 ? pe : pu));
            }
            // ExtendableReflexivKmer is what the loop goes on with; UnExtendableReflexivKmer waits for the union
            next_arena();
            RFX_TRY(detach(pu, unext));
            {   // pe -> a in the current arena
                RFX_TRY(dev_records_alloc(ctx, a, pe.n, pe.words, pe.kw));
                if (pe.n > 0) {
                    RFX_HIP(hipMemcpyAsync(a.key.p, pe.key.p, (size_t)pe.n * 8 * pe.kw, hipMemcpyDeviceToDevice, ctx->stream));
                    RFX_HIP(hipMemcpyAsync(a.marker.p, pe.marker.p, (size_t)pe.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    RFX_HIP(hipMemcpyAsync(a.left.p, pe.left.p, (size_t)pe.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    RFX_HIP(hipMemcpyAsync(a.right.p, pe.right.p, (size_t)pe.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    if (pe.words > 0) RFX_HIP(hipMemcpyAsync(a.ext.p, pe.ext.p, (size_t)pe.words * 8, hipMemcpyDeviceToDevice, ctx->stream));
                }
                RFX_HIP(hipMemcpyAsync(a.ext_off.p, pe.ext_off.p, (size_t)(pe.n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
                a.n = pe.n; a.words = pe.words;
            }
            RFX_TRY(sync_checked(ctx));
            split_done = true;
        }
        if (iterations >= prm->min_iter + 3 && iterations % 3 == 0) {             // 64 :621-622
            const int64_t current = a.n;                                          // 64 :633-635
            if (contigNumber == current) {                                        // 64 :639
                if (scramble == 2) { scramble = 3; contigNumber = current; }      // 64 :640-642
                else break;                                                       // 64 :644
            } else contigNumber = current;                                        // 64 :647
            // the coalesce of 64 :651-655 is assigned to a variable the loop does not read: no effect
        }
        RFX_TRY(one_pass(2, scramble == 3 ? 1 : 2));                              // 64 :659-660, :667-669
    }
    while (!wide && iterations <= prm->max_iter) {                                // :265-296
        { bool took; RFX_TRY(small_tail(&took)); if (took) break; }
        iterations++;
        if (iterations >= prm->min_iter && iterations % 3 == 0) {
            const int64_t current = a.n;                                          // count() :270
            if (contigNumber == current) break;
            contigNumber = current;
            if (prm->coalesce && partitionNumber >= 16 && current / partitionNumber <= 20) {
                partitionNumber = partitionNumber / 4 + 1;                        // :277-281
                P = partitionNumber;
            }
        }
        RFX_TRY(one_pass(2));
    }
    if (n_trace) *n_trace = nt;
    if (split_done) {                                                             // 64 :672-712
        // union: the extendable set's records, then the unextendable set's (:678)
        DevRecords u;
        { Arena *sv = tl_arena; tl_arena = nullptr; int st = dev_records_alloc(ctx, u, a.n + unext.n, a.words + unext.words, kw); tl_arena = sv; RFX_TRY(st); }
        const DevRecords *two[2] = {&a, &unext};
        int64_t m = 0, w = 0;
        for (int t = 0; t < 2; t++) {
            const DevRecords &r = *two[t];
            if (r.n > 0) {
                RFX_HIP(hipMemcpyAsync(u.key.as<uint64_t>() + m * kw, r.key.p, (size_t)r.n * 8 * kw, hipMemcpyDeviceToDevice, ctx->stream));
                RFX_HIP(hipMemcpyAsync(u.marker.as<int32_t>() + m, r.marker.p, (size_t)r.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                RFX_HIP(hipMemcpyAsync(u.left.as<int32_t>() + m, r.left.p, (size_t)r.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                RFX_HIP(hipMemcpyAsync(u.right.as<int32_t>() + m, r.right.p, (size_t)r.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                if (r.words > 0) RFX_HIP(hipMemcpyAsync(u.ext.as<uint64_t>() + w, r.ext.p, (size_t)r.words * 8, hipMemcpyDeviceToDevice, ctx->stream));
            }
            m += r.n; w += r.words;
        }
        // ext_off of the union: built on the host (the sets are small by now)
        {
            std::vector<int64_t> ha((size_t)a.n + 1), hu((size_t)unext.n + 1), ho((size_t)(a.n + unext.n) + 1);
            RFX_HIP(hipMemcpyAsync(ha.data(), a.ext_off.p, ha.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            RFX_HIP(hipMemcpyAsync(hu.data(), unext.ext_off.p, hu.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            RFX_TRY(sync_checked(ctx));
            for (int64_t i = 0; i <= a.n; i++) ho[(size_t)i] = ha[(size_t)i];
            for (int64_t i = 0; i <= unext.n; i++) ho[(size_t)(a.n + i)] = a.words + hu[(size_t)i];
            RFX_HIP(hipMemcpyAsync(u.ext_off.p, ho.data(), ho.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            RFX_TRY(sync_checked(ctx));
        }
        u.n = a.n + unext.n; u.words = a.words + unext.words;
        // left ends: all forward, sort, longer-of-key; right ends: all reflected, sort, longer-of-key  (:689-707)
        DevBuf one;
        { Arena *sv = tl_arena; tl_arena = nullptr; hipError_t e = one.alloc(16, ctx->stream); tl_arena = sv; RFX_HIP(e); }
        const DevRecords *cur = &u;
        for (int side = 0; side < 2; side++) {
            int64_t h1[2] = {0, cur->n};
            RFX_HIP(hipMemcpyAsync(one.p, h1, 16, hipMemcpyHostToDevice, ctx->stream));
            RFX_TRY(sync_checked(ctx));
            next_arena();
            DevBuf ops1;
            RFX_TRY(extras_operator(ctx, side == 0 ? RFX_OP_ALL_FORWARD : RFX_OP_ALL_REFLECTED, *cur, one.as<int64_t>(), 1, k, b, ops1));
            next_arena();
            DevRecords srt;
            RFX_TRY(sort_records(ctx, b, P, key_bits, srt, ps, k));
            next_arena();
            RFX_TRY(extras_operator(ctx, RFX_OP_LONGER_OF_KEY, srt, ps.as<int64_t>(), P, k, a, ops));
            cur = &a;
            if (side == 0) { RFX_TRY(detach(a, u)); cur = &u; }                   // (a's arena is about to be reused)
        }
    }
    DevRecords *fin = &a;
    if (wide) {                                                                   // 64 :714
        next_arena();
        RFX_TRY(sort_records(ctx, a, P, key_bits, b, ps, k));
        fin = &b;
    }
    const double t_loop1 = verbose ? now_ms() : 0;
    // the surviving records come down through the context's pinned staging block
    const DevRecords &a_fin = *fin;
    const int64_t hn = std::max<int64_t>(a_fin.n, 1), hw = std::max<int64_t>(a_fin.words, 1);
    const size_t need = (size_t)hn * 8 * kw + (size_t)(hn + 1) * 8 + (size_t)hw * 8 + (size_t)hn * 12 + 64;
    char *pin = (char *)ctx->pinned_get(need);
    if (!pin) { ctx->last_error = "pinned staging allocation failed"; return RFX_E_HIP; }
    rfx_records hv{};
    hv.key = (uint64_t *)pin; pin += (size_t)hn * 8 * kw;
    hv.ext_off = (int64_t *)pin; pin += (size_t)(hn + 1) * 8;
    hv.ext = (uint64_t *)pin; pin += (size_t)hw * 8;
    hv.marker = (int32_t *)pin; pin += (size_t)hn * 4;
    hv.left = (int32_t *)pin; pin += (size_t)hn * 4;
    hv.right = (int32_t *)pin;
    hv.cap_n = a_fin.n; hv.cap_words = a_fin.words;
    RFX_TRY(dev_records_download(ctx, a_fin, &hv));
    const double t_dl = verbose ? now_ms() : 0;
    int64_t len = contigs_text_host(&hv, k, prm->min_contig, twin, out, out ? cap : 0, out_contigs);
    if (verbose) fprintf(stderr, "assemble: loop %.3f ms, download %.3f ms, text %.3f ms\n", t_loop1 - t_loop0, t_dl - t_loop1,
                         now_ms() - t_dl);
    *out_len = len;
    return len > cap ? RFX_E_CAP : RFX_OK;
}

extern "C" {

int rfx_dev_assemble(rfx_ctx *ctx, const uint64_t *d_keys, const int32_t *d_counts, int64_t n,
                     const rfx_params *prm, char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs,
                     int64_t *trace, int64_t trace_cap, int64_t *n_trace) try {
    return assemble_impl(ctx, false, d_keys, d_counts, n, prm, out, cap, out_len, out_contigs, trace, trace_cap, n_trace, nullptr);
} RFX_API_CATCH(ctx)

int rfx_dev_assemble_w(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n,
                       const rfx_params *prm, char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs,
                       int64_t *trace, int64_t trace_cap, int64_t *n_trace) try {
    return assemble_impl(ctx, true, d_kmers, d_counts, n, prm, out, cap, out_len, out_contigs, trace, trace_cap, n_trace, nullptr);
} RFX_API_CATCH(ctx)

int rfx_assemble_counts_w(rfx_ctx *ctx, const uint64_t *kmers, const int32_t *counts, int64_t n, const rfx_params *prm,
                          char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs, int64_t *trace, int64_t trace_cap,
                          int64_t *n_trace) try {
    if (!ctx || !prm || !out_len || n < 0 || (n > 0 && (!kmers || !counts))) return RFX_E_ARG;
    RFX_TRY(check_k_rec(prm->k));
    if (prm->k <= 31) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    const int aw = asm_words(prm->k);
    DevBuf dk, dc;
    RFX_HIP(dk.alloc((size_t)std::max<int64_t>(n, 1) * 8 * aw, ctx->stream));
    RFX_HIP(dc.alloc((size_t)std::max<int64_t>(n, 1) * 4, ctx->stream));
    if (n > 0) {
        RFX_HIP(hipMemcpyAsync(dk.p, kmers, (size_t)n * 8 * aw, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(dc.p, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    return assemble_impl(ctx, true, dk.as<uint64_t>(), dc.as<int32_t>(), n, prm, out, cap, out_len, out_contigs, trace, trace_cap, n_trace, nullptr);
} RFX_API_CATCH(ctx)

int rfx_dev_order_kmers_w(rfx_ctx *ctx, uint64_t *d_keys, int64_t *d_counts, int64_t n, int k) try {
    if (!ctx || n < 0 || (n > 0 && (!d_keys || !d_counts))) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    if (!wide_fast_path(k)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    return order_wide2(ctx, d_keys, d_counts, n, k);
} RFX_API_CATCH(ctx)

int rfx_dev_counter_to_asm(rfx_ctx *ctx, const uint64_t *d_keys32, const int64_t *d_counts64, int64_t n, int k,
                           int min_cov, int max_cov, uint64_t *d_out_kmers, int32_t *d_out_counts, int64_t *out_n) try {
    if (!ctx || n < 0 || !out_n || (n > 0 && (!d_keys32 || !d_counts64 || !d_out_kmers || !d_out_counts))) return RFX_E_ARG;
    RFX_TRY(check_k_w(k));
    RFX_TRY(check_k_rec(k));
    RFX_HIP(hipSetDevice(ctx->device));
    return counter_to_asm(ctx, d_keys32, d_counts64, n, k, min_cov, max_cov, d_out_kmers, d_out_counts, out_n);
} RFX_API_CATCH(ctx)

int rfx_assemble_reads(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off, int64_t n_reads,
                       const rfx_params *prm, char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs,
                       int64_t *trace, int64_t trace_cap, int64_t *n_trace, int64_t *out_kept) try {
    if (!ctx || !read_off || !prm || !out_len || n_reads < 0) return RFX_E_ARG;
    RFX_TRY(check_k(prm->k));
    RFX_HIP(hipSetDevice(ctx->device));
    const bool verbose = getenv("RFX_TRACE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_in = now_ms();
    const int64_t nb = n_reads ? read_off[n_reads] - read_off[0] : 0;
    // the two big buffers live in workspace slots of the context (grow-only, allocated once): slot 5 = the ASCII bases,
    // slot 6 = offsets | packed words | lengths
    DevSpan d_bases, d_off, d_words, d_len;
    DevBuf d_keys, d_counts;
    d_bases.p = ctx->ws_get(5, (size_t)std::max<int64_t>(nb, 1));
    if (!d_bases.p) { ctx->last_error = "rfx_assemble_reads: out of device memory (bases)"; return RFX_E_HIP; }
    // PCIe is the slow part (5 GB of ASCII at ~50 GB/s from pinned memory), so everything else hides under it: the bases go
    // up in chunks of whole reads on a second stream, queued FIRST; while they travel the host scans the read lengths (the
    // width of the packed layout), and every chunk is 2-bit encoded on the context's stream as soon as it has landed.  The
    // offsets go up as they are: k_encode reads bases[read_off[r] + i], the device copy is addressed from read_off[0].
    // (the copy stream is DRAINED before it is destroyed on every way out, exceptions included: copies out of the caller's
    // `bases` may still be in flight when an error turns up)
    struct Upload {
        hipStream_t cs = nullptr;
        std::vector<hipEvent_t> evs;
        hipEvent_t ready = nullptr;
        ~Upload() {
            if (cs) (void)hipStreamSynchronize(cs);
            for (auto e : evs) (void)hipEventDestroy(e);
            if (ready) (void)hipEventDestroy(ready);
            if (cs) (void)hipStreamDestroy(cs);
        }
    } up;
    hipStream_t &cs = up.cs;
    std::vector<hipEvent_t> &evs = up.evs;
    std::vector<int64_t> cuts(1, 0);
    hipEvent_t &ready = up.ready;                          // the allocations above are stream-ordered on the context's stream
    hipError_t err = hipSuccess;
    if (n_reads > 0) {
        err = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventRecord(ready, ctx->stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(cs, ready, 0);
        const int64_t chunk_bytes = (int64_t)256 << 20;
        int64_t r0 = 0;
        while (err == hipSuccess && r0 < n_reads) {
            // reads [r0, r1): as many whole reads as fit the chunk (binary search on the ascending offsets)
            int64_t lo = r0 + 1, hi = n_reads;
            const int64_t limit = read_off[r0] + chunk_bytes;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (read_off[mid] <= limit) lo = mid; else hi = mid - 1; }
            const int64_t r1 = lo, b0 = read_off[r0], b1 = read_off[r1];
            if (b1 > b0) err = hipMemcpyAsync(d_bases.as<uint8_t>() + (b0 - read_off[0]), bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, cs);
            hipEvent_t e = nullptr;
            if (err == hipSuccess) err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (err == hipSuccess) { evs.push_back(e); err = hipEventRecord(e, cs); }
            cuts.push_back(r1);
            r0 = r1;
        }
    }
    int64_t maxlen = 1, minlen = INT64_MAX;
    for (int64_t r = 0; r < n_reads; r++) {
        const int64_t l = read_off[r + 1] - read_off[r];
        maxlen = std::max(maxlen, l); minlen = std::min(minlen, l);
    }
    const bool uniform = n_reads > 0 && minlen == maxlen;       // equal-length reads (PE150): the uniform count path
    const int wpr = (int)((maxlen + 31) / 32);
    const double t_scan = now_ms();
    int st_enc = RFX_OK;
    {
        const size_t off_b = ((size_t)(n_reads + 1) * 8 + 255) & ~(size_t)255, words_b = ((size_t)std::max<int64_t>(n_reads, 1) * wpr * 8 + 255) & ~(size_t)255;
        char *base = (char *)ctx->ws_get(6, off_b + words_b + (size_t)std::max<int64_t>(n_reads, 1) * 4);
        if (!base) { ctx->last_error = "rfx_assemble_reads: out of device memory (packed reads)"; err = hipErrorOutOfMemory; }
        d_off.p = base; d_words.p = base + off_b; d_len.p = base + off_b + words_b;
        if (err == hipSuccess && n_reads > 0) err = hipMemcpyAsync(d_off.p, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    }
    const uint8_t *dev_base = d_bases.as<uint8_t>() - (n_reads ? read_off[0] : 0);
    for (size_t c = 0; err == hipSuccess && st_enc == RFX_OK && c < evs.size(); c++) {
        err = hipStreamWaitEvent(ctx->stream, evs[c], 0);
        const int64_t r0 = cuts[c], r1 = cuts[c + 1];
        if (err == hipSuccess)
            st_enc = encode_reads(ctx, dev_base, d_off.as<int64_t>() + r0, r1 - r0, wpr, d_words.as<uint64_t>() + r0 * wpr, d_len.as<uint32_t>() + r0);
    }
    if (cs) { const hipError_t e2 = hipStreamSynchronize(cs); if (err == hipSuccess) err = e2; }
    { const hipError_t e2 = hipStreamSynchronize(ctx->stream); if (err == hipSuccess) err = e2; }
    if (err != hipSuccess) { ctx->last_error = std::string("rfx_assemble_reads upload: ") + hipGetErrorString(err); return RFX_E_HIP; }
    // the ASCII staging is dead weight from here on (the packed reads are a quarter of it): a large one goes back to the
    // driver now, so that the count and extend stages of a capacity-stress run have the HBM; up to RFX_KEEP_STAGING_BYTES
    // (default 8 GiB: config 2's 5 GB stays, and with it the 100-600 ms a fresh allocation of that size costs per call) it
    // is kept for the next call.  rfx_ctx_trim() releases everything at any time.
    {
        const char *e = getenv("RFX_KEEP_STAGING_BYTES");
        const size_t keep = e ? (size_t)atoll(e) : (size_t)8 << 30;
        if (ctx->ws[5].bytes > keep) { (void)hipFree(ctx->ws[5].p); ctx->ws[5].p = nullptr; ctx->ws[5].bytes = 0; d_bases.p = nullptr; }
    }
    RFX_TRY(st_enc);
    ReadStore rs{d_words.as<uint64_t>(), n_reads, wpr, (int)maxlen, prm->k, prm->front_clip, prm->end_clip};
    const double t_up = now_ms();
    if (verbose) fprintf(stderr, "assemble_reads: upload queued + length scan %.1f ms, until the last chunk is encoded %.1f ms more (%.1f GB/s of ASCII in all)\n",
                         t_scan - t_in, t_up - t_scan, nb / ((t_up - t_in) * 1e6));
    int64_t n_inst = 0;
    if (!uniform) {
        rs.read_len_arr = d_len.as<uint32_t>();
        RFX_TRY(ragged_instances(ctx, rs.read_len_arr, n_reads, prm->k, prm->front_clip, prm->end_clip, &rs.n_instances));
        n_inst = rs.n_instances;
    } else n_inst = kmers_per_read((int)maxlen, prm->k, prm->front_clip, prm->end_clip) * n_reads;
    int64_t m = 0, dist = 0;
    int64_t kcap = std::max<int64_t>(1 << 20, n_inst / 8);
    for (;;) {                                      // survivors are few; grow on RFX_E_CAP
        RFX_HIP(d_keys.alloc((size_t)kcap * 8, ctx->stream));
        RFX_HIP(d_counts.alloc((size_t)kcap * 4, ctx->stream));
        const int st = count_filter(ctx, &rs, nullptr, 0, prm->min_cov, prm->max_cov, prm->twin, nullptr, 0,
                                    d_keys.as<uint64_t>(), d_counts.as<int32_t>(), kcap, &m, &dist);
        if (st == RFX_E_CAP && m > kcap) { kcap = m; continue; }
        RFX_TRY(st);
        break;
    }
    if (out_kept) *out_kept = m;
    if (verbose) fprintf(stderr, "assemble_reads: count %.1f ms\n", now_ms() - t_up);
    return rfx_dev_assemble(ctx, d_keys.as<uint64_t>(), d_counts.as<int32_t>(), m, prm, out, cap, out_len, out_contigs, trace,
                            trace_cap, n_trace);
} RFX_API_CATCH(ctx)

}  // extern "C"
