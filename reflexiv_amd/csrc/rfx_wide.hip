// rfx_wide.hip -- k > 31: the counter's multi-word canonical k-mers.
//
//   reference: P/ReflexivDataFrameCounter64.java
//     ReverseComplementKmerBinaryExtractionFromDataset64.call :401-650   -> k_extract_w
//     compareLongArrayBlocks :652-687 (base-wise fwd < rc, ties -> fwd)  -> word-wise unsigned compare
//     groupBy("kmerBlocks").count() + filters :191-205                   -> sort + run heads + compaction
//
// Layout as in the reference: W = k/32+1 words per k-mer, words 0..W-2 hold 32 bases each, the last
// word the k%32 remaining bases right-aligned; k > 32, k % 32 != 0.  On the device the words live as
// structure of arrays (word w of k-mer i at [w*N + i]); the C ABI speaks the reference's array of
// W longs per k-mer.
//
// This round's k > 31 count is the plain exact formulation -- LSD radix sort of the W words through a
// 32-bit permutation, equal-key run heads, scan, compaction -- not the bucketed LDS-table path of
// k <= 31: ~70 B of workspace per instance, N < 2^32 per call.  
#include "rfx_internal.h"
#include "rfx_device.h"

using namespace rfxd;

namespace {

constexpr int MAXW = 8;

// n <= 32 bases starting at base p of a packed read, right-aligned
__device__ __forceinline__ uint64_t chunk_at(const uint64_t *__restrict__ w, int p, int n) { return kmer_at(w, p, n); }

// counter64 skip rule (:410) and loop bounds (:419, :524)
__device__ __host__ __forceinline__ int64_t nk_of_w(int64_t len, int k, int fc, int ec) {
    if (len - k - ec + 1 <= 0 || fc > len) return 0;
    const int64_t m = (len - ec - fc) - (k - 1);
    return m > 0 ? m : 0;
}

__global__ void k_nk_per_read_w(const int64_t *__restrict__ read_off, int64_t n_reads, int k, int fc, int ec,
                                uint64_t *__restrict__ nk) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) nk[r] = (uint64_t)nk_of_w(read_off[r + 1] - read_off[r], k, fc, ec);
}

// One wave per read, lanes over window positions; reference emission order (read, window).
// kmer_off == nullptr: uniform reads, read r emits nk_uniform windows at r * nk_uniform.
__global__ void k_extract_w(const uint64_t *__restrict__ words, int wpr, const uint64_t *__restrict__ kmer_off,
                            int64_t nk_uniform, int64_t n_reads, int k, int fc, uint64_t *__restrict__ out, int64_t N,
                            int aos) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_reads) return;
    const int lane = lane_id();
    const int W = k / 32 + 1, res = k % 32;
    const uint64_t o = kmer_off ? kmer_off[r] : (uint64_t)(r * nk_uniform);
    const int64_t nk = kmer_off ? (int64_t)(kmer_off[r + 1] - o) : nk_uniform;
    const uint64_t *w = words + r * wpr;
    for (int64_t p = lane; p < nk; p += 64) {
        const int b = fc + (int)p;                      // first base of the window
        uint64_t f[MAXW], rc[MAXW];
        // forward: 32-base chunks from the left, then the residue
        for (int j = 0; j < W - 1; j++) f[j] = chunk_at(w, b + 32 * j, 32);
        f[W - 1] = chunk_at(w, b + 32 * (W - 1), res);
        // reverse complement: word j = revcomp of the 32 bases that end 32*j bases before the
        // window's end; the last word = revcomp of the window's first `res` bases
        for (int j = 0; j < W - 1; j++) rc[j] = revcomp(chunk_at(w, b + k - 32 * (j + 1), 32), 32);
        rc[W - 1] = revcomp(chunk_at(w, b, res), res);
        bool use_f = true;                              // ties -> forward (:685-686)
        for (int j = 0; j < W; j++) {
            if (f[j] != rc[j]) { use_f = f[j] < rc[j]; break; }
        }
        for (int j = 0; j < W; j++) out[aos ? ((int64_t)o + p) * W + j : (int64_t)j * N + (int64_t)o + p] = use_f ? f[j] : rc[j];
    }
}

// uniform reads: one thread per window, consecutive threads on consecutive windows (and consecutive
// output elements); the wave-per-read form above left a third of the lanes idle on 88-window reads
__global__ void k_extract_w_flat(const uint64_t *__restrict__ words, int wpr, int64_t nk, int64_t N, int k, int fc,
                                 uint64_t *__restrict__ out, int aos) {
    const int W = k / 32 + 1, res = k % 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nk;
        const int b = fc + (int)(i - r * nk);
        const uint64_t *w = words + r * wpr;
        uint64_t f[MAXW], rc[MAXW];
        for (int j = 0; j < W - 1; j++) f[j] = chunk_at(w, b + 32 * j, 32);
        f[W - 1] = chunk_at(w, b + 32 * (W - 1), res);
        for (int j = 0; j < W - 1; j++) rc[j] = revcomp(chunk_at(w, b + k - 32 * (j + 1), 32), 32);
        rc[W - 1] = revcomp(chunk_at(w, b, res), res);
        bool use_f = true;
        for (int j = 0; j < W; j++) {
            if (f[j] != rc[j]) { use_f = f[j] < rc[j]; break; }
        }
        for (int j = 0; j < W; j++) out[aos ? i * W + j : (int64_t)j * N + i] = use_f ? f[j] : rc[j];
    }
}

// k = 33..63, uniform reads: a thread owns 16 consecutive windows of one read and ROLLS the two-word
// k-mer and its reverse complement through them (one base in, one out on both strands: a dozen
// operations per window instead of four funnel-shifted chunks and two full reverse complements).
constexpr int WSEG = 16;
__global__ __launch_bounds__(256) void k_extract_w2_roll(const uint64_t *__restrict__ words, int wpr, int64_t nk, int64_t n_reads,
                                                         int k, int fc, uint64_t *__restrict__ out /* AoS */) {
    const int res = k - 32;                          // bases in the second word, 1..31
    const int64_t segs = (nk + WSEG - 1) / WSEG;
    const int64_t total = n_reads * segs;
    const uint64_t mres = low_mask(res);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / segs;
        const int p0 = (int)(g - r * segs) * WSEG;
        int v = (int)(nk - p0);
        v = v > WSEG ? WSEG : v;
        const uint64_t *w = words + r * wpr;
        const int b = fc + p0;
        uint64_t f0 = chunk_at(w, b, 32), f1 = chunk_at(w, b + 32, res);
        uint64_t r0 = revcomp(chunk_at(w, b + k - 32, 32), 32), r1 = revcomp(chunk_at(w, b, res), res);
        const uint64_t nxt = v > 1 ? chunk_at(w, b + k, v - 1) : 0;      // the v-1 bases that enter, right-aligned
        uint64_t *o = out + 2 * (r * nk + p0);
        for (int j = 0; j < v; j++) {
            const bool use_f = f0 != r0 ? f0 < r0 : f1 <= r1;            // ties -> forward
            o[2 * j] = use_f ? f0 : r0;
            o[2 * j + 1] = use_f ? f1 : r1;
            if (j + 1 < v) {
                const uint64_t nb = (nxt >> (2 * (v - 2 - j))) & 3;
                const uint64_t cf = f1 >> (2 * (res - 1));
                f1 = ((f1 << 2) | nb) & mres;
                f0 = (f0 << 2) | cf;
                const uint64_t cr = r0 & 3;
                r0 = (r0 >> 2) | ((nb ^ 3) << 62);
                r1 = (r1 >> 2) | (cr << (2 * (res - 1)));
            }
        }
    }
}

__global__ void k_iota(uint32_t *__restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}

__global__ void k_gather_u64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx, int64_t n,
                             uint64_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// AoS (host layout) <-> SoA
__global__ void k_aos_to_soa(const uint64_t *__restrict__ aos, int64_t n, int W, uint64_t *__restrict__ soa) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * W) { const int64_t i = t / W; const int w = (int)(t - i * W); soa[(int64_t)w * n + i] = aos[t]; }
}
__global__ void k_soa_to_aos(const uint64_t *__restrict__ soa, int64_t n, int W, uint64_t *__restrict__ aos) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * W) { const int64_t i = t / W; const int w = (int)(t - i * W); aos[t] = soa[(int64_t)w * n + i]; }
}

// sorted (SoA) -> 1 where a new key starts
__global__ void k_heads_w(const uint64_t *__restrict__ sw, int64_t n, int W, uint32_t *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    for (int w = 0; w < W && !h; w++) h = sw[(int64_t)w * n + i] != sw[(int64_t)w * n + i - 1];
    head[i] = h ? 1u : 0u;
}

__global__ void k_starts_w(const uint32_t *__restrict__ head, const uint64_t *__restrict__ pos, int64_t n,
                           uint64_t *__restrict__ start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) start[pos[i]] = (uint64_t)i;
    if (i == 0) start[pos[n]] = (uint64_t)n;
}

// the two filters of the counter (:197-205)
__global__ void k_keep_w(const uint64_t *__restrict__ start, int64_t D, int min_cov, int max_cov,
                         uint32_t *__restrict__ keep) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint64_t c = start[d + 1] - start[d];
    bool k = true;
    if (min_cov > 1 && c < (uint64_t)min_cov) k = false;
    if (max_cov < 10000000 && c > (uint64_t)max_cov) k = false;
    keep[d] = k ? 1u : 0u;
}

__global__ void k_emit_w(const uint64_t *__restrict__ sw, int64_t n, int W, const uint64_t *__restrict__ start,
                         const uint32_t *__restrict__ keep, const uint64_t *__restrict__ opos, int64_t D, int64_t cap,
                         uint64_t *__restrict__ out_keys, int64_t *__restrict__ out_counts) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D || !keep[d]) return;
    const int64_t o = (int64_t)opos[d];
    if (o >= cap) return;
    const int64_t i = (int64_t)start[d];
    for (int w = 0; w < W; w++) out_keys[o * W + w] = sw[(int64_t)w * n + i];
    out_counts[o] = (int64_t)(start[d + 1] - start[d]);
}

// final ordering of the fast path's survivors: (hi, lo, count) through the sorted permutation
__global__ void k_permute_w2(const uint64_t *__restrict__ aos, const int64_t *__restrict__ cnt,
                             const uint32_t *__restrict__ idx, int64_t m, uint64_t *__restrict__ oaos,
                             int64_t *__restrict__ ocnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t s = idx[i];
    oaos[2 * i] = aos[2 * (int64_t)s]; oaos[2 * i + 1] = aos[2 * (int64_t)s + 1];
    ocnt[i] = cnt[s];
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, ceil_div(n, 256)); }

}  // namespace

namespace rfx {

int check_k_w(int k) { return (k > 32 && k % 32 != 0 && k / 32 + 1 <= MAXW) ? RFX_OK : RFX_E_ARG; }

int64_t kmers_per_read_w(int read_len, int k, int fc, int ec) { return nk_of_w(read_len, k, fc, ec); }

int kmer_counts_per_read_w(rfx_ctx *ctx, const int64_t *d_read_off, int64_t n_reads, int k, int fc, int ec,
                           uint64_t *d_nk) {
    if (n_reads <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_nk_per_read_w, dim3(grid_for(n_reads)), dim3(256), 0, ctx->stream, d_read_off, n_reads, k, fc,
                       ec, d_nk);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

// packed reads -> canonical W-word k-mers, SoA [w*N + i]
int extract_w(rfx_ctx *ctx, const uint64_t *d_words, int wpr, const uint64_t *d_kmer_off, int64_t nk_uniform,
              int64_t n_reads, int k, int fc, uint64_t *d_soa, int64_t N, int aos) {
    if (n_reads <= 0 || N <= 0) return RFX_OK;
    if (!d_kmer_off && aos && k / 32 + 1 == 2 && !getenv("RFX_WIDE_NOROLL")) {
        const int64_t total = n_reads * ceil_div(nk_uniform, WSEG);
        const int64_t blocks = std::min<int64_t>(ceil_div(total, 256), (int64_t)ctx->num_cu * 32);
        hipLaunchKernelGGL(k_extract_w2_roll, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_words, wpr, nk_uniform,
                           n_reads, k, fc, d_soa);
        RFX_HIP(hipGetLastError());
        return RFX_OK;
    }
    if (!d_kmer_off) {
        const int64_t blocks = std::min<int64_t>(ceil_div(N, 256), (int64_t)ctx->num_cu * 32);
        hipLaunchKernelGGL(k_extract_w_flat, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_words, wpr, nk_uniform, N, k,
                           fc, d_soa, aos);
        RFX_HIP(hipGetLastError());
        return RFX_OK;
    }
    hipLaunchKernelGGL(k_extract_w, dim3(grid_for(n_reads * 64)), dim3(256), 0, ctx->stream, d_words, wpr, d_kmer_off,
                       nk_uniform, n_reads, k, fc, d_soa, N, aos);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int aos_to_soa(rfx_ctx *ctx, const uint64_t *d_aos, int64_t n, int W, uint64_t *d_soa) {
    if (n <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_aos_to_soa, dim3(grid_for(n * W)), dim3(256), 0, ctx->stream, d_aos, n, W, d_soa);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int soa_to_aos(rfx_ctx *ctx, const uint64_t *d_soa, int64_t n, int W, uint64_t *d_aos) {
    if (n <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_soa_to_aos, dim3(grid_for(n * W)), dim3(256), 0, ctx->stream, d_soa, n, W, d_aos);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

bool wide_fast_path(int k) {
    return k / 32 + 1 == 2 && !(getenv("RFX_WIDE_SORT") && atoi(getenv("RFX_WIDE_SORT")) == 1);
}

// k = 33..63: the bucketed path (hash digits, write-combining scatters, LDS-table leaves with two-word
// keys: rfx_kmer.hip count_wide2) on N elements {word0, word1}; the survivors are then put in
// ascending order.
int count_filter_w2(rfx_ctx *ctx, const uint64_t *d_elems, int64_t N, int k, int min_cov, int max_cov,
                    uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) {
    *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (N <= 0) return RFX_OK;
    if (N >= (1LL << 32)) { ctx->last_error = "k > 31 count: at most 2^32-1 instances per call"; return RFX_E_ARG; }
    int64_t m = 0;
    const int st = count_wide2(ctx, d_elems, N, min_cov, max_cov, d_out_keys, d_out_counts, cap, &m, out_distinct);
    *out_n = m;
    if (st != RFX_OK) return st;
    return order_wide2(ctx, d_out_keys, d_out_counts, m, k);
}

// survivors of the fast path (unordered, AoS) -> ascending by (word0, word1)
int order_wide2(rfx_ctx *ctx, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t m, int k) {
    if (m <= 1) return RFX_OK;
    if (m >= (1LL << 32)) { ctx->last_error = "k > 31 count: too many survivors to order"; return RFX_E_LIMIT; }
    const int res = k % 32;
    ScopedTimer t(ctx, "sort");
    DevBuf soa, idx, idx2, keys, keys2, oaos, ocnt;
    RFX_HIP(soa.alloc((size_t)m * 16, ctx->stream));
    RFX_HIP(idx.alloc((size_t)m * 4, ctx->stream));
    RFX_HIP(idx2.alloc((size_t)m * 4, ctx->stream));
    RFX_HIP(keys.alloc((size_t)m * 8, ctx->stream));
    RFX_HIP(keys2.alloc((size_t)m * 8, ctx->stream));
    RFX_TRY(aos_to_soa(ctx, d_out_keys, m, 2, soa.as<uint64_t>()));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(m)), dim3(256), 0, ctx->stream, idx.as<uint32_t>(), m);
    RFX_HIP(hipGetLastError());
    for (int w = 1; w >= 0; w--) {
        hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(m)), dim3(256), 0, ctx->stream, soa.as<uint64_t>() + (int64_t)w * m,
                           (const uint32_t *)idx.as<uint32_t>(), m, keys.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(sort_pairs(ctx, keys.as<uint64_t>(), idx.as<uint32_t>(), m, w == 1 ? 2 * res : 64, keys2.as<uint64_t>(),
                           idx2.as<uint32_t>()));
    }
    RFX_HIP(oaos.alloc((size_t)m * 16, ctx->stream));
    RFX_HIP(ocnt.alloc((size_t)m * 8, ctx->stream));
    hipLaunchKernelGGL(k_permute_w2, dim3(grid_for(m)), dim3(256), 0, ctx->stream, (const uint64_t *)d_out_keys,
                       (const int64_t *)d_out_counts, (const uint32_t *)idx.as<uint32_t>(), m, oaos.as<uint64_t>(),
                       ocnt.as<int64_t>());
    RFX_HIP(hipGetLastError());
    RFX_HIP(hipMemcpyAsync(d_out_keys, oaos.p, (size_t)m * 16, hipMemcpyDeviceToDevice, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_out_counts, ocnt.p, (size_t)m * 8, hipMemcpyDeviceToDevice, ctx->stream));
    t.stop();
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

// groupBy + count + filter on N W-word k-mers (SoA, destroyed).  d_out_keys: cap*W words (AoS,
// ascending by base string), d_out_counts: cap int64.
int count_filter_w(rfx_ctx *ctx, uint64_t *d_soa, int64_t N, int k, int min_cov, int max_cov, uint64_t *d_out_keys,
                   int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) {
    *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (N <= 0) return RFX_OK;
    if (N >= (1LL << 32)) { ctx->last_error = "k > 31 count: at most 2^32-1 instances per call"; return RFX_E_ARG; }
    const int W = k / 32 + 1, res = k % 32;
    if (wide_fast_path(k)) {
        DevBuf elems;
        RFX_HIP(elems.alloc((size_t)N * 16, ctx->stream));
        RFX_TRY(soa_to_aos(ctx, d_soa, N, 2, elems.as<uint64_t>()));
        return count_filter_w2(ctx, elems.as<uint64_t>(), N, k, min_cov, max_cov, d_out_keys, d_out_counts, cap, out_n,
                               out_distinct);
    }
    DevBuf idx, idx2, keys, keys2, sorted, head, pos;
    RFX_HIP(idx.alloc((size_t)N * 4, ctx->stream));
    RFX_HIP(idx2.alloc((size_t)N * 4, ctx->stream));
    RFX_HIP(keys.alloc((size_t)N * 8, ctx->stream));
    RFX_HIP(keys2.alloc((size_t)N * 8, ctx->stream));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N)), dim3(256), 0, ctx->stream, idx.as<uint32_t>(), N);
    RFX_HIP(hipGetLastError());
    // least significant word first; every pass is stable, so the result is ordered by (w0, w1, ...)
    for (int w = W - 1; w >= 0; w--) {
        hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(N)), dim3(256), 0, ctx->stream, d_soa + (int64_t)w * N,
                           (const uint32_t *)idx.as<uint32_t>(), N, keys.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(sort_pairs(ctx, keys.as<uint64_t>(), idx.as<uint32_t>(), N, w == W - 1 ? 2 * res : 64, keys2.as<uint64_t>(),
                           idx2.as<uint32_t>()));
    }
    keys.release(); keys2.release(); idx2.release();
    RFX_HIP(sorted.alloc((size_t)N * W * 8, ctx->stream));
    for (int w = 0; w < W; w++) {
        hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(N)), dim3(256), 0, ctx->stream, d_soa + (int64_t)w * N,
                           (const uint32_t *)idx.as<uint32_t>(), N, sorted.as<uint64_t>() + (int64_t)w * N);
        RFX_HIP(hipGetLastError());
    }
    RFX_HIP(head.alloc((size_t)N * 4, ctx->stream));
    RFX_HIP(pos.alloc((size_t)(N + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_heads_w, dim3(grid_for(N)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.as<uint64_t>(), N, W,
                       head.as<uint32_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, head.as<uint32_t>(), pos.as<uint64_t>(), N));
    uint64_t D = 0;
    RFX_HIP(hipMemcpyAsync(&D, pos.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (out_distinct) *out_distinct = (int64_t)D;
    DevBuf start, keep, opos;
    RFX_HIP(start.alloc((size_t)(D + 1) * 8, ctx->stream));
    RFX_HIP(keep.alloc((size_t)D * 4, ctx->stream));
    RFX_HIP(opos.alloc((size_t)(D + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_starts_w, dim3(grid_for(N)), dim3(256), 0, ctx->stream, (const uint32_t *)head.as<uint32_t>(),
                       (const uint64_t *)pos.as<uint64_t>(), N, start.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_keep_w, dim3(grid_for((int64_t)D)), dim3(256), 0, ctx->stream, (const uint64_t *)start.as<uint64_t>(),
                       (int64_t)D, min_cov, max_cov, keep.as<uint32_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, keep.as<uint32_t>(), opos.as<uint64_t>(), (int64_t)D));
    uint64_t M = 0;
    RFX_HIP(hipMemcpyAsync(&M, opos.as<uint64_t>() + D, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    *out_n = (int64_t)M;
    if ((int64_t)M > cap) return RFX_E_CAP;
    if (M == 0) return RFX_OK;
    hipLaunchKernelGGL(k_emit_w, dim3(grid_for((int64_t)D)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.as<uint64_t>(),
                       N, W, (const uint64_t *)start.as<uint64_t>(), (const uint32_t *)keep.as<uint32_t>(),
                       (const uint64_t *)opos.as<uint64_t>(), (int64_t)D, cap, d_out_keys, d_out_counts);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

}  // namespace rfx
