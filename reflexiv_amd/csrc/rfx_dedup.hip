// rfx_dedup.hip -- contig RC de-duplication: P/ReflexivDSDynamicKmerDedup.java on the GPU.
//
// The reference's driver (`assemblyFromKmer`, :138-339) runs three rounds of: marker 31-mers of every contig of >= 300 bases
// (seeds at the block starts, probes at a few windows; round 1 probes the reverse complement only, rounds 2 and 3 both
// strands: :2674-3096, :2206-2673) -> sort -> DSMarkerKmerSelection (:1788-1870: a probe of a shorter contig that meets a
// seed of the same 31-mer names the pair) -> groupBy().count() >= 2 -> the shorter contig is sent to its target
// (:3186-3207, :3097-3132) -> the removal class merges the contigs that share a target into the longest by 15-mer seed
// voting (`merge2RCContigs`, :1462-1557 / :565-728), an unmatched short contig goes back to the pool.
//
// Here: contigs live in HBM, one byte per base.  Kernels: k_dd_markers (one thread per seed / probe), the library's stable
// radix sort on the sign-flipped 31-mer, k_dd_select (the owner of an equal-31-mer run walks it: the reference's one-row
// `LongestKmer` and its `shorterKmer` list never outlive a run of equal k-mers except to be compared with the next run's
// first seed), a second sort + k_dd_pair_runs for count >= 2; per merge k_dd_seed_insert (open-addressing table, the
// HashMap's "a later position replaces an earlier one" as atomicMax), k_dd_query (all 15-mers of the short contig, either
// strand), a sort of the distances, k_dd_vote (the reference's sequential vote, one thread: it is a scan with a data-
// dependent anchor) and k_dd_copy for the flanks.  The pairing of contigs with their targets is bookkeeping on contig IDS
// (a few numbers per contig) and runs on the host between the kernels, as the Spark driver's plan does in the reference.
// Order contract as everywhere: one logical partition, stable sorts on the signed column, union =
// left then right, groupBy().count() ascending, zipWithIndex = position.
#include <algorithm>
#include <string>
#include <vector>
#include "rfx_internal.h"

using namespace rfx;

namespace {

constexpr int M31 = 31;

__device__ __forceinline__ uint64_t dd_mer31(const uint8_t *s, int64_t p) {
    uint64_t x = 0;
#pragma unroll
    for (int j = 0; j < 31; j++) x = (x << 2) | s[p + j];
    return (x << 2) | 1;
}
__device__ __forceinline__ uint64_t dd_mer31_rc(uint64_t m) {       // binaryLongReverseComplementary :2877-2906
    uint64_t x = 0;
#pragma unroll
    for (int j = 0; j < 31; j++) x = (x << 2) | (((m >> (2 * (j + 1))) & 3) ^ 3);
    return (x << 2) | 1;
}
__host__ __device__ __forceinline__ int64_t dd_attr3(int marker, int64_t left, int64_t right) {     // :2908-2934
    if (left >= 500000000) left = 500000000; else if (left <= -500000000) left = 1000000000; else if (left < 0) left = 500000000 - left;
    if (right >= 1000000000) right = 1000000000; else if (right <= -1000000000) right = 2000000000; else if (right < 0) right = 1000000000 - right;
    return (int64_t)(((uint64_t)marker << 62) | ((uint64_t)(uint32_t)left << 32) | (uint64_t)(uint32_t)right);
}
__device__ __forceinline__ int dd_marker(int64_t a) { return (int)((uint64_t)a >> 62); }
__device__ __forceinline__ int32_t dd_left(int64_t a) {               // getLeftMarker :1882-1892
    int32_t l = (int32_t)((uint64_t)a >> 32) & ~(3 << 30);
    if (l > 500000000) l = 500000000 - l;
    return l;
}
__device__ __forceinline__ int32_t dd_right(int64_t a) {              // getRightMarker :1894-1902
    int32_t r = (int32_t)a;
    if (r > 1000000000) r = 1000000000 - r;
    return r;
}

// probe windows of a contig of L bases (getRCKmerProbBinary :2713-2875): up to five [a, b)
__host__ __device__ inline int dd_windows(int64_t L, int64_t w[5][2]) {
    int n = 0;
    const int64_t M = M31;
    if (L >= 4000) {
        w[n][0] = 0; w[n++][1] = M; w[n][0] = 1000 - M + 1; w[n++][1] = 1000; w[n][0] = (L - 2 * M) / 2; w[n][1] = w[n][0] + M; n++;
        w[n][0] = L - 1000 - M + 1; w[n++][1] = L - 1000; w[n][0] = L - 2 * M; w[n++][1] = L - M;
    } else if (L >= 2000) {
        w[n][0] = 0; w[n++][1] = M; w[n][0] = 600 - M + 1; w[n++][1] = 600; w[n][0] = (L - 2 * M) / 2; w[n][1] = w[n][0] + M; n++;
        w[n][0] = L - 600 - M + 1; w[n++][1] = L - 600; w[n][0] = L - 2 * M; w[n++][1] = L - M;
    } else {
        w[n][0] = 0; w[n++][1] = M; w[n][0] = (L - 2 * M) / 3; w[n][1] = w[n][0] + M; n++;
        w[n][0] = (L - 2 * M) * 2 / 3; w[n][1] = w[n][0] + M; n++; w[n][0] = L - 2 * M; w[n++][1] = L - M;
    }
    return n;
}
inline int64_t dd_seed_count(int64_t L) { return L < 300 ? 0 : (L - 1) / 31 + (L % 31 == 0 ? 1 : 0); }
inline int64_t dd_probe_positions(int64_t L) {
    if (L < 300) return 0;
    int64_t w[5][2];
    const int n = dd_windows(L, w);
    int64_t t = 0;
    for (int i = 0; i < n; i++) t += w[i][1] - w[i][0];
    return t;
}

// one thread per marker row, in the reference's emission order: contig after contig; inside a contig the seeds (block
// starts, ascending), then window after window, position after position (forward probe, then its reverse complement, when
// both strands are probed)
__global__ __launch_bounds__(256) void k_dd_markers(const uint8_t *__restrict__ pool, const int64_t *__restrict__ coff,
                                                    const int64_t *__restrict__ clen, const int64_t *__restrict__ cid,
                                                    const int64_t *__restrict__ moff, int64_t n_contigs, int64_t n_markers, int both,
                                                    uint64_t *__restrict__ key, uint32_t *__restrict__ val, int64_t *__restrict__ attr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_markers) return;
    int64_t lo = 0, hi = n_contigs;                       // the contig whose marker range holds t
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (moff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t c = lo, L = clen[c];
    const uint8_t *s = pool + coff[c];
    int64_t j = t - moff[c];
    const int64_t nb = (L - 1) / 31 + 1, nseed = (nb - 1) + (L % 31 == 0 ? 1 : 0);
    uint64_t k;
    int64_t a;
    if (j < nseed) {
        k = dd_mer31(s, 31 * j);
        a = dd_attr3(1, L, (int32_t)cid[c]);
    } else {
        j -= nseed;
        const int per = both ? 2 : 1;
        int64_t p = j / per;
        int64_t w[5][2];
        const int nw = dd_windows(L, w);
        int64_t pos = 0;
        for (int q = 0; q < nw; q++) {
            const int64_t len = w[q][1] - w[q][0];
            if (p < len) { pos = w[q][0] + p; break; }
            p -= len;
        }
        const uint64_t f = dd_mer31(s, pos);
        k = (both && (j % per) == 0) ? f : dd_mer31_rc(f);
        a = dd_attr3(2, L, (int32_t)cid[c]);
    }
    key[t] = k ^ 0x8000000000000000ull;                   // sort("kmerBinary") orders the SIGNED long
    val[t] = (uint32_t)t;
    attr[t] = a;
}

// DSMarkerKmerSelection.call (:1796-1868) on the sorted rows: the thread that owns the head of an equal-31-mer run walks it.
// A probe ahead of the run's first seed meets that seed (`s` of the reference, the row that opens the new k-mer); a probe
// behind it meets the run's longest seed (`LongestKmer` when the list is flushed).  pair[q] = the pair id or -1.
__global__ __launch_bounds__(256) void k_dd_select(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                   const int64_t *__restrict__ attr, int64_t n, int64_t *__restrict__ pair) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    if (q > 0 && key[q - 1] == key[q]) return;
    int64_t e = q;
    int64_t first = -1, best_a = 0, first_a = 0;
    while (e < n && key[e] == key[q]) {
        const int64_t a = attr[val[e]];
        if (dd_marker(a) == 1) {
            if (first < 0) { first = e; first_a = a; best_a = a; }
            else if (dd_left(a) > dd_left(best_a)) best_a = a;
        }
        e++;
    }
    for (int64_t i = q; i < e; i++) {
        const int64_t a = attr[val[i]];
        int64_t out = -1;
        if (dd_marker(a) != 1 && first >= 0) {
            const int64_t ref = i < first ? first_a : best_a;
            const bool hit = dd_left(a) < dd_left(ref) || (dd_left(a) == dd_left(ref) && dd_right(a) > dd_right(ref));
            if (hit) out = (int64_t)(((uint64_t)(uint32_t)dd_right(a) << 32) | (uint64_t)(uint32_t)dd_right(ref));   // :1870-1880
        }
        pair[i] = out;
    }
}

// pair ids >= 0 -> compacted keys for the second sort (the rest are dropped)
__global__ __launch_bounds__(256) void k_dd_compact_pairs(const int64_t *__restrict__ pair, int64_t n, uint64_t *__restrict__ out,
                                                          unsigned long long *__restrict__ cnt) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || pair[q] < 0) return;
    out[atomicAdd(cnt, 1ull)] = (uint64_t)pair[q];
}
// sorted pair ids -> the ids seen at least twice (groupBy().count() >= 2, :186-194)
__global__ __launch_bounds__(256) void k_dd_pair_runs(const uint64_t *__restrict__ key, int64_t n, uint64_t *__restrict__ out,
                                                      unsigned long long *__restrict__ cnt) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    if (q > 0 && key[q - 1] == key[q]) return;
    if (q + 1 < n && key[q + 1] == key[q]) out[atomicAdd(cnt, 1ull)] = key[q];
}

// ---- merge2RCContigs --------------------------------------------------------------------------------------------------
// the 15-mer at p of a contig; past the end the block's 01 terminator reads as one C, then A's (what
// (int)(leftShiftOutFromArray(leftShiftArray(c, p), 15)[0] >>> 2*(32-15)) yields there).  rc: of the reverse complement.
__device__ __forceinline__ uint32_t dd_seed_at(const uint8_t *s, int64_t n, int64_t p, int rc) {
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 15; j++) {
        const int64_t q = p + j;
        uint32_t b;
        if (q < n) b = rc ? 3u - s[n - 1 - q] : s[q];
        else b = q == n ? 1u : 0u;
        x = (x << 2) | b;
    }
    return x;
}
constexpr uint32_t DD_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t dd_hash(uint32_t k) { return (uint32_t)(((uint64_t)k * 0x9E3779B97F4A7C15ull) >> 24); }

__global__ __launch_bounds__(256) void k_dd_seed_insert(const uint8_t *__restrict__ lng, int64_t ln, uint32_t *__restrict__ tkey,
                                                        int32_t *__restrict__ tpos, uint32_t mask) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t * 15;
    if (i > ln) return;
    const uint32_t k = dd_seed_at(lng, ln, i, 0);
    uint32_t h = dd_hash(k) & mask;
    for (;;) {
        const uint32_t old = atomicCAS(&tkey[h], DD_EMPTY, k);
        if (old == DD_EMPTY || old == k) { atomicMax(&tpos[h], (int32_t)(i + 1)); return; }     // HashMap.put: the later position stays
        h = (h + 1) & mask;
    }
}
__global__ __launch_bounds__(256) void k_dd_query(const uint8_t *__restrict__ sh, int64_t sn, int rc, const uint32_t *__restrict__ tkey,
                                                  const int32_t *__restrict__ tpos, uint32_t mask, uint64_t *__restrict__ dist,
                                                  unsigned long long *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sn) return;
    const uint32_t k = dd_seed_at(sh, sn, i, rc);
    uint32_t h = dd_hash(k) & mask;
    for (;;) {
        const uint32_t kk = tkey[h];
        if (kk == DD_EMPTY) return;
        if (kk == k) {
            const int32_t d = (int32_t)(i + 1) - tpos[h];
            dist[atomicAdd(cnt, 1ull)] = (uint64_t)((int64_t)d + 0x80000000ll);       // biased: unsigned order = signed order
            return;
        }
        h = (h + 1) & mask;
    }
}
// the vote over the sorted distances (:1478-1497 with 3 votes, :578-597 / :617-636 with 4): sequential by definition (the
// anchor of a run is the first distance that left the previous run)
__global__ __launch_bounds__(64) void k_dd_vote(const uint64_t *__restrict__ dist, int64_t total, int min_votes, int32_t *__restrict__ out) {
    // one wave: 64 distances per coalesced load, then the scan itself on wave-uniform values (readlane) -- the anchor chain
    // is sequential, the memory latency need not be (a thread walking the list alone paid ~13 ns a distance).  And a run
    // need not be walked at all: the list is sorted, so once an element and the LAST element of its block both lie within
    // one of the anchor, everything between them does, the run's end beyond the block is found by a 64-ary search (the
    // true overlap of two 2.6 Mbp contigs is one run of a million equal distances: 1.9 ms walked, four probes searched), and
    // the vote is won at a known element of it -- the first frequency f with f / total >= 0.3 and f >= min_votes.
    const int lane = threadIdx.x;
    auto val = [&](int64_t i) -> int32_t { return (int32_t)((int64_t)dist[i] - 0x80000000ll); };
    int64_t T = (int64_t)(0.3 * (double)total);
    if (T < 1) T = 1;
    while (T > 1 && (double)(T - 1) / (double)total >= 0.3) T--;
    while ((double)T / (double)total < 0.3) T++;
    if (T < (int64_t)min_votes) T = min_votes;
    int32_t lastDistance = 0, fin = -1;
    int64_t lastFrequency = 0;
    bool done = false;
    int64_t base = 0;
    while (base < total && !done) {
        const int64_t i = base + lane;
        const int32_t v = i < total ? val(i) : 0;
        const int cnt = (int)(total - base < 64 ? total - base : 64);
        const int32_t dl = __builtin_amdgcn_readlane(v, cnt - 1);
        int64_t next = base + cnt;
        for (int j = 0; j < cnt; j++) {
            const int32_t d = __builtin_amdgcn_readlane(v, j);
            if (d - lastDistance >= -1 && d - lastDistance <= 1) {
                if (dl - lastDistance <= 1) {
                    // the run covers the rest of this block; its end e beyond it
                    int64_t lo = base + cnt, hi = cnt == 64 ? total : lo;
                    while (lo < hi) {
                        const int64_t step = (hi - lo + 63) / 64;
                        const int64_t p = lo + (int64_t)lane * step;
                        const bool in_run = p < hi && val(p) - lastDistance <= 1;
                        const int m = (int)__popcll(__ballot(in_run));              // a prefix of the lanes (sorted)
                        if (m == 0) { hi = lo; break; }
                        const int64_t last_in = lo + (int64_t)(m - 1) * step;
                        const int64_t first_out = lo + (int64_t)m * step;
                        lo = last_in + 1;
                        if (m < 64 && first_out < hi) hi = first_out;
                    }
                    const int64_t start = base + j, added = lo - start;
                    if (lastFrequency + added >= T) { fin = val(start + (T - lastFrequency) - 1); done = true; }
                    lastFrequency += added;
                    next = lo;
                    break;
                }
                lastFrequency++;
                if (lastFrequency >= T) { fin = d; done = true; break; }
            } else { lastFrequency = 1; lastDistance = d; }
        }
        base = next;
    }
    if (lane == 0) *out = fin;
}
__global__ __launch_bounds__(256) void k_dd_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t src_n, int64_t from,
                                                 int64_t n, int rc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t q = from + i;                           // position in the (reverse-complemented) source
    dst[i] = rc ? (uint8_t)(3 - src[src_n - 1 - q]) : src[q];
}
__global__ __launch_bounds__(256) void k_dd_fill(uint32_t *p, uint32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Contig { int64_t off, len, id; };      // a contig in a device pool
struct Row { int kind; int64_t id; int64_t idx; int64_t target; };   // kind 0: contig `idx` of the pool; 1: a marker row {-1, target}

struct Dedup {
    rfx_ctx *ctx;
    // scratch that grows with the largest merge
    DevBuf tkey, tpos, dist, dtmp, dval, dvtmp, cnt, fin;
    size_t tcap = 0, dcap = 0;

    int merge_scratch(int64_t ln, int64_t sn) {
        size_t want = 64;
        while (want < (size_t)(ln / 15 + 2) * 2 + 8) want *= 2;
        if (want > tcap) {
            RFX_HIP(tkey.alloc(want * 4, ctx->stream));
            RFX_HIP(tpos.alloc(want * 4, ctx->stream));
            tcap = want;
        }
        const size_t dw = (size_t)sn * 2 + 16;
        if (dw > dcap) {
            RFX_HIP(dist.alloc(dw * 8, ctx->stream)); RFX_HIP(dtmp.alloc(dw * 8, ctx->stream));
            RFX_HIP(dval.alloc(dw * 4, ctx->stream)); RFX_HIP(dvtmp.alloc(dw * 4, ctx->stream));
            dcap = dw;
        }
        if (!cnt.p) { RFX_HIP(cnt.alloc(16, ctx->stream)); RFX_HIP(fin.alloc(16, ctx->stream)); }
        return RFX_OK;
    }

    // one query pass + vote: the distances of `sh` (strand rc) join those already in the list; -> finalDistance
    int query_vote(const uint8_t *sh, int64_t sn, int rc, uint32_t mask, int min_votes, int64_t *n_dist, int32_t *out) {
        hipLaunchKernelGGL(k_dd_query, dim3((unsigned)ceil_div(std::max<int64_t>(sn, 1), 256)), dim3(256), 0, ctx->stream, sh, sn, rc,
                           (const uint32_t *)tkey.as<uint32_t>(), (const int32_t *)tpos.as<int32_t>(), mask, dist.as<uint64_t>(),
                           cnt.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
        unsigned long long c = 0;
        RFX_HIP(hipMemcpyAsync(&c, cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        *n_dist = (int64_t)c;
        // (a second query pass appends to a SORTED prefix: the whole list is sorted again, as Collections.sort does)
        RFX_TRY(sort_pairs(ctx, dist.as<uint64_t>(), dval.as<uint32_t>(), (int64_t)c, 33, dtmp.as<uint64_t>(), dvtmp.as<uint32_t>()));
        hipLaunchKernelGGL(k_dd_vote, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *)dist.as<uint64_t>(), (int64_t)c, min_votes,
                           fin.as<int32_t>());
        RFX_HIP(hipGetLastError());
        RFX_HIP(hipMemcpyAsync(out, fin.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        return RFX_OK;
    }
};

inline void launch_copy(rfx_ctx *ctx, uint8_t *dst, const uint8_t *src, int64_t src_n, int64_t from, int64_t n, int rc) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_dd_copy, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, dst, src, src_n, from, n, rc);
}

}  // namespace

namespace rfx {

// contigs: bases 0..3 on the host, contig i = bases[off[i], off[i+1]).  -> the survivors of round 3 (host), in order.
int dedup_contigs(rfx_ctx *ctx, const uint8_t *h_bases, const int64_t *h_off, int64_t n, std::vector<uint8_t> &out_bases,
                  std::vector<int64_t> &out_off, int64_t *round_n) {
    const int64_t total_in = n ? h_off[n] - h_off[0] : 0;
    // two pools (a round reads one and writes the other) and two work buffers for a contig that grows while shorter ones
    // are merged into it; merges only ever add pieces of their inputs, so nothing outgrows the input
    const size_t pool_cap = (size_t)total_in + 64 * (size_t)(n + 1) + 4096;          // (+ room for the 62-base rows a leftover marker reads as)
    DevBuf poolA, poolB, workA, workB;
    RFX_HIP(poolA.alloc(pool_cap, ctx->stream)); RFX_HIP(poolB.alloc(pool_cap, ctx->stream));
    RFX_HIP(workA.alloc(pool_cap, ctx->stream)); RFX_HIP(workB.alloc(pool_cap, ctx->stream));
    if (total_in > 0) RFX_HIP(hipMemcpyAsync(poolA.p, h_bases + h_off[0], (size_t)total_in, hipMemcpyHostToDevice, ctx->stream));
    std::vector<Contig> cur((size_t)n);
    for (int64_t i = 0; i < n; i++) cur[(size_t)i] = Contig{h_off[i] - h_off[0], h_off[i + 1] - h_off[i], i};
    uint8_t *pin = poolA.as<uint8_t>(), *pout = poolB.as<uint8_t>();
    int64_t in_used = total_in;                           // bytes of the input pool that hold contigs
    Dedup dd; dd.ctx = ctx;

    for (int rnd = 1; rnd <= 3; rnd++) {
        const int both = rnd > 1;
        const int64_t nc = (int64_t)cur.size();
        // ---- markers
        std::vector<int64_t> coff((size_t)nc), clen((size_t)nc), cid((size_t)nc), moff((size_t)nc + 1, 0);
        for (int64_t i = 0; i < nc; i++) {
            coff[(size_t)i] = cur[(size_t)i].off; clen[(size_t)i] = cur[(size_t)i].len; cid[(size_t)i] = cur[(size_t)i].id;
            moff[(size_t)i + 1] = moff[(size_t)i] + dd_seed_count(clen[(size_t)i]) + dd_probe_positions(clen[(size_t)i]) * (both ? 2 : 1);
        }
        const int64_t M = moff[(size_t)nc];
        if (M >= ((int64_t)1 << 32)) { ctx->last_error = "dedup: more than 2^32 marker rows"; return RFX_E_LIMIT; }
        std::vector<uint64_t> cand;                         // pair ids seen at least twice, ascending
        if (M > 0) {
            DevBuf d_meta, key, val, attr, tk, tv, pair, pk, pcnt, ck;
            RFX_HIP(d_meta.alloc((size_t)(4 * nc + 1) * 8, ctx->stream));
            int64_t *dm = d_meta.as<int64_t>();
            RFX_HIP(hipMemcpyAsync(dm, coff.data(), (size_t)nc * 8, hipMemcpyHostToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dm + nc, clen.data(), (size_t)nc * 8, hipMemcpyHostToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dm + 2 * nc, cid.data(), (size_t)nc * 8, hipMemcpyHostToDevice, ctx->stream));
            RFX_HIP(hipMemcpyAsync(dm + 3 * nc, moff.data(), (size_t)(nc + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            RFX_HIP(key.alloc((size_t)M * 8, ctx->stream)); RFX_HIP(tk.alloc((size_t)M * 8, ctx->stream));
            RFX_HIP(val.alloc((size_t)M * 4, ctx->stream)); RFX_HIP(tv.alloc((size_t)M * 4, ctx->stream));
            RFX_HIP(attr.alloc((size_t)M * 8, ctx->stream)); RFX_HIP(pair.alloc((size_t)M * 8, ctx->stream));
            RFX_HIP(pk.alloc((size_t)M * 8, ctx->stream)); RFX_HIP(ck.alloc((size_t)M * 8, ctx->stream));
            RFX_HIP(pcnt.alloc(16, ctx->stream));
            hipLaunchKernelGGL(k_dd_markers, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, (const uint8_t *)pin,
                               (const int64_t *)dm, (const int64_t *)(dm + nc), (const int64_t *)(dm + 2 * nc), (const int64_t *)(dm + 3 * nc), nc,
                               M, both, key.as<uint64_t>(), val.as<uint32_t>(), attr.as<int64_t>());
            RFX_HIP(hipGetLastError());
            RFX_TRY(sort_pairs(ctx, key.as<uint64_t>(), val.as<uint32_t>(), M, 64, tk.as<uint64_t>(), tv.as<uint32_t>()));
            hipLaunchKernelGGL(k_dd_select, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)key.as<uint64_t>(),
                               (const uint32_t *)val.as<uint32_t>(), (const int64_t *)attr.as<int64_t>(), M, pair.as<int64_t>());
            RFX_HIP(hipGetLastError());
            RFX_HIP(hipMemsetAsync(pcnt.p, 0, 16, ctx->stream));
            hipLaunchKernelGGL(k_dd_compact_pairs, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, (const int64_t *)pair.as<int64_t>(),
                               M, pk.as<uint64_t>(), pcnt.as<unsigned long long>());
            RFX_HIP(hipGetLastError());
            unsigned long long np = 0;
            RFX_HIP(hipMemcpyAsync(&np, pcnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            RFX_TRY(sync_checked(ctx));
            if (np > 1) {
                RFX_TRY(sort_pairs(ctx, pk.as<uint64_t>(), val.as<uint32_t>(), (int64_t)np, 64, tk.as<uint64_t>(), tv.as<uint32_t>()));
                RFX_HIP(hipMemsetAsync(pcnt.p, 0, 16, ctx->stream));
                hipLaunchKernelGGL(k_dd_pair_runs, dim3((unsigned)ceil_div((int64_t)np, 256)), dim3(256), 0, ctx->stream,
                                   (const uint64_t *)pk.as<uint64_t>(), (int64_t)np, ck.as<uint64_t>(), pcnt.as<unsigned long long>());
                RFX_HIP(hipGetLastError());
                unsigned long long ncand = 0;
                RFX_HIP(hipMemcpyAsync(&ncand, pcnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                RFX_TRY(sync_checked(ctx));
                cand.resize((size_t)ncand);
                if (ncand) RFX_HIP(hipMemcpyAsync(cand.data(), ck.p, (size_t)ncand * 8, hipMemcpyDeviceToHost, ctx->stream));
                RFX_TRY(sync_checked(ctx));
                std::sort(cand.begin(), cand.end());       // (the kernel appends in any order; groupBy().count() rows ascending)
            }
        }
        // ---- the plan: union, sort("count"), DSShorterRCContigSeqAndTargetExtraction (:3102-3131), sort("count") -- on ids
        std::vector<Row> u;
        for (int64_t i = 0; i < nc; i++) u.push_back(Row{0, cur[(size_t)i].id, i, 0});
        for (uint64_t p : cand) u.push_back(Row{1, (int64_t)(int32_t)(p >> 32), -1, (int64_t)(int32_t)p});      // DSMarkerKmerShorterID :3192-3199
        std::stable_sort(u.begin(), u.end(), [](const Row &a, const Row &b) { return a.id < b.id; });
        std::vector<Row> st;
        const Row *last = nullptr;
        for (size_t q = 0; q < u.size(); q++) {
            const Row *s = &u[q];
            if (!last) { last = s; continue; }
            if (s->id == last->id) {
                if (s->kind == 1) { Row r = *last; r.id = s->target; st.push_back(r); }
                else if (last->kind == 1) { Row r = *s; r.id = last->target; st.push_back(r); }
                last = nullptr;
            } else { st.push_back(*last); last = s; }
        }
        if (last) st.push_back(*last);
        std::stable_sort(st.begin(), st.end(), [](const Row &a, const Row &b) { return a.id < b.id; });
        // a leftover marker row {-1, target} is read as blocks by the removal class: 31 T's and the bases
        // currentKmerSizeFromBinaryBlockArray (:1636-1645) finds in `target`; give it a place in the input pool
        std::vector<Contig> rows;
        size_t extra = 0;
        for (const Row &r : st) {
            if (r.kind == 0) { rows.push_back(Contig{cur[(size_t)r.idx].off, cur[(size_t)r.idx].len, r.id}); continue; }
            const uint64_t t = (uint64_t)r.target;
            const int tz = t ? __builtin_ctzll(t) : 64;
            const int64_t len = std::max<int64_t>(0, 31 + (32 - tz / 2 - 1));
            std::vector<uint8_t> g((size_t)std::max<int64_t>(len, 1));
            for (int64_t i = 0; i < len; i++) g[(size_t)i] = i < 31 ? 3 : (uint8_t)((t >> (2 * (31 - (i - 31)))) & 3);
            const int64_t at = in_used + (int64_t)extra;
            if ((size_t)at + (size_t)len > pool_cap) { ctx->last_error = "dedup: pool exhausted by marker rows"; return RFX_E_LIMIT; }
            if (len) RFX_HIP(hipMemcpy((uint8_t *)pin + at, g.data(), (size_t)len, hipMemcpyHostToDevice));
            extra += (size_t)len;
            rows.push_back(Contig{at, len, r.id});
        }
        // ---- the removal class (:1413-1460 / :516-563): groups of equal id, merged into their longest
        std::vector<Contig> nxt;
        int64_t used = 0;
        auto emit = [&](const uint8_t *src, int64_t len) -> int {
            if ((size_t)(used + len) > pool_cap) { ctx->last_error = "dedup: output pool exhausted"; return RFX_E_LIMIT; }
            if (len) RFX_HIP(hipMemcpyAsync(pout + used, src, (size_t)len, hipMemcpyDeviceToDevice, ctx->stream));
            nxt.push_back(Contig{used, len, (int64_t)nxt.size()});
            used += len;
            return RFX_OK;
        };
        const int variant = rnd == 1 ? 0 : 1;
        size_t g0 = 0;
        while (g0 < rows.size()) {
            size_t g1 = g0 + 1;
            while (g1 < rows.size() && rows[g1].id == rows[g0].id) g1++;
            // the longest of the group (the first of the longest), the others in row order
            size_t li = g0;
            std::vector<size_t> shorts;
            for (size_t q = g0 + 1; q < g1; q++) {
                if (rows[q].len > rows[li].len) { shorts.push_back(li); li = q; } else shorts.push_back(q);
            }
            if (shorts.empty()) { RFX_TRY(emit(pin + rows[li].off, rows[li].len)); g0 = g1; continue; }
            uint8_t *wa = workA.as<uint8_t>(), *wb = workB.as<uint8_t>();
            const uint8_t *lng = pin + rows[li].off;
            int64_t ln = rows[li].len;
            for (size_t si : shorts) {
                const uint8_t *sh = pin + rows[si].off;
                const int64_t sn = rows[si].len;
                RFX_TRY(dd.merge_scratch(ln, sn));
                const uint32_t mask = (uint32_t)dd.tcap - 1;
                hipLaunchKernelGGL(k_dd_fill, dim3((unsigned)ceil_div((int64_t)dd.tcap, 256)), dim3(256), 0, ctx->stream, dd.tkey.as<uint32_t>(),
                                   DD_EMPTY, (int64_t)dd.tcap);
                hipLaunchKernelGGL(k_dd_fill, dim3((unsigned)ceil_div((int64_t)dd.tcap, 256)), dim3(256), 0, ctx->stream, dd.tpos.as<uint32_t>(),
                                   0xFFFFFFFFu, (int64_t)dd.tcap);
                hipLaunchKernelGGL(k_dd_seed_insert, dim3((unsigned)ceil_div(ln / 15 + 1, 256)), dim3(256), 0, ctx->stream, lng, ln,
                                   dd.tkey.as<uint32_t>(), dd.tpos.as<int32_t>(), mask);
                RFX_HIP(hipGetLastError());
                RFX_HIP(hipMemsetAsync(dd.cnt.p, 0, 16, ctx->stream));
                int64_t nd = 0;
                int32_t fd = -1;
                bool done = false;
                uint8_t *dst = lng == wa ? wb : wa;
                if (variant == 1) {                           // the forward strand first (:565-615)
                    RFX_TRY(dd.query_vote(sh, sn, 0, mask, 4, &nd, &fd));
                    if (fd == -1 || fd == 0) {
                    } else if (fd < 0) {
                        int64_t flank = sn - (ln + fd);
                        if (flank > sn) flank = sn;
                        if (flank > 0) {
                            launch_copy(ctx, dst, lng, ln, 0, ln, 0); launch_copy(ctx, dst + ln, sh, sn, sn - flank, flank, 0);
                            lng = dst; ln += flank; done = true;
                        }
                    } else {
                        const int64_t p = std::min<int64_t>(sn, fd);
                        launch_copy(ctx, dst, sh, sn, 0, p, 0); launch_copy(ctx, dst + p, lng, ln, 0, ln, 0);
                        lng = dst; ln += p; done = true;
                    }
                }
                if (!done) {                                  // the reverse complement (:1462-1557 / :616-728)
                    RFX_TRY(dd.query_vote(sh, sn, 1, mask, variant == 0 ? 3 : 4, &nd, &fd));
                    if (fd == -1) { RFX_TRY(emit(sh, sn)); }             // back to the pool, ahead of the long contig
                    else if (fd == 0) {
                    } else if (fd < 0) {
                        int64_t flank = sn - (ln + fd);
                        if (flank > sn) flank = sn;
                        if (flank > 0) {
                            launch_copy(ctx, dst, lng, ln, 0, ln, 0); launch_copy(ctx, dst + ln, sh, sn, sn - flank, flank, 1);
                            lng = dst; ln += flank;
                        }
                    } else {
                        const int64_t p = std::min<int64_t>(sn, fd);
                        launch_copy(ctx, dst, sh, sn, 0, p, 1); launch_copy(ctx, dst + p, lng, ln, 0, ln, 0);
                        lng = dst; ln += p;
                    }
                }
                RFX_HIP(hipGetLastError());
                if ((size_t)ln > pool_cap) { ctx->last_error = "dedup: a merged contig outgrew the pool"; return RFX_E_LIMIT; }
            }
            RFX_TRY(emit(lng, ln));
            g0 = g1;
        }
        RFX_TRY(sync_checked(ctx));
        cur = nxt;                                              // zipWithIndex: ids = positions
        if (round_n) round_n[rnd - 1] = (int64_t)cur.size();
        std::swap(pin, pout);
        in_used = used;
    }
    int64_t tb = 0;
    for (auto &c : cur) tb += c.len;
    out_bases.resize((size_t)tb);
    out_off.assign(cur.size() + 1, 0);
    int64_t p = 0;
    for (size_t i = 0; i < cur.size(); i++) {
        out_off[i] = p;
        if (cur[i].len) RFX_HIP(hipMemcpyAsync(out_bases.data() + p, pin + cur[i].off, (size_t)cur[i].len, hipMemcpyDeviceToHost, ctx->stream));
        p += cur[i].len;
    }
    out_off[cur.size()] = p;
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

}  // namespace rfx

extern "C" {

// TagRowContigDSID.call + changeLine (:3397-3443)
static int64_t dedup_text(const std::vector<uint8_t> &bases, const std::vector<int64_t> &off, int min_contig, char *out, int64_t cap) {
    int64_t pos = 0;
    const int64_t LIM = 10000000;
    const int64_t n = (int64_t)off.size() - 1;
    for (int64_t i = 0; i < n; i++) {
        const int64_t L = off[(size_t)i + 1] - off[(size_t)i];
        if (L < min_contig) continue;
        char hdr[64];
        const int hl = snprintf(hdr, sizeof hdr, ">Contig-%lld-%lld\n", (long long)L, (long long)i);
        for (int j = 0; j < hl; j++) { if (pos < cap) out[pos] = hdr[j]; pos++; }
        for (int64_t j0 = 0; j0 < L; j0 += LIM) {                 // lines of LIM bases
            if (j0 > 0) { if (pos < cap) out[pos] = '\n'; pos++; }
            const int64_t nl = std::min<int64_t>(LIM, L - j0);
            const uint8_t *src = bases.data() + off[(size_t)i] + j0;
            if (pos + nl <= cap) { for (int64_t j = 0; j < nl; j++) out[pos + j] = "ACGT"[src[j]]; }
            else for (int64_t j = 0; j < nl; j++) if (pos + j < cap) out[pos + j] = "ACGT"[src[j]];
            pos += nl;
        }
        if (pos < cap) out[pos] = '\n';
        pos++;
    }
    return pos;
}

int rfx_dedup_contigs(rfx_ctx *ctx, const uint8_t *bases_ascii, const int64_t *contig_off, int64_t n_contigs, int min_contig,
                      uint8_t *out_bases_ascii, int64_t cap_bases, int64_t *out_off, int64_t cap_contigs, int64_t *out_n,
                      char *text, int64_t text_cap, int64_t *text_len, int64_t *round_n) try {
    if (!ctx || !contig_off || n_contigs < 0 || (n_contigs > 0 && !bases_ascii)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    const int64_t nb = n_contigs ? contig_off[n_contigs] - contig_off[0] : 0;
    std::vector<uint8_t> codes((size_t)nb);
    {                                                     // A0 C1 G2, anything else 3 (nucleotideValue :453-465)
        uint8_t lut[256];
        memset(lut, 3, sizeof lut);
        lut[(unsigned char)'A'] = 0; lut[(unsigned char)'C'] = 1; lut[(unsigned char)'G'] = 2;
        const uint8_t *src = bases_ascii + contig_off[0];
        for (int64_t i = 0; i < nb; i++) codes[(size_t)i] = lut[src[i]];
    }
    std::vector<int64_t> off((size_t)n_contigs + 1);
    for (int64_t i = 0; i <= n_contigs; i++) off[(size_t)i] = contig_off[i] - contig_off[0];
    std::vector<uint8_t> ob;
    std::vector<int64_t> oo;
    if (n_contigs == 0) { oo.assign(1, 0); if (round_n) round_n[0] = round_n[1] = round_n[2] = 0; }
    else RFX_TRY(rfx::dedup_contigs(ctx, codes.data(), off.data(), n_contigs, ob, oo, round_n));
    const int64_t m = (int64_t)oo.size() - 1;
    if (out_n) *out_n = m;
    int st = RFX_OK;
    if (out_bases_ascii && out_off) {
        if ((int64_t)ob.size() > cap_bases || m > cap_contigs) st = RFX_E_CAP;
        else {
            for (size_t i = 0; i < ob.size(); i++) out_bases_ascii[i] = (uint8_t)"ACGT"[ob[i]];
            for (int64_t i = 0; i <= m; i++) out_off[i] = oo[(size_t)i];
        }
    }
    if (text_len) {
        *text_len = dedup_text(ob, oo, min_contig, text, text ? text_cap : 0);
        if (text && *text_len > text_cap) st = RFX_E_CAP;
    }
    return st;
} RFX_API_CATCH(ctx)

// The same from the contig TEXT the path writes (">Contig-<len>-...\n" + the sequence wrapped at 100 columns; either twin's
// header): every record is a contig, in order, ids = positions -> the de-duplicated text (TagRowContigDSID's format).
int rfx_dedup_contig_text(rfx_ctx *ctx, const char *contig_text, int64_t len, int min_contig, char *out, int64_t cap, int64_t *out_len,
                          int64_t *out_contigs, int64_t *round_n) try {
    if (!ctx || (len > 0 && !contig_text) || !out_len) return RFX_E_ARG;
    std::vector<uint8_t> bases;
    std::vector<int64_t> off(1, 0);
    bases.reserve((size_t)len);
    int64_t p = 0;
    bool open = false;
    while (p < len) {
        const char *nl = (const char *)memchr(contig_text + p, '\n', (size_t)(len - p));
        const int64_t e = nl ? (int64_t)(nl - contig_text) : len;
        if (e > p && contig_text[p] == '>') {
            if (open) off.push_back((int64_t)bases.size());
            open = true;
        } else if (open) {
            // a line of bases in one piece (9 MB of text a base at a time was most of this call's 20 ms on the host)
            int64_t q = e;
            if (q > p && contig_text[q - 1] == '\r') q--;
            if (memchr(contig_text + p, '\r', (size_t)(q - p)) == nullptr)
                bases.insert(bases.end(), (const uint8_t *)contig_text + p, (const uint8_t *)contig_text + q);
            else
                for (int64_t i = p; i < q; i++) if (contig_text[i] != '\r') bases.push_back((uint8_t)contig_text[i]);
        }
        p = e + 1;
    }
    if (open) off.push_back((int64_t)bases.size());
    const int64_t n = (int64_t)off.size() - 1;
    int64_t m = 0;
    const int st = rfx_dedup_contigs(ctx, bases.data(), off.data(), n, min_contig, nullptr, 0, nullptr, 0, &m, out, cap, out_len, round_n);
    if (out_contigs) *out_contigs = m;
    return st;
} RFX_API_CATCH(ctx)

}  // extern "C"
