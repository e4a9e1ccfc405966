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
// ---- the merges of a round, BATCHED (round 4).  Until round 3 every (short, long) pair was its own launch sequence with two
// host waits (93 us a pair: 4.7 s for the 50,000 pairs of a 100,000-contig set); now step j of a round takes the j-th short
// contig of EVERY group at once: one table region cut into per-merge tables, one seed-insert launch, one query launch whose
// hits carry their merge's number above the distance, ONE sort of all distances, one vote launch (a wave per merge), one
// readback of all votes, one copy launch for all the pieces.  Same arithmetic per merge, same order of the outputs.
struct MergeB {                        // one merge of a batch (device copy)
    const uint8_t *lng, *sh;
    int64_t ln, sn;
    int64_t toff;                      // its table inside the region (slots)
    uint32_t tmask;
    int32_t rc, min_votes, active;
    int64_t spre, qpre;                // first seed thread / first query thread of this merge
};
__device__ __forceinline__ int64_t dd_find(const MergeB *__restrict__ mb, int64_t nm, int64_t t, bool query) {
    int64_t lo = 0, hi = nm;           // the last merge whose first thread is <= t
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((query ? mb[mid].qpre : mb[mid].spre) <= t) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_dd_seed_insert_b(const MergeB *__restrict__ mb, int64_t nm, int64_t total, uint32_t *__restrict__ tkey,
                                                          int32_t *__restrict__ tpos) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const MergeB m = mb[dd_find(mb, nm, t, false)];
    const int64_t i = (t - m.spre) * 15;
    if (i > m.ln) return;
    const uint32_t k = dd_seed_at(m.lng, m.ln, i, 0);
    uint32_t h = dd_hash(k) & m.tmask;
    for (;;) {
        const uint32_t old = atomicCAS(&tkey[m.toff + h], DD_EMPTY, k);
        if (old == DD_EMPTY || old == k) { atomicMax(&tpos[m.toff + h], (int32_t)(i + 1)); return; }
        h = (h + 1) & m.tmask;
    }
}
__global__ __launch_bounds__(256) void k_dd_query_b(const MergeB *__restrict__ mb, int64_t nm, int64_t total, const uint32_t *__restrict__ tkey,
                                                    const int32_t *__restrict__ tpos, uint64_t *__restrict__ dist,
                                                    unsigned long long *__restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t g = dd_find(mb, nm, t, true);
    const MergeB m = mb[g];
    if (!m.active) return;
    const int64_t i = t - m.qpre;
    if (i >= m.sn) return;
    const uint32_t k = dd_seed_at(m.sh, m.sn, i, m.rc);
    uint32_t h = dd_hash(k) & m.tmask;
    for (;;) {
        const uint32_t kk = tkey[m.toff + h];
        if (kk == DD_EMPTY) return;
        if (kk == k) {
            const int32_t d = (int32_t)(i + 1) - tpos[m.toff + h];
            // the merge's number above the biased distance: one sort orders every merge's list
            dist[atomicAdd(cnt, 1ull)] = ((uint64_t)g << 33) | (uint64_t)((int64_t)d + 0x80000000ll);
            return;
        }
        h = (h + 1) & m.tmask;
    }
}
// seg[g] = first entry of merge g in the sorted list (seg[nm] = total)
__global__ __launch_bounds__(256) void k_dd_seg_bounds(const uint64_t *__restrict__ dist, int64_t total, int64_t nm, int64_t *__restrict__ seg) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > nm) return;
    const uint64_t want = (uint64_t)g << 33;
    int64_t lo = 0, hi = total;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (dist[mid] < want) lo = mid + 1; else hi = mid; }
    seg[g] = lo;
}
// one copy launch for all the pieces of a step (or all the contigs of a round's output)
struct CopyB { uint8_t *dst; const uint8_t *src; int64_t src_n, from, n, pre; int32_t rc, pad; };
__global__ __launch_bounds__(256) void k_dd_copy_b(const CopyB *__restrict__ cb, int64_t nc, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = nc;
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (cb[mid].pre <= t) lo = mid; else hi = mid; }
    const CopyB c = cb[lo];
    const int64_t i = t - c.pre;
    if (i >= c.n) return;
    const int64_t q = c.from + i;
    c.dst[i] = c.rc ? (uint8_t)(3 - c.src[c.src_n - 1 - q]) : c.src[q];
}

// the vote over the sorted distances (:1478-1497 with 3 votes, :578-597 / :617-636 with 4): sequential by definition (the
// anchor of a run is the first distance that left the previous run)
__global__ __launch_bounds__(64) void k_dd_vote(const uint64_t *__restrict__ dist_all, const int64_t *__restrict__ seg, const MergeB *__restrict__ mb,
                                                int64_t total_1, int min_votes_1, int32_t *__restrict__ out_all) {
    // batched: block b votes on merge b's slice of the one sorted list (seg; the merge's number sits above bit 33 of every
    // entry); seg == nullptr: the single list of rounds 1-3's form
    const uint64_t *dist = dist_all;
    int64_t total = total_1;
    int min_votes = min_votes_1;
    int32_t *out = out_all;
    if (seg) {
        if (!mb[blockIdx.x].active) return;
        dist = dist_all + seg[blockIdx.x]; total = seg[blockIdx.x + 1] - seg[blockIdx.x]; min_votes = mb[blockIdx.x].min_votes; out = out_all + blockIdx.x;
    }
    // one wave: 64 distances per coalesced load, then the scan itself on wave-uniform values (readlane) -- the anchor chain
    // is sequential, the memory latency need not be (a thread walking the list alone paid ~13 ns a distance).  And a run
    // need not be walked at all: the list is sorted, so once an element and the LAST element of its block both lie within
    // one of the anchor, everything between them does, the run's end beyond the block is found by a 64-ary search (the
    // true overlap of two 2.6 Mbp contigs is one run of a million equal distances: 1.9 ms walked, four probes searched), and
    // the vote is won at a known element of it -- the first frequency f with f / total >= 0.3 and f >= min_votes.
    const int lane = threadIdx.x;
    auto val = [&](int64_t i) -> int32_t { return (int32_t)((int64_t)(dist[i] & 0x1FFFFFFFFull) - 0x80000000ll); };
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
        hipLaunchKernelGGL(k_dd_vote, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *)dist.as<uint64_t>(), (const int64_t *)nullptr,
                           (const MergeB *)nullptr, (int64_t)c, min_votes, fin.as<int32_t>());
        RFX_HIP(hipGetLastError());
        RFX_HIP(hipMemcpyAsync(out, fin.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        return RFX_OK;
    }
};

// ASCII <-> base codes on the device, 16 bytes a thread (round 4: on the host these two loops and the text's "ACGT"[code] were 180 of
// the 223 ms rfx_dedup_contigs spent on 10^5 contigs).  A0 C1 G2, anything else 3 (nucleotideValue :453-465).
__global__ void k_dd_to_codes(uint8_t *__restrict__ b, int64_t n) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int64_t i = i0; i < i0 + 16 && i < n; i++) { const uint8_t c = b[i]; b[i] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
}
__global__ void k_dd_to_ascii(uint8_t *__restrict__ b, int64_t n) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int64_t i = i0; i < i0 + 16 && i < n; i++) b[i] = (uint8_t)"ACGT"[b[i] & 3];
}

inline void launch_copy(rfx_ctx *ctx, uint8_t *dst, const uint8_t *src, int64_t src_n, int64_t from, int64_t n, int rc) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_dd_copy, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, dst, src, src_n, from, n, rc);
}

}  // namespace

namespace rfx {

// contigs: bases on the host -- codes 0..3, or ASCII letters with `ascii` (encoded and decoded on the device) --, contig i =
// bases[off[i], off[i+1]).  -> the survivors of round 3 (host, the same alphabet), in order.
int dedup_contigs(rfx_ctx *ctx, const uint8_t *h_bases, const int64_t *h_off, int64_t n, std::vector<uint8_t> &out_bases,
                  std::vector<int64_t> &out_off, int64_t *round_n, bool ascii) {
    const int64_t total_in = n ? h_off[n] - h_off[0] : 0;
    // two pools (a round reads one and writes the other) and two work buffers for a contig that grows while shorter ones
    // are merged into it; merges only ever add pieces of their inputs, so nothing outgrows the input
    const size_t pool_cap = (size_t)total_in + 64 * (size_t)(n + 1) + 4096;          // (+ room for the 62-base rows a leftover marker reads as)
    DevBuf poolA, poolB, workA, workB;
    RFX_HIP(poolA.alloc(pool_cap, ctx->stream)); RFX_HIP(poolB.alloc(pool_cap, ctx->stream));
    RFX_HIP(workA.alloc(pool_cap, ctx->stream)); RFX_HIP(workB.alloc(pool_cap, ctx->stream));
    if (total_in > 0) RFX_HIP(hipMemcpyAsync(poolA.p, h_bases + h_off[0], (size_t)total_in, hipMemcpyHostToDevice, ctx->stream));
    if (ascii && total_in > 0) {
        hipLaunchKernelGGL(k_dd_to_codes, dim3((unsigned)ceil_div(ceil_div(total_in, 16), 256)), dim3(256), 0, ctx->stream, poolA.as<uint8_t>(), total_in);
        RFX_HIP(hipGetLastError());
    }
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
        // ---- the removal class (:1413-1460 / :516-563): groups of equal id, merged into their longest.  Step j of the round
        // merges the j-th short contig of EVERY group in one batch of launches (the groups are independent; inside a group
        // the shorts meet the growing long contig in row order, as in the reference's loop).
        const int variant = rnd == 1 ? 0 : 1;
        struct Group {
            size_t li; std::vector<size_t> shorts;
            const uint8_t *lng; int64_t ln;                   // the long contig as it stands
            int64_t woff;                                     // the group's place in the two work buffers
            std::vector<size_t> back;                         // shorts that found no place: back to the pool, ahead of the long one
        };
        std::vector<Group> groups;
        int64_t wused = 0;
        size_t max_shorts = 0;
        for (size_t g0 = 0; g0 < rows.size();) {
            size_t g1 = g0 + 1;
            while (g1 < rows.size() && rows[g1].id == rows[g0].id) g1++;
            Group G;
            G.li = g0;
            for (size_t q = g0 + 1; q < g1; q++) {            // the longest of the group (the first of the longest), the others in row order
                if (rows[q].len > rows[G.li].len) { G.shorts.push_back(G.li); G.li = q; } else G.shorts.push_back(q);
            }
            G.lng = pin + rows[G.li].off; G.ln = rows[G.li].len; G.woff = wused;
            if (!G.shorts.empty()) { for (size_t q = g0; q < g1; q++) wused += rows[q].len; }
            max_shorts = std::max(max_shorts, G.shorts.size());
            groups.push_back(std::move(G));
            g0 = g1;
        }
        if ((size_t)wused > pool_cap) { ctx->last_error = "dedup: work area exhausted"; return RFX_E_LIMIT; }
        uint8_t *const wa = workA.as<uint8_t>(), *const wb = workB.as<uint8_t>();
        DevBuf d_mb, d_cb, d_tkey, d_tpos, d_dist, d_dtmp, d_dval, d_dvtmp, d_cnt, d_seg, d_fd;
        RFX_HIP(d_cnt.alloc(16, ctx->stream));
        auto run_copies = [&](std::vector<CopyB> &cb) -> int {
            int64_t pre = 0;
            for (auto &c : cb) { c.pre = pre; pre += c.n; }
            if (cb.empty() || pre == 0) return RFX_OK;
            RFX_HIP(d_cb.alloc(cb.size() * sizeof(CopyB), ctx->stream));
            RFX_HIP(hipMemcpyAsync(d_cb.p, cb.data(), cb.size() * sizeof(CopyB), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_dd_copy_b, dim3((unsigned)ceil_div(pre, 256)), dim3(256), 0, ctx->stream, (const CopyB *)d_cb.as<CopyB>(), (int64_t)cb.size(), pre);
            RFX_HIP(hipGetLastError());
            RFX_TRY(sync_checked(ctx));                       // (`cb` is read by the queued copy until here)
            return RFX_OK;
        };
        for (size_t j = 0; j < max_shorts; j++) {
            std::vector<size_t> gi;                           // groups that have a j-th short
            for (size_t g = 0; g < groups.size(); g++) if (groups[g].shorts.size() > j) gi.push_back(g);
            const int64_t nm = (int64_t)gi.size();
            if (nm >= ((int64_t)1 << 30)) { ctx->last_error = "dedup: more than 2^30 merges in one step"; return RFX_E_LIMIT; }
            std::vector<MergeB> mb((size_t)nm);
            int64_t tslots = 0, sthreads = 0, qthreads = 0;
            for (int64_t m = 0; m < nm; m++) {
                Group &G = groups[gi[(size_t)m]];
                const size_t si = G.shorts[j];
                size_t want = 64;
                while (want < (size_t)(G.ln / 15 + 2) * 2 + 8) want *= 2;
                MergeB &M = mb[(size_t)m];
                M.lng = G.lng; M.ln = G.ln; M.sh = pin + rows[si].off; M.sn = rows[si].len;
                M.toff = tslots; M.tmask = (uint32_t)want - 1;
                M.rc = variant == 1 ? 0 : 1; M.min_votes = variant == 0 ? 3 : 4; M.active = 1;     // variant 1: the forward strand first (:565-615)
                M.spre = sthreads; M.qpre = qthreads;
                tslots += (int64_t)want; sthreads += G.ln / 15 + 1; qthreads += std::max<int64_t>(M.sn, 1);
            }
            if (nm == 0) continue;
            RFX_HIP(d_mb.alloc((size_t)nm * sizeof(MergeB), ctx->stream));
            RFX_HIP(d_tkey.alloc((size_t)tslots * 4, ctx->stream)); RFX_HIP(d_tpos.alloc((size_t)tslots * 4, ctx->stream));
            const size_t dcap = (size_t)qthreads * 2 + 16;    // (two query passes may append)
            RFX_HIP(d_dist.alloc(dcap * 8, ctx->stream)); RFX_HIP(d_dtmp.alloc(dcap * 8, ctx->stream));
            RFX_HIP(d_dval.alloc(dcap * 4, ctx->stream)); RFX_HIP(d_dvtmp.alloc(dcap * 4, ctx->stream));
            RFX_HIP(d_seg.alloc((size_t)(nm + 1) * 8, ctx->stream)); RFX_HIP(d_fd.alloc((size_t)nm * 4, ctx->stream));
            RFX_HIP(hipMemcpyAsync(d_mb.p, mb.data(), (size_t)nm * sizeof(MergeB), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_dd_fill, dim3((unsigned)ceil_div(tslots, 256)), dim3(256), 0, ctx->stream, d_tkey.as<uint32_t>(), DD_EMPTY, tslots);
            hipLaunchKernelGGL(k_dd_fill, dim3((unsigned)ceil_div(tslots, 256)), dim3(256), 0, ctx->stream, d_tpos.as<uint32_t>(), 0xFFFFFFFFu, tslots);
            hipLaunchKernelGGL(k_dd_seed_insert_b, dim3((unsigned)ceil_div(sthreads, 256)), dim3(256), 0, ctx->stream, (const MergeB *)d_mb.as<MergeB>(), nm,
                               sthreads, d_tkey.as<uint32_t>(), d_tpos.as<int32_t>());
            RFX_HIP(hipGetLastError());
            RFX_HIP(hipMemsetAsync(d_cnt.p, 0, 16, ctx->stream));
            int key_bits = 33;
            while (((int64_t)1 << (key_bits - 33)) < nm) key_bits++;
            std::vector<int32_t> fd((size_t)nm, -1);
            // one query pass of the active merges + the vote on every active merge's (grown) list -> fd
            auto query_vote_b = [&]() -> int {
                hipLaunchKernelGGL(k_dd_query_b, dim3((unsigned)ceil_div(qthreads, 256)), dim3(256), 0, ctx->stream, (const MergeB *)d_mb.as<MergeB>(), nm, qthreads,
                                   (const uint32_t *)d_tkey.as<uint32_t>(), (const int32_t *)d_tpos.as<int32_t>(), d_dist.as<uint64_t>(),
                                   d_cnt.as<unsigned long long>());
                RFX_HIP(hipGetLastError());
                unsigned long long c = 0;
                RFX_HIP(hipMemcpyAsync(&c, d_cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                RFX_TRY(sync_checked(ctx));
                // (a second pass appends to a SORTED prefix: the whole list is sorted again, as Collections.sort does)
                RFX_TRY(sort_pairs(ctx, d_dist.as<uint64_t>(), d_dval.as<uint32_t>(), (int64_t)c, key_bits, d_dtmp.as<uint64_t>(), d_dvtmp.as<uint32_t>()));
                hipLaunchKernelGGL(k_dd_seg_bounds, dim3((unsigned)ceil_div(nm + 1, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)d_dist.as<uint64_t>(),
                                   (int64_t)c, nm, d_seg.as<int64_t>());
                hipLaunchKernelGGL(k_dd_vote, dim3((unsigned)nm), dim3(64), 0, ctx->stream, (const uint64_t *)d_dist.as<uint64_t>(),
                                   (const int64_t *)d_seg.as<int64_t>(), (const MergeB *)d_mb.as<MergeB>(), (int64_t)0, 0, d_fd.as<int32_t>());
                RFX_HIP(hipGetLastError());
                std::vector<int32_t> got((size_t)nm);
                RFX_HIP(hipMemcpyAsync(got.data(), d_fd.p, (size_t)nm * 4, hipMemcpyDeviceToHost, ctx->stream));
                RFX_TRY(sync_checked(ctx));
                for (int64_t m = 0; m < nm; m++) if (mb[(size_t)m].active) fd[(size_t)m] = got[(size_t)m];
                return RFX_OK;
            };
            std::vector<CopyB> cb;
            std::vector<char> done((size_t)nm, 0);
            // what a vote means for merge m (rc: the strand the short contig was read in)
            auto apply = [&](int64_t m, int rc, bool last_pass) {
                Group &G = groups[gi[(size_t)m]];
                const size_t si = G.shorts[j];
                const uint8_t *sh = pin + rows[si].off;
                const int64_t sn = rows[si].len, ln = G.ln;
                const int32_t f = fd[(size_t)m];
                uint8_t *dst = (G.lng == wa + G.woff ? wb : wa) + G.woff;
                if (f == -1) { if (last_pass) { G.back.push_back(si); done[(size_t)m] = 1; } return; }      // (:1499-1503: back to the pool)
                if (f == 0) { if (last_pass) done[(size_t)m] = 1; return; }
                if (f < 0) {
                    int64_t flank = sn - (ln + f);
                    if (flank > sn) flank = sn;
                    if (flank > 0) {
                        cb.push_back(CopyB{dst, G.lng, ln, 0, ln, 0, 0, 0}); cb.push_back(CopyB{dst + ln, sh, sn, sn - flank, flank, 0, rc, 0});
                        G.lng = dst; G.ln = ln + flank; done[(size_t)m] = 1;
                    } else if (last_pass) done[(size_t)m] = 1;
                } else {
                    const int64_t p = std::min<int64_t>(sn, f);
                    cb.push_back(CopyB{dst, sh, sn, 0, p, 0, rc, 0}); cb.push_back(CopyB{dst + p, G.lng, ln, 0, ln, 0, 0, 0});
                    G.lng = dst; G.ln = ln + p; done[(size_t)m] = 1;
                }
            };
            RFX_TRY(query_vote_b());
            if (variant == 1) {
                for (int64_t m = 0; m < nm; m++) apply(m, 0, false);
                // the merges the forward strand did not settle go on to the reverse complement (:616-728), their list growing
                bool any = false;
                for (int64_t m = 0; m < nm; m++) { mb[(size_t)m].active = done[(size_t)m] ? 0 : 1; mb[(size_t)m].rc = 1; mb[(size_t)m].min_votes = 4; any = any || !done[(size_t)m]; }
                if (any) {
                    RFX_HIP(hipMemcpyAsync(d_mb.p, mb.data(), (size_t)nm * sizeof(MergeB), hipMemcpyHostToDevice, ctx->stream));
                    RFX_TRY(query_vote_b());
                    for (int64_t m = 0; m < nm; m++) if (mb[(size_t)m].active) apply(m, 1, true);
                }
            } else {
                for (int64_t m = 0; m < nm; m++) apply(m, 1, true);
            }
            for (int64_t m = 0; m < nm; m++) {
                const Group &G = groups[gi[(size_t)m]];
                if ((size_t)(G.woff + G.ln) > pool_cap) { ctx->last_error = "dedup: a merged contig outgrew the pool"; return RFX_E_LIMIT; }
            }
            RFX_TRY(run_copies(cb));
        }
        // ---- the round's output, in the reference's order: per group the shorts that went back to the pool, then the long one
        std::vector<Contig> nxt;
        int64_t used = 0;
        {
            std::vector<CopyB> cb;
            auto emit = [&](const uint8_t *src, int64_t len) -> int {
                if ((size_t)(used + len) > pool_cap) { ctx->last_error = "dedup: output pool exhausted"; return RFX_E_LIMIT; }
                if (len) cb.push_back(CopyB{pout + used, src, len, 0, len, 0, 0, 0});
                nxt.push_back(Contig{used, len, (int64_t)nxt.size()});
                used += len;
                return RFX_OK;
            };
            for (Group &G : groups) {
                for (size_t si : G.back) RFX_TRY(emit(pin + rows[si].off, rows[si].len));
                RFX_TRY(emit(G.lng, G.ln));
            }
            RFX_TRY(run_copies(cb));
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
    bool dense = true;                                        // (a round's output is written back to back, in order)
    for (size_t i = 0; i < cur.size(); i++) { out_off[i] = p; if (cur[i].off != p) dense = false; p += cur[i].len; }
    if (ascii && in_used > 0) {                               // (the whole pool in use: the survivors lie inside it)
        hipLaunchKernelGGL(k_dd_to_ascii, dim3((unsigned)ceil_div(ceil_div(in_used, 16), 256)), dim3(256), 0, ctx->stream, pin, in_used);
        RFX_HIP(hipGetLastError());
    }
    if (dense && tb > 0) RFX_HIP(hipMemcpyAsync(out_bases.data(), pin, (size_t)tb, hipMemcpyDeviceToHost, ctx->stream));
    else
        for (size_t i = 0; i < cur.size(); i++)
            if (cur[i].len) RFX_HIP(hipMemcpyAsync(out_bases.data() + out_off[i], pin + cur[i].off, (size_t)cur[i].len, hipMemcpyDeviceToHost, ctx->stream));
    out_off[cur.size()] = p;
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

}  // namespace rfx

extern "C" {

// TagRowContigDSID.call + changeLine (:3397-3443)
// (`bases`: ASCII letters)
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
            if (pos + nl <= cap) memcpy(out + pos, src, (size_t)nl);
            else if (pos < cap) memcpy(out + pos, src, (size_t)(cap - pos));
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
    std::vector<int64_t> off((size_t)n_contigs + 1);
    for (int64_t i = 0; i <= n_contigs; i++) off[(size_t)i] = contig_off[i] - contig_off[0];
    std::vector<uint8_t> ob;
    std::vector<int64_t> oo;
    if (n_contigs == 0) { oo.assign(1, 0); if (round_n) round_n[0] = round_n[1] = round_n[2] = 0; }
    else RFX_TRY(rfx::dedup_contigs(ctx, bases_ascii + contig_off[0], off.data(), n_contigs, ob, oo, round_n, true));   // (letters in, letters out)
    const int64_t m = (int64_t)oo.size() - 1;
    if (out_n) *out_n = m;
    int st = RFX_OK;
    if (out_bases_ascii && out_off) {
        if ((int64_t)ob.size() > cap_bases || m > cap_contigs) st = RFX_E_CAP;
        else {
            if (!ob.empty()) memcpy(out_bases_ascii, ob.data(), ob.size());
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
