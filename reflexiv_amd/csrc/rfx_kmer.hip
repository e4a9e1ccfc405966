// rfx_kmer.hip -- 2-bit read encoding, canonical k-mer extraction and the
// radix-partitioned count + coverage filter (K1, K2, K3 of SURVEY.md 2.3).
//
// Replaces, for one GPU:
//   ReverseComplementKmerBinaryExtraction.call   P/ReflexivMain.java:3013-3075
//   reduceByKey(KmerCounting)                    P/ReflexivMain.java:155, 2895-2899
//   filter(KmerCoverageFilter)                   P/ReflexivMain.java:160-163, 3115-3119
//
// Data layout in HBM
//   reads   : 2 bits/base, 32 bases per uint64 (first base in the top pair), every read
//             starts on a word boundary (words_per_read words each);
//   k-mers  : one uint64 per instance, partitioned MSD-radix style on a bijective hash of
//             the canonical k-mer: level l writes the instances of each parent bucket into
//             2^bits_l child buckets (exact offsets from a histogram pass), until a bucket
//             holds ~8 K instances;
//   leaves  : one workgroup streams a leaf bucket through an LDS hash table
//             (k-mer -> count), applies min <= count <= max and appends the survivors;
//   output  : survivors sorted ascending by k-mer (rfx_sort.hip), the order contract's
//             count-stage order.
// All of it is HBM-bound integer work: no MFMA.  Extraction never rolls a window: the
// k-mer at base p is a funnel shift of two packed words and its reverse complement is
// ~x bit-reversed with the pair bits swapped, so every instance is independent and
// consecutive lanes read consecutive positions of the same words (coalesced, L1-served).
#include "rfx_internal.h"
#include "rfx_device.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace rfxd;

namespace {

// ------------------------------------------------------------------ encoding

__device__ __forceinline__ uint64_t nuc_code(uint8_t c) {
    // nucleotideValue  P/ReflexivMain.java:3062-3074: A0 C1 G2, anything else 3
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
}

__global__ void k_encode(const uint8_t *__restrict__ bases, const int64_t *__restrict__ read_off,
                         int64_t n_reads, int wpr, uint64_t *__restrict__ words,
                         uint32_t *__restrict__ read_len) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads * wpr) return;
    int64_t r = t / wpr;
    int w = (int)(t % wpr);
    int64_t b0 = read_off[r];
    int64_t len = read_off[r + 1] - b0;
    if (w == 0 && read_len) read_len[r] = (uint32_t)len;
    uint64_t x = 0;
    int64_t s = (int64_t)w * 32;
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        int64_t i = s + j;
        uint64_t v = i < len ? nuc_code(bases[b0 + i]) : 0;
        x = (x << 2) | v;
    }
    words[t] = x;
}

// number of k-mers a read of length len emits (skip rule :3020, loop bounds :3027,:3050)
__device__ __host__ __forceinline__ int64_t nk_of(int64_t len, int k, int fc, int ec) {
    if (len - k - ec <= 1 || fc > len) return 0;
    int64_t m = (len - ec - fc) - (k - 1);
    return m > 0 ? m : 0;
}

__global__ void k_nk_per_read(const int64_t *__restrict__ read_off, int64_t n_reads, int k, int fc,
                              int ec, uint64_t *__restrict__ nk) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) nk[r] = (uint64_t)nk_of(read_off[r + 1] - read_off[r], k, fc, ec);
}

__global__ void k_nk_from_len(const uint32_t *__restrict__ len, int64_t n_reads, int k, int fc, int ec,
                              uint64_t *__restrict__ nk) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) nk[r] = (uint64_t)nk_of((int64_t)len[r], k, fc, ec);
}

// K1 in the reference's emission order: one wave per read, lanes over window positions.
__global__ void k_extract_ordered(const uint64_t *__restrict__ words, int wpr,
                                  const uint64_t *__restrict__ kmer_off, int64_t n_reads, int k,
                                  int fc, uint64_t *__restrict__ out) {
    int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_reads) return;
    int lane = lane_id();
    uint64_t o = kmer_off[r];
    int64_t nk = (int64_t)(kmer_off[r + 1] - o);
    const uint64_t *w = words + r * wpr;
    for (int64_t p = lane; p < nk; p += 64) out[o + p] = canonical(kmer_at(w, fc + (int)p, k), k);
}

// ------------------------------------------------------------ hash, levels

// Hash bits (kmer_hash in rfx_device.h): the top OWNER_BITS pick the owning GPU in the multi-GPU
// path (mulhi(h, n_owners)), the next <= 30 bits are the local radix digits, then 12 bits of
// leaf-table slot and 16 bits for the leaf split fallback.
__device__ __forceinline__ uint64_t local_hash(uint64_t key) { return kmer_hash(key) << OWNER_BITS; }

constexpr int PT = 512;               // threads per workgroup
constexpr int PK = 16;                // instances per thread
constexpr int PTILE = PT * PK;        // 8192 instances per tile
constexpr int MAX_BITS = 10;

struct Level {
    int bits;                 // digit width of this level
    int shift;                // digit = (h >> shift) & (2^bits - 1); child id = h >> shift
    int parent_shift;         // parent id = h >> parent_shift (64 -> id 0)
    int n_owners;             // > 0: the "digit" is the owning rank, mulhi(kmer_hash, n_owners)
    int sub_bits;             // level 1 in one sweep by owner: digit = owner << sub_bits | the header's top sub_bits bits
                              // (bins = 2^bits, bits = ceil(log2(n_owners)) + sub_bits; see bucket_records_by_owner_sweep)
};

__device__ __forceinline__ unsigned digit_of(uint64_t key, const Level &lv) {
    if (lv.n_owners > 0) return (unsigned)__umul64hi(kmer_hash(key), (uint64_t)lv.n_owners);
    return (unsigned)((local_hash(key) >> lv.shift) & ((1u << lv.bits) - 1));
}

// LDS layout shared by the scatter kernels: sorted tile + per-digit bookkeeping.  Sized at launch
// for the actual digit count nb: PTILE*8 + (nb+1)*4 + nb*8 bytes (76 KB at 1024 digits, so two
// workgroups fit a CU's 160 KB).
struct ScatterLds {
    uint64_t *skey;     // PTILE sorted keys
    uint64_t *priv;     // nb running private output cursors of this workgroup
    uint32_t *cnt;      // nb+1: per-digit count, then (in place) local exclusive offset; [nb] = total
};
__device__ __forceinline__ ScatterLds scatter_lds(unsigned char *smem, int nb) {
    ScatterLds l;
    l.skey = reinterpret_cast<uint64_t *>(smem);
    l.priv = l.skey + PTILE;
    l.cnt = reinterpret_cast<uint32_t *>(l.priv + nb);
    return l;
}

// One tile's worth of keys sits in registers with their (rank | digit << 16): turn the
// per-digit counts into local offsets, advance the workgroup's private cursors (its exact output
// ranges come from the per-workgroup histogram rows -- no atomics anywhere), place the keys digit
// by digit in LDS and copy the runs out (consecutive lanes -> consecutive addresses of a run).
// Ends with a barrier after which skey / cnt may be reused.
__device__ __forceinline__ void scatter_tile(const ScatterLds &l, const Level &lv, int nb,
                                             const uint64_t (&key)[PK], const uint32_t (&rank)[PK],
                                             const bool (&ok)[PK], uint64_t *__restrict__ out, uint32_t *wsum) {
    uint32_t total;
    {
        uint32_t c0 = 0, c1 = 0;
        const int d0 = 2 * threadIdx.x, d1 = d0 + 1;
        if (d0 < nb) c0 = l.cnt[d0];
        if (d1 < nb) c1 = l.cnt[d1];
        const uint32_t ex = block_exclusive_scan(c0 + c1, wsum, &total);   // barriers inside
        if (d0 < nb) { l.cnt[d0] = ex; l.priv[d0] += c0; }
        if (d1 < nb) { l.cnt[d1] = ex + c0; l.priv[d1] += c1; }
        if (threadIdx.x == 0) l.cnt[nb] = total;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PK; i++)
        if (ok[i]) l.skey[l.cnt[rank[i] >> 16] + (rank[i] & 0xFFFFu)] = key[i];
    __syncthreads();
    for (int i = threadIdx.x; i < (int)total; i += PT) {
        const uint64_t kk = l.skey[i];
        const unsigned d = digit_of(kk, lv);
        const uint32_t lo = l.cnt[d], hi = l.cnt[d + 1];
        // the run of digit d starts where the cursor stood before this tile
        out[l.priv[d] - (uint64_t)(hi - lo) + (uint64_t)(i - lo)] = kk;
    }
    __syncthreads();
}

// ------------------------------------------------- level 1 straight from reads

// A thread owns PK consecutive windows of ONE read ("segment"): it loads the <= 3 packed words
// they span, builds the first k-mer and its reverse complement with shifts and a bit reversal,
// and rolls both through the remaining windows in registers -- no LDS, no per-base loads.
struct ReadSrc {
    const uint64_t *words;
    int64_t n_reads, n_threads;    // n_threads = n_reads * segs
    int wpr, nk, fc, k, segs;      // words/read, k-mers/read (of the longest read), front clip, k, segments/read
    const uint32_t *len_arr;       // ragged reads: per-read length (nullptr: every read emits nk k-mers)
    int ec;                        // end clip (for the per-read count)
    int wfl;                       // k = 33..63 records: bases of the k-mer on either side of the central window (else 0)
};

// k-mers read r emits
__device__ __forceinline__ int read_nk(const ReadSrc &s, int64_t r) {
    return s.len_arr ? (int)nk_of((int64_t)s.len_arr[r], s.k, s.fc, s.ec) : s.nk;
}

struct SegPos { int64_t r; int sgm; };

template <int SEG = 16>
__device__ __forceinline__ void seg_load(const ReadSrc &s, const SegPos &q, uint64_t (&w)[3]) {
    const uint64_t *g = s.words + q.r * s.wpr;
    const int wi = (s.fc + q.sgm * SEG) >> 5;
    const int last = s.wpr - 1;
    w[0] = g[wi < last ? wi : last];
    w[1] = g[wi + 1 < last ? wi + 1 : last];
    w[2] = g[wi + 2 < last ? wi + 2 : last];
}

// -> number of valid windows; key[i] canonical k-mer of window sgm*PK + i
__device__ __forceinline__ int seg_keys(const ReadSrc &s, int nk_r, int sgm, const uint64_t (&w)[3], uint64_t (&key)[PK]) {
    const int p0 = sgm * PK;
    int v = nk_r - p0;
    v = v > PK ? PK : v;
    const int sh = 2 * ((s.fc + p0) & 31);
    // 64 bases starting at the first window's first base
    const uint64_t hi = sh ? (w[0] << sh) | (w[1] >> (64 - sh)) : w[0];
    const uint64_t lo = sh ? (w[1] << sh) | (w[2] >> (64 - sh)) : w[1];
    const int k2 = 2 * s.k;
    uint64_t fwd = hi >> (64 - k2);
    uint64_t rc = revcomp(fwd, s.k);
    uint64_t rest = (hi << k2) | (lo >> (64 - k2));          // bases k, k+1, ... (k2 in [6, 62])
    const uint64_t mask = low_mask(s.k);
    const int top = k2 - 2;
#pragma unroll
    for (int i = 0; i < PK; i++) {
        key[i] = fwd < rc ? fwd : rc;                          // P/ReflexivMain.java:3051-3055
        const uint64_t b = rest >> 62;                          // roll one base  (:3032-3047)
        rest <<= 2;
        fwd = ((fwd << 2) | b) & mask;
        rc = (rc >> 2) | ((b ^ 3) << top);
    }
    return v;
}

// persistent: every workgroup strides over the tiles with one LDS histogram and stores ITS row
// blockhist[bin * gridDim.x + blockIdx.x]: scanned in that order the table hands every workgroup
// of the scatter kernel (same grid, same tile assignment) exact, private output ranges.
__global__ __launch_bounds__(PT) void k_reads_hist(ReadSrc s, Level lv, uint64_t *__restrict__ blockhist) {
    __shared__ uint32_t h[1 << MAX_BITS];
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * PT;
    const int64_t dq = stride / s.segs;
    const int dr = (int)(stride - dq * s.segs);
    int64_t g = (int64_t)blockIdx.x * PT + threadIdx.x;
    SegPos q;
    q.r = g / s.segs;
    q.sgm = (int)(g - q.r * s.segs);
    for (; g < s.n_threads; g += stride) {
        uint64_t w[3], key[PK];
        seg_load(s, q, w);
        const int v = seg_keys(s, read_nk(s, q.r), q.sgm, w, key);
#pragma unroll
        for (int i = 0; i < PK; i++)
            if (i < v) atomicAdd(&h[digit_of(key[i], lv)], 1u);
        q.r += dq; q.sgm += dr;
        if (q.sgm >= s.segs) { q.sgm -= s.segs; q.r++; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += PT) blockhist[(int64_t)i * gridDim.x + blockIdx.x] = h[i];
}

// bucket offsets of the next level = every bin's first workgroup entry of the scanned table
__global__ void k_bin_offsets(const uint64_t *__restrict__ scanned, int nb, int64_t grid,
                              uint64_t *__restrict__ seg_off) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nb) seg_off[i] = scanned[(int64_t)i * grid];      // entry nb*grid = total
}

// persistent scatter: private cursors in LDS (no atomics); the next tile's words are in flight
// while this tile is ranked, placed and copied out
__global__ __launch_bounds__(PT) void k_reads_scatter(ReadSrc s, Level lv, const uint64_t *__restrict__ scanned,
                                                      uint64_t *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wsum[PT / 64];
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    const ScatterLds l = scatter_lds(smem, nb);
    for (int i = threadIdx.x; i < nb; i += PT) l.priv[i] = scanned[(int64_t)i * gridDim.x + blockIdx.x];
    const int64_t stride = (int64_t)gridDim.x * PT;
    const int64_t dq = stride / s.segs;
    const int dr = (int)(stride - dq * s.segs);
    int64_t g = (int64_t)blockIdx.x * PT + threadIdx.x;
    SegPos q;
    q.r = g / s.segs;
    q.sgm = (int)(g - q.r * s.segs);
    uint64_t w[3] = {0, 0, 0};
    if (g < s.n_threads) seg_load(s, q, w);
    const int64_t tiles = (s.n_threads + PT - 1) / PT;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x, g += stride) {
        for (int i = threadIdx.x; i < nb; i += PT) l.cnt[i] = 0;
        __syncthreads();
        uint64_t key[PK];
        uint32_t rank[PK];
        bool ok[PK];
        const int v = g < s.n_threads ? seg_keys(s, read_nk(s, q.r), q.sgm, w, key) : 0;
        // advance and prefetch the next tile's words
        q.r += dq; q.sgm += dr;
        if (q.sgm >= s.segs) { q.sgm -= s.segs; q.r++; }
        if (g + stride < s.n_threads) seg_load(s, q, w);
#pragma unroll
        for (int i = 0; i < PK; i++) {
            ok[i] = i < v;
            if (ok[i]) {
                const unsigned d = digit_of(key[i], lv);
                rank[i] = atomicAdd(&l.cnt[d], 1u) | (d << 16);     // rank < 8192 fits 16 bits
            }
        }
        __syncthreads();
        scatter_tile(l, lv, nb, key, rank, ok, out, wsum);
    }
}

// ------------------------------------------------- levels >= 2 (and arrays)

// The instances of every parent bucket are cut into "virtual workgroups" of tpb consecutive
// tiles.  Virtual workgroup (p, g) histograms its tiles into row g of parent p's table region
// (entry (d, g) at nb*vb_start[p] + d*G_p + g); one exclusive scan over the table in that order
// yields, for every child bucket (p, d), its start and, inside it, every virtual workgroup's
// private range.  The scatter kernel then needs no atomics and the output is deterministic.
struct VbMap {
    const uint64_t *seg_off;     // nseg+1 element offsets of the parents
    const uint64_t *vb_start;    // nseg+1 exclusive scan of virtual workgroups per parent
    int64_t nseg;
    int tpb;                     // tiles per virtual workgroup
    const uint64_t *seg_end = nullptr;   // parents with gaps between them (the one-sweep level 1): parent p ends at
                                         // seg_end[p], not at seg_off[p + 1]
};

__global__ void k_vb_per_seg(const uint64_t *__restrict__ seg_off, const uint64_t *__restrict__ seg_end, int64_t nseg, int tpb,
                             uint64_t *__restrict__ nvb) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg) {
        const uint64_t per = (uint64_t)tpb * PTILE;
        nvb[s] = ((seg_end ? seg_end[s] : seg_off[s + 1]) - seg_off[s] + per - 1) / per;
    }
}

struct VbPos { int64_t p, g, G; uint64_t begin, end; };

__device__ __forceinline__ bool locate_vb(const VbMap &m, int64_t vb, VbPos *q) {
    if (vb >= (int64_t)m.vb_start[m.nseg]) return false;
    int64_t lo = 0, hi = m.nseg;           // last p with vb_start[p] <= vb
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)m.vb_start[mid] <= vb) lo = mid; else hi = mid;
    }
    q->p = lo;
    q->g = vb - (int64_t)m.vb_start[lo];
    q->G = (int64_t)m.vb_start[lo + 1] - (int64_t)m.vb_start[lo];
    const uint64_t per = (uint64_t)m.tpb * PTILE;
    q->begin = m.seg_off[lo] + (uint64_t)q->g * per;
    const uint64_t e = m.seg_end ? m.seg_end[lo] : m.seg_off[lo + 1];
    q->end = q->begin + per < e ? q->begin + per : e;
    return true;
}

__device__ __forceinline__ void load_striped(const uint64_t *__restrict__ kmers, uint64_t base, uint64_t end,
                                             uint64_t (&key)[PK]) {
#pragma unroll
    for (int i = 0; i < PK; i++) {
        const uint64_t idx = base + (uint64_t)i * PT + threadIdx.x;
        key[i] = idx < end ? kmers[idx] : 0;
    }
}

__global__ __launch_bounds__(PT) void k_vb_hist(const uint64_t *__restrict__ kmers, VbMap m, Level lv,
                                                uint32_t *__restrict__ table) {
    __shared__ uint32_t h[1 << MAX_BITS];
    VbPos q;
    if (!locate_vb(m, blockIdx.x, &q)) return;
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    uint64_t kn[PK];
    load_striped(kmers, q.begin, q.end, kn);
    for (uint64_t base = q.begin; base < q.end; base += PTILE) {
        uint64_t key[PK];
#pragma unroll
        for (int i = 0; i < PK; i++) key[i] = kn[i];
        if (base + PTILE < q.end) load_striped(kmers, base + PTILE, q.end, kn);   // next tile in flight
#pragma unroll
        for (int i = 0; i < PK; i++)
            if (base + (uint64_t)i * PT + threadIdx.x < q.end) atomicAdd(&h[digit_of(key[i], lv)], 1u);
    }
    __syncthreads();
    const int64_t tb = (int64_t)nb * (int64_t)m.vb_start[q.p];
    for (int i = threadIdx.x; i < nb; i += PT) table[tb + (int64_t)i * q.G + q.g] = h[i];
}

// child bucket (p, d) starts at scanned[nb*vb_start[p] + d*G_p]; entry nchild = total
__global__ void k_child_offsets(const uint64_t *__restrict__ scanned, VbMap m, int nb, uint64_t total,
                                uint64_t *__restrict__ child_off) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nchild = m.nseg * nb;
    if (c > nchild) return;
    if (c == nchild) { child_off[c] = total; return; }
    const int64_t p = c / nb;
    const int d = (int)(c - p * nb);
    const int64_t G = (int64_t)m.vb_start[p + 1] - (int64_t)m.vb_start[p];
    // an empty parent has no rows: its children start where the next non-empty region starts
    child_off[c] = scanned[(int64_t)nb * (int64_t)m.vb_start[p] + (int64_t)d * G];
}

__global__ __launch_bounds__(PT) void k_vb_scatter(const uint64_t *__restrict__ kmers, VbMap m, Level lv,
                                                   const uint64_t *__restrict__ scanned,
                                                   uint64_t *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wsum[PT / 64];
    VbPos q;
    if (!locate_vb(m, blockIdx.x, &q)) return;
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    const ScatterLds l = scatter_lds(smem, nb);
    const int64_t tb = (int64_t)nb * (int64_t)m.vb_start[q.p];
    for (int i = threadIdx.x; i < nb; i += PT) l.priv[i] = scanned[tb + (int64_t)i * q.G + q.g];
    uint64_t kn[PK];
    load_striped(kmers, q.begin, q.end, kn);
    for (uint64_t base = q.begin; base < q.end; base += PTILE) {
        for (int i = threadIdx.x; i < nb; i += PT) l.cnt[i] = 0;
        __syncthreads();
        uint64_t key[PK];
        uint32_t rank[PK];
        bool ok[PK];
#pragma unroll
        for (int i = 0; i < PK; i++) { key[i] = kn[i]; ok[i] = base + (uint64_t)i * PT + threadIdx.x < q.end; }
        if (base + PTILE < q.end) load_striped(kmers, base + PTILE, q.end, kn);   // next tile in flight
#pragma unroll
        for (int i = 0; i < PK; i++) {
            if (ok[i]) {
                const unsigned d = digit_of(key[i], lv);
                rank[i] = atomicAdd(&l.cnt[d], 1u) | (d << 16);
            }
        }
        __syncthreads();
        scatter_tile(l, lv, nb, key, rank, ok, out, wsum);
    }
}

// ------------------------------------------------------------------- leaves

#ifndef RFX_LT
#define RFX_LT 768
#endif
#ifndef RFX_LEAF_WAVES_PER_EU
#define RFX_LEAF_WAVES_PER_EU 6
#endif
// 12 waves per workgroup, two workgroups per CU, registers capped for 6 waves per SIMD: the kernel is bound by LDS and
// issue latency (one workgroup per CU instead of two: 13.1 -> 21.7 ms), and 8 waves with 4 per SIMD measured 13.2 ms
// against 12.6 (tools/build_variant.sh + tools/ab_many.sh; 10 waves per workgroup: 20 ms)
constexpr int LT = RFX_LT;              // threads per leaf workgroup
// RFX_LEAF_QUEUE (experiment, off): probing as a per-wave queue of attempts, as in the two-word leaf -- a key whose
// first probe met another key is queued with its next slot, and whenever 32 attempts wait they are made together.
// Measured: 13.3 ms against 12.3 on the bench, 758 ms against 77 on a human-scale share whose tables run at 58 % -- two
// keys per lane already probe side by side here, and a batch of 32 attempts is half a wave.  (An earlier form that
// parked the keys and walked 64 whole sequences at once: 12.7 ms and 298 ms.)  What pays in the two-word leaf, where
// one key per lane went round a ballot loop, does not pay here.
#ifndef RFX_LEAF_QUEUE
#define RFX_LEAF_QUEUE 0
#endif
constexpr int LQCAP = RFX_LEAF_QUEUE ? 96 : 0;    // attempts a wave has pending (record leaves): < 32 + 64 new ones
constexpr int LQBATCH = 32;
// RFX_LEAF_AGG: records of a leaf are first counted in a small LDS table keyed by the WHOLE record (a deep data set
// repeats its super-k-mer records: at 1000x four windows in five lie in a record seen before in the same leaf), and
// only the distinct records are expanded into k-mers, each k-mer inserted once with the record's count as weight.  The
// table is a cache, not a set: a record that finds no slot in RPROBE probes is expanded on the spot with weight 1.
#ifndef RFX_LEAF_AGG
#define RFX_LEAF_AGG 1
#endif
#ifndef RFX_WALK_NOUNROLL
#define RFX_WALK_NOUNROLL 1
#endif
constexpr int RSLOTS = 768;             // record slots (one 64-slot block per wave of the workgroup in the sweep)
constexpr int RPROBE = 4;
constexpr int LOBUF = RFX_LEAF_QUEUE ? 64 : 512;  // survivors k_leaf_count buffers in LDS between flushes (k-mer and pair leaves)
// Record leaves: the record table takes the room of that buffer, so 128 survivors wait in a buffer of their own and
// what a leaf has beyond them goes to the waves' expansion areas, which are idle during the sweep -- and must be free
// again before the next leaf: a sweep that ends with LOBUF1 / 2 or more waiting flushes.
constexpr int LOBUF1 = RFX_LEAF_AGG ? 128 : LOBUF;
constexpr int WSTAGE = 160 + LQCAP + LQCAP / 2;   // u64 words of a wave's private expansion area (record leaves) + its queue
constexpr int LSTAGE = WSTAGE * (LT / 64);
#ifndef RFX_LCAP
#define RFX_LCAP 4096
#endif
constexpr int LCAP = RFX_LCAP;          // hash slots: 4096, or 3072 (three quarters of 12 hash bits; probe steps prime to it)
constexpr int LCAP_BITS = 12;
static_assert(LCAP == 4096 || LCAP == 3072, "slots");
__device__ __forceinline__ uint32_t leaf_slot(uint32_t g) {
    const uint32_t x = g >> (32 - LCAP_BITS);
    return LCAP == 4096 ? x : (x * 3u) >> 2;
}
__device__ __forceinline__ uint32_t leaf_step(uint32_t g) {
    return LCAP == 4096 ? (((g >> 8) & (LCAP - 1)) | 1u) : ((g >> 8) & 511u) * 6u + 1u;      // (6 t + 1 is prime to 2^10 x 3)
}
__device__ __forceinline__ uint32_t leaf_next(uint32_t slot, uint32_t step) {
    if (LCAP == 4096) return (slot + step) & (LCAP - 1);
    const uint32_t x = slot + step;
    return x >= (uint32_t)LCAP ? x - LCAP : x;
}
constexpr int LPROBE = 48;              // a probe sequence this long means the table is too full: split the leaf
constexpr uint64_t EMPTY = ~0ULL;
constexpr int LSTACK = 48;
constexpr int LB = 8;                   // instance loads in flight per lane
constexpr int PBLOCK = 8192;            // pair_out: pairs per output block a workgroup reserves at a time (>= LCAP)
constexpr unsigned long long NOBLK = ~0ULL;

struct CountOut {
    unsigned long long n_out;        // survivors appended (may exceed cap)
    unsigned long long n_distinct;   // distinct k-mers seen
    unsigned long long n_failed;     // leaves that ran out of split depth (must stay 0)
    unsigned long long n_passes;     // table passes run (>= non-empty leaves)
    unsigned long long n_overflow;   // passes abandoned because the table filled up
    unsigned long long n_leaves;     // non-empty leaves
    unsigned long long t_wait, t_all; // RFX_LEAF_DBG & 32: clocks waves spent at the two barriers of a leaf / in the kernel
    // RFX_LEAF_DBG & 128: the record table -- records counted in it / expanded on the spot, occupied slots at the sweeps,
    // windows expanded on the spot / from the table (against the instances: what the table saved)
    unsigned long long r_placed, r_direct, r_slots, w_direct, w_table;
};

// Persistent workgroups each walk a CONTIGUOUS chunk of leaf buckets, i.e. one contiguous
// stream of elements: the next leaf's data is always in flight while the current one is inserted.
// A leaf is streamed through an LDS hash table (k-mer -> count, 64-bit CAS + add); after ONE
// barrier the workgroup sweeps the whole table (filter, reset); survivors collect in an LDS buffer
// and leave with ONE global atomic per flush (a per-leaf atomic on one hot counter serialises the
// whole grid).  A leaf whose table fills up (a probe sequence > LPROBE) is re-streamed in 2, 4, ...
// hash-selected parts; a HEAVY leaf is cut into slices for a second launch (finish_leaves).
// Leaf input element: a k-mer instance (8 B) or a super-k-mer record (16 B, <= 16 instances).
struct alignas(16) Rec { uint64_t w0, w1; };
// Rec: w0 = bases 0..31 of the run's base string, w1 = [63..36] bases 32..45, [35..32] windows-1,
// [31..0] the top 32 bits of the minimiser's local hash (radix digits peel off its top).
__device__ __forceinline__ int rec_len(const Rec &r) { return (int)((r.w1 >> 32) & 15) + 1; }
// WRec: the super-k-mer record of the k = 33..63 path.  b0..b2 = bases 0..95 of the run's base string (the
// run's first k-mer starts at base 0; k + windows - 1 <= 78 bases are used), hd = [35..32] windows-1,
// [31..0] the top 32 bits of the minimiser's local hash.  The minimiser is taken over the CENTRAL 31
// (k odd) or 30 (k even) bases of the k-mer, which the reverse complement maps onto themselves, so it
// is a function of the canonical k-mer as on the k <= 31 path -- and the front end is the k = 31 / 30 one.
struct alignas(32) WRec { uint64_t b0, b1, b2, hd; };
__device__ __forceinline__ int rec_len(const WRec &r) { return (int)((r.hd >> 32) & 15) + 1; }
__device__ __forceinline__ uint32_t rec_hdr(const Rec &r) { return (uint32_t)r.w1; }

// ELEM 0: a k-mer instance (8 B); 1: a super-k-mer record (16 B); 2: a (k-mer, partial count) pair
// (16 B {key, count}: what crosses the multi-GPU exchange after the local combine)
template <int ELEM> struct LeafElem;
template <> struct LeafElem<0> {
    using T = uint64_t;
    static constexpr int PER_LANE = LB;
    __device__ static __forceinline__ T none() { return 0; }
    __device__ static __forceinline__ uint64_t key(const T &e) { return e; }
    __device__ static __forceinline__ uint32_t weight(const T &) { return 1u; }
    template <class F> __device__ static __forceinline__ void for_each_kmer(const T &e, int, F &&f) { f(e); }
};
template <> struct LeafElem<2> {
    using T = Rec;
    static constexpr int PER_LANE = LB / 2;
    __device__ static __forceinline__ T none() { return Rec{0, 0}; }
    __device__ static __forceinline__ uint64_t key(const T &e) { return e.w0; }
    __device__ static __forceinline__ uint32_t weight(const T &e) { return (uint32_t)e.w1; }
};
template <> struct LeafElem<1> {
    using T = Rec;
#ifndef RFX_RPF
#define RFX_RPF 1
#endif
    static constexpr int PER_LANE = RFX_RPF; // 64-record steps a wave holds in registers per leaf
    __device__ static __forceinline__ T none() { return Rec{0, 0}; }
    __device__ static __forceinline__ uint64_t key(const T &e) { return e.w0; }
    __device__ static __forceinline__ uint32_t weight(const T &) { return 1u; }
    // canonical k-mers of the record's windows, rolled in registers (same arithmetic as seg_keys)
    template <class F> __device__ static __forceinline__ void for_each_kmer(const T &e, int k, F &&f) {
        const int n = rec_len(e);
        const int k2 = 2 * k;
        uint64_t fwd = e.w0 >> (64 - k2);
        uint64_t rc = revcomp(fwd, k);
        uint64_t rest = (e.w0 << k2) | (e.w1 >> (64 - k2));
        const uint64_t mask = low_mask(k);
        const int top = k2 - 2;
        for (int i = 0; i < n; i++) {
            f(fwd < rc ? fwd : rc);
            const uint64_t b = rest >> 62;
            rest <<= 2;
            fwd = ((fwd << 2) | b) & mask;
            rc = (rc >> 2) | ((b ^ 3) << top);
        }
    }
};

// inclusive +scan over the 64 lanes of a wave on the DPP path (no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return x;
}

// KC: the k-mer length as a compile-time constant (0 = the run-time argument): the window shifts, the reverse
// complement's alignment and the masks of the record path fold into immediates
template <int ELEM, int KC = 0>
__global__ __launch_bounds__(LT, RFX_LEAF_WAVES_PER_EU) void k_leaf_count(const typename LeafElem<ELEM>::T *__restrict__ keys,
                                                   const uint64_t *__restrict__ leaf_off, const uint64_t *__restrict__ leaf_end, int64_t nleaf,
                                                   const uint64_t *__restrict__ sl_begin, const uint64_t *__restrict__ sl_end,
                                                   uint64_t heavy, uint64_t n_elems, int k_rt,
                                                   int min_cov, int max_cov, int apply_filter,
                                                   uint64_t *__restrict__ out_keys, int32_t *__restrict__ out_counts,
                                                   unsigned long long cap, CountOut *__restrict__ co, int dbg,
                                                   int pair_out, uint32_t presplit) {
    constexpr bool RECS = ELEM == 1;
    const int k = KC ? KC : k_rt;
    __shared__ __attribute__((aligned(16))) unsigned long long tkey[LCAP];
    __shared__ __attribute__((aligned(16))) uint32_t tcnt[LCAP];
    constexpr int OB1 = RECS ? LOBUF1 : LOBUF;                                   // survivors in the buffer of their own
    constexpr int OB2 = RECS && RFX_LEAF_AGG && !RFX_LEAF_QUEUE ? (LSTAGE * 8) / 12 : 0;   // ... in the expansion areas (see LOBUF1)
    constexpr int OBT = OB1 + OB2;
    __shared__ unsigned long long obk[OB1];
    __shared__ int32_t obc[OB1];
    __shared__ uint32_t stackS[LSTACK], stacks[LSTACK];
    __shared__ int sp;
    __shared__ uint32_t overflow, ob_n, ob_lim;
    __shared__ unsigned long long g_emit;
    // pair_out: survivors go straight to blocks of PBLOCK pairs this workgroup takes from the global
    // cursor (one global atomic per 8192 pairs; a per-wave atomic on that one counter costs 30 ns a piece)
    __shared__ unsigned long long blk_base, blk_next;
    __shared__ uint32_t blk_pos, have_next, need_grab;
    __shared__ uint32_t ps_eff;              // elements one table takes (starts at `presplit`, shrinks on overflow)
    __shared__ __attribute__((aligned(16))) uint64_t stage[RECS ? LSTAGE : 2];  // records: per-wave expansion area
    unsigned long long *const obk2 = (unsigned long long *)stage;               // OB2 keys, then OB2 counts
    int32_t *const obc2 = (int32_t *)(obk2 + OB2);
    auto ob_put = [&](uint32_t i, unsigned long long key, int32_t c) __attribute__((always_inline)) {
        if (OB2 == 0 || i < (uint32_t)OB1) { obk[i] = key; obc[i] = c; }
        else { obk2[i - OB1] = key; obc2[i - OB1] = c; }
    };
    constexpr bool AGG = RECS && RFX_LEAF_AGG != 0;
    static_assert(!AGG || RSLOTS == 64 * (LT / 64), "one block of record slots per wave");
    // record table: rA = bases 0..31 of the record, rBC = [63..32] bases 32..45 | windows - 1, [31..0] the count
    __shared__ unsigned long long rA[AGG ? RSLOTS : 1], rBC[AGG ? RSLOTS : 1];
    // whether the table pays is a property of the data set (its depth): the workgroup adds up, over its first leaves, how
    // many records met an entry that was already there, and goes without the table when that is under half of them
    // (measured break-even: a 47x data set, 37 % met one, ran 85 ms with the table and 81 without)
    __shared__ uint32_t agg_on, agg_saved, agg_total;
    uint32_t my_distinct = 0;                                                // every thread
    long long t_wait = 0;
    const long long t_begin = (dbg & 32) ? clock64() : 0;
    unsigned long long my_passes = 0, my_overflows = 0;                      // thread 0 only
    const int lane_ = threadIdx.x & 63;
    const int wave_ = threadIdx.x >> 6;
    constexpr int NW = LT / 64;

    const int64_t l0 = (int64_t)(((unsigned long long)blockIdx.x * (unsigned long long)nleaf) / gridDim.x);
    const int64_t l1 = (int64_t)(((unsigned long long)(blockIdx.x + 1) * (unsigned long long)nleaf) / gridDim.x);
    if (l0 >= l1) return;
    const uint64_t stream_end = n_elems;
    // Leaf l = elements [b, e).  Normal run: consecutive ranges of leaf_off, with HEAVY leaves (more
    // than `heavy` elements: low-complexity sequence piles millions of instances of a few k-mers on one
    // bucket) treated as empty -- they are cut into slices and counted by the whole grid in a second
    // launch of this kernel that takes its ranges from (sl_begin, sl_end) instead.
    auto range = [&](int64_t l, uint64_t &b, uint64_t &e) __attribute__((always_inline)) {
        if (sl_begin) { b = sl_begin[l]; e = sl_end[l]; return; }
        // (leaf_end = leaf_off + 1 when the leaves lie back to back; a level-2 sweep leaves slack between them)
        b = leaf_off[l]; e = leaf_end[l];
        if (heavy && e - b > heavy) e = b;
    };

    for (int i = threadIdx.x; i < LCAP; i += LT) { tkey[i] = EMPTY; tcnt[i] = 0; }
    if constexpr (AGG) for (int i = threadIdx.x; i < RSLOTS; i += LT) { rA[i] = EMPTY; rBC[i] = EMPTY; }
    if (threadIdx.x == 0) {
        ob_n = 0; ob_lim = 0xffffffffu; overflow = 0; sp = 0;
        blk_base = NOBLK; blk_next = NOBLK; blk_pos = PBLOCK; have_next = 0; need_grab = 1;
        ps_eff = presplit ? presplit : 0xffffffffu;
        agg_on = 1; agg_saved = 0; agg_total = 0;
    }

    // flush the survivor buffer (every thread calls)
    auto flush = [&]() {
        __syncthreads();
        const uint32_t cntv = ob_n < ob_lim ? ob_n : ob_lim;
        if (ob_n == 0) return;
        if (threadIdx.x == 0 && cntv) g_emit = atomicAdd(&co->n_out, (unsigned long long)cntv);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cntv; i += LT) {
            const unsigned long long pos = g_emit + i;
            if (pos < cap) {
                const bool own = OB2 == 0 || i < (uint32_t)OB1;
                const unsigned long long key = own ? obk[i] : obk2[i - OB1];
                const int32_t c = own ? obc[i] : obc2[i - OB1];
                if (pair_out) ((Rec *)out_keys)[pos] = Rec{key, (uint64_t)(uint32_t)c};
                else { out_keys[pos] = key; out_counts[pos] = c; }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { ob_n = 0; ob_lim = 0xffffffffu; }
        __syncthreads();
    };

    // prefetched elements.  k-mers: kn[j] = keys[pf + j*LT + tid] (empty beyond the chunk's stream);
    // records: kn[i] = record 64*i + lane of this wave's share of the leaf about to be processed
    using Elem = typename LeafElem<ELEM>::T;
    constexpr int NB = LeafElem<ELEM>::PER_LANE;      // elements in flight per lane
    constexpr int RPF = NB;
    Elem kn[NB];
    uint64_t begin, end;
    range(l0, begin, end);
    uint64_t pf = begin;
    auto prefetch = [&](uint64_t pos) __attribute__((always_inline)) {
        pf = pos;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const uint64_t i = pos + (uint64_t)j * LT + threadIdx.x;
            kn[j] = i < stream_end ? keys[i] : LeafElem<ELEM>::none();
        }
    };
    if constexpr (RECS) {
        const uint64_t b0 = begin, n0 = end - begin;
        const uint64_t ws = b0 + n0 * wave_ / NW, we = b0 + n0 * (wave_ + 1) / NW;
#pragma unroll
        for (int i = 0; i < RPF; i++) {
            const uint64_t r = ws + 64u * i + lane_;
            kn[i] = r < we ? keys[r] : Rec{0, 0};
        }
    } else {
        prefetch(pf);
    }
    __syncthreads();

    // one table pass over the leaf [begin, end): inserts the keys selected by (S, s); no barriers
    auto run_pass = [&](uint32_t S, uint32_t s, bool first, uint64_t begin_next, uint64_t end_next) __attribute__((always_inline)) {
        uint32_t qn = 0;                 // keys parked in this wave's queue (wave-uniform; < 64 between rounds)
        // Two keys per lane probe in lock-step so that two LDS compare-and-swaps are in flight.  A key
        // is done when its slot held EMPTY (claimed) or the key itself; either way its count goes up.
        // There is no occupancy counter: a probe sequence longer than LPROBE flags the pass as
        // overflowing (the table is too full to be worth probing) and the leaf is split.
        // the top n <= LQBATCH entries of the wave's queue: one probe each, what is not finished goes back
        auto attempts = [&](uint32_t n) __attribute__((always_inline)) {
            if constexpr (RECS && LQCAP > 0) {
                uint64_t *wq = stage + wave_ * WSTAGE + 160;
                uint32_t *wqs = (uint32_t *)(wq + LQCAP);
                qn -= n;
                const bool v = (uint32_t)lane_ < n;
                const uint64_t key = v ? wq[qn + lane_] : 0;
                const uint32_t st = v ? wqs[qn + lane_] : 0u;
                __builtin_amdgcn_wave_barrier();
                uint32_t slot = st & 0xffffu, probe = st >> 16;
                bool done = !v;
                if (v) {
                    const unsigned long long p = atomicCAS(&tkey[slot], EMPTY, (unsigned long long)key);
                    if (p == EMPTY || p == key) { atomicAdd(&tcnt[slot], 1u); done = true; }
                    else {
                        const uint32_t g = ((uint32_t)key ^ __builtin_rotateleft32((uint32_t)(key >> 32), 13)) * 0x9E3779B1u;
                        slot = leaf_next(slot, (dbg & 16) ? 1u : leaf_step(g));
                        if (++probe >= (uint32_t)LPROBE) { overflow = 1; done = true; }
                    }
                }
                const uint64_t m = __ballot(!done);
                if (m) {
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (!done) { wq[qn + r] = key; wqs[qn + r] = slot | (probe << 16); }
                    qn += (uint32_t)__popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        auto insert2 = [&](uint64_t keyA, bool a, uint64_t keyB, bool b, uint32_t wA, uint32_t wB) __attribute__((always_inline)) {
            const uint32_t gA = ((uint32_t)keyA ^ __builtin_rotateleft32((uint32_t)(keyA >> 32), 13)) * 0x9E3779B1u;
            const uint32_t gB = ((uint32_t)keyB ^ __builtin_rotateleft32((uint32_t)(keyB >> 32), 13)) * 0x9E3779B1u;
            uint32_t slotA = leaf_slot(gA), slotB = leaf_slot(gB);
            // (the hash-selected part of a split leaf; one part: mask 0 and s = 0, nothing is selected away -- as one AND and one
            // compare per key, where `if (S > 1)` around it became a chain of selects on the wave-uniform condition)
            const uint32_t pmask = (S - 1u) & 0xffffu;
            a = a && ((gA >> 4) & pmask) == s;
            b = b && ((gB >> 4) & pmask) == s;
            // first probe of both keys (nine in ten end here: the key is already in the table)
            unsigned long long pA = EMPTY, pB = EMPTY;
            if (a) pA = atomicCAS(&tkey[slotA], EMPTY, (unsigned long long)keyA);
            if (b) pB = atomicCAS(&tkey[slotB], EMPTY, (unsigned long long)keyB);
            const bool dA = pA == EMPTY || pA == keyA, dB = pB == EMPTY || pB == keyB;
            if (a && dA) atomicAdd(&tcnt[slotA], wA);
            if (b && dB) atomicAdd(&tcnt[slotB], wB);
            // the rest walk their probe sequences one key at a time
            // (double hashing: an odd step from other hash bits -- the wave waits for its longest probe sequence,
            // and linear probing's clusters make that one long in a leaf that fills its table)
            auto walk = [&](uint64_t key, uint32_t slot, uint32_t w, uint32_t g) __attribute__((always_inline)) {
                const uint32_t step = (dbg & 16) ? 1u : leaf_step(g);
                // (rounds 1-2 left this loop to the compiler's full unroll -- 2 % faster then; with the record table in
                // front of it few keys walk at all, and the rolled loop takes the kernel from 69 KB to a third, 1264 spilled
                // SGPRs to 45)
#if RFX_WALK_NOUNROLL
#pragma nounroll
#endif
                for (int probe = 1;; probe++) {
                    slot = leaf_next(slot, step);
                    const unsigned long long p = atomicCAS(&tkey[slot], EMPTY, (unsigned long long)key);
                    if (p == EMPTY || p == key) { atomicAdd(&tcnt[slot], w); break; }
                    if (probe >= LPROBE) { overflow = 1; break; }
                }
            };
            // (one loop per key; a single loop walking a lane's two sequences one after the other measured 10 % slower)
            if constexpr (RECS && LQCAP > 0) {
                uint64_t *wq = stage + wave_ * WSTAGE + 160;
                uint32_t *wqs = (uint32_t *)(wq + LQCAP);
#pragma nounroll
                for (int ph = 0; ph < 2; ph++) {
                    const bool u = ph == 0 ? !dA : !dB;
                    const uint64_t m = __ballot(u);
                    if (m) {
                        const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (u) {
                            const uint32_t g = ph == 0 ? gA : gB;
                            wq[qn + r] = ph == 0 ? keyA : keyB;
                            wqs[qn + r] = leaf_next(ph == 0 ? slotA : slotB, (dbg & 16) ? 1u : leaf_step(g)) | (1u << 16);
                        }
                        qn += (uint32_t)__popcll(m);
                        __builtin_amdgcn_wave_barrier();
#pragma nounroll
                        while (qn >= (uint32_t)LQBATCH) attempts((uint32_t)LQBATCH);
                    }
                }
            } else {
                if (!dA) walk(keyA, slotA, wA, gA);
                if (!dB) walk(keyB, slotB, wB, gB);
            }
        };
        // the attempts still pending when the wave has been through its share of the leaf
        auto drain_queue = [&]() __attribute__((always_inline)) {
            if constexpr (RECS && LQCAP > 0) {
#pragma nounroll
                while (qn > 0u && !__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    attempts(qn < (uint32_t)LQBATCH ? qn : (uint32_t)LQBATCH);
            }
        };
        if constexpr (RECS) {
            // Records: every wave takes an equal contiguous share of the leaf and walks it 64 records
            // at a time.  The first RPF steps of a leaf's first pass come from registers (loaded
            // while the previous leaf was processed); anything else is loaded on the spot.
            const uint64_t n = end - begin;
            const uint64_t ws = begin + n * wave_ / NW, we = begin + n * (wave_ + 1) / NW;
            Rec cur[RPF];
            if (first) {
#pragma unroll
                for (int i = 0; i < RPF; i++) cur[i] = kn[i];
                // this wave's share of the next leaf travels while this leaf is processed
                const uint64_t nn = end_next - begin_next;
                const uint64_t ns = begin_next + nn * wave_ / NW, ne = begin_next + nn * (wave_ + 1) / NW;
#pragma unroll
                for (int i = 0; i < RPF; i++) {
                    const uint64_t r = ns + 64u * i + lane_;
                    kn[i] = r < ne ? keys[r] : Rec{0, 0};
                }
            } else {
#pragma unroll
                for (int i = 0; i < RPF; i++) {
                    const uint64_t r = ws + 64u * i + lane_;
                    cur[i] = r < we ? keys[r] : Rec{0, 0};
                }
            }
            uint32_t *wbits = (uint32_t *)(stage + wave_ * WSTAGE);   // 32 words of head bits
            uint32_t *wcum = wbits + 32;                              // exclusive popcount per word
            uint4 *wrec = (uint4 *)(wbits + 64);                      // the wave's 64 records (see step)
            const int k2 = 2 * k;
            // Records hold 1..16 windows each.  The wave lays its 64 records out in LDS, marks where
            // each record's windows start in the wave's output sequence (one bit per output position)
            // and then every lane extracts k-mers by position: lane j finds its record by a prefix
            // popcount of the head bits.  Balanced lanes; no workgroup barrier (LDS ops of one wave
            // stay in order).  A record lies in LDS as its base string shifted right by ONE bit (x, y, z) with
            // its first output position in w: window i then starts 31 - 2i bits up in (x, y) and (y, z), a
            // shift in 1..31, which is what one 32-bit funnel shift takes (64-bit shifts issue at a fraction
            // of the rate).
            // (weighted: the record's count sits above its first output position -- the parked records of the record table)
            auto kmer_at_pos = [&](uint32_t j, uint32_t &wgt, auto weighted) __attribute__((always_inline)) -> uint64_t {
                const uint32_t wd = wbits[j >> 5];
                const uint32_t r = wcum[j >> 5] + (uint32_t)__popc(wd & (0xffffffffu >> (31 - (j & 31)))) - 1u;
                uint4 rr = wrec[r];
                if constexpr (decltype(weighted)::value) { wgt = rr.w >> 11; rr.w &= 2047u; } else wgt = 1u;
                const uint32_t t = 31u - 2u * (j - rr.w);                        // window index <= 15
                const uint32_t W0 = __builtin_amdgcn_alignbit(rr.x, rr.y, t), W1 = __builtin_amdgcn_alignbit(rr.y, rr.z, t);
                uint32_t fh, fl;
                if constexpr (KC != 0) {
                    constexpr int sft = 64 - 2 * KC;
                    if constexpr (sft >= 32) { fh = 0; fl = W0 >> (sft - 32); }
                    else if constexpr (sft == 0) { fh = W0; fl = W1; }
                    else { fh = W0 >> sft; fl = __builtin_amdgcn_alignbit(W0, W1, sft); }
                } else {
                    const uint64_t f = (((uint64_t)W0 << 32) | W1) >> (64 - k2);
                    fh = (uint32_t)(f >> 32); fl = (uint32_t)f;
                }
                // reverse complement: the halves swap under the 64-bit reversal
                const uint32_t rh = pair_swap32(__brev(~fl)), rl = pair_swap32(__brev(~fh));
                uint32_t ch, cl;
                if constexpr (KC != 0) {
                    constexpr int sft = 64 - 2 * KC;
                    if constexpr (sft >= 32) { ch = 0; cl = rh >> (sft - 32); }
                    else if constexpr (sft == 0) { ch = rh; cl = rl; }
                    else { ch = rh >> sft; cl = __builtin_amdgcn_alignbit(rh, rl, sft); }
                } else {
                    const uint64_t c = (((uint64_t)rh << 32) | rl) >> (64 - k2);
                    ch = (uint32_t)(c >> 32); cl = (uint32_t)c;
                }
                const uint64_t fwd = ((uint64_t)fh << 32) | fl, rc = ((uint64_t)ch << 32) | cl;
                return fwd < rc ? fwd : rc;
            };
            // (weight: how many times the record was seen; lands in the upper bits of the record's first output position)
            auto step = [&](const Rec rcur, const bool valid, const uint32_t /*weight: 1*/) __attribute__((always_inline)) {
                if (dbg & 1) {       // ablation: stream only
                    if (rcur.w0 == 0x123456789ULL) overflow = 1;
                    return;
                }
                const uint32_t nwin = valid ? (uint32_t)rec_len(rcur) : 0u;
                const uint32_t x = wave_incl_scan(nwin);
                const uint32_t off = x - nwin;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                if (lane_ < 32) wbits[lane_] = 0;
                __builtin_amdgcn_wave_barrier();
                if (nwin) atomicOr(&wbits[off >> 5], 1u << (off & 31));
                {
                    const uint32_t s0 = (uint32_t)(rcur.w0 >> 32), s1 = (uint32_t)rcur.w0, s2 = (uint32_t)(rcur.w1 >> 32);
                    // (a record is found by its ordinal among the records that have windows: with the record table the
                    // lanes without any are not only the tail)
                    uint32_t ord = (uint32_t)lane_;
                    if constexpr (AGG) {
                        const uint64_t vm = __ballot(nwin != 0u);
                        ord = __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
                    }
                    if (!AGG || nwin)
                        wrec[ord] = make_uint4(s0 >> 1, __builtin_amdgcn_alignbit(s0, s1, 1), __builtin_amdgcn_alignbit(s1, s2, 1), off);
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const uint32_t c = (uint32_t)__popc(wbits[lane_ & 31]);
                    const uint32_t y = wave_incl_scan(c);          // lanes 0..31 hold the prefix over the 32 words
                    if (lane_ < 32) wcum[lane_] = y - c;
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t wb = 0; wb < total; wb += 128) {
                    const uint32_t j0 = wb + lane_, j1 = j0 + 64;
                    const bool v0 = j0 < total, v1 = j1 < total;
                    uint32_t g0, g1;
                    const uint64_t c0 = kmer_at_pos(v0 ? j0 : 0, g0, std::false_type{}), c1 = kmer_at_pos(v1 ? j1 : 0, g1, std::false_type{});
                    if (dbg & 2) {       // ablation: expand, no table
                        if ((v0 && c0 == 0x123456789ULL) || (v1 && c1 == 0x123456789ULL)) overflow = 1;
                        continue;
                    }
                    insert2(c0, v0, c1, v1, g0, g1);
                }
                __builtin_amdgcn_wave_barrier();
            };
            // With the record table the records to expand come a few per 64-record step (those that found no slot) and
            // from half-empty blocks of slots: they are PARKED in the wave's window until 64 wait, and expanded together.
            uint32_t parked = 0;                                          // wave-uniform
            auto flush_parked = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_wave_barrier();
                const uint32_t w4 = (uint32_t)lane_ < parked ? ((const uint32_t *)&wrec[lane_])[3] : 0u;    // windows | weight << 11
                const uint32_t nwin = w4 & 31u;
                const uint32_t x = wave_incl_scan(nwin);
                const uint32_t off = x - nwin;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                if (lane_ < 32) wbits[lane_] = 0;
                __builtin_amdgcn_wave_barrier();
                if (nwin) { atomicOr(&wbits[off >> 5], 1u << (off & 31)); ((uint32_t *)&wrec[lane_])[3] = off | (w4 & ~2047u); }
                __builtin_amdgcn_wave_barrier();
                {
                    const uint32_t c = (uint32_t)__popc(wbits[lane_ & 31]);
                    const uint32_t y = wave_incl_scan(c);
                    if (lane_ < 32) wcum[lane_] = y - c;
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t wb = 0; wb < total; wb += 128) {
                    const uint32_t j0 = wb + lane_, j1 = j0 + 64;
                    const bool v0 = j0 < total, v1 = j1 < total;
                    uint32_t g0, g1;
                    const uint64_t c0 = kmer_at_pos(v0 ? j0 : 0, g0, std::true_type{}), c1 = kmer_at_pos(v1 ? j1 : 0, g1, std::true_type{});
                    if (dbg & 2) {       // ablation: expand, no table
                        if ((v0 && c0 == 0x123456789ULL) || (v1 && c1 == 0x123456789ULL)) overflow = 1;
                        continue;
                    }
                    insert2(c0, v0, c1, v1, g0, g1);
                }
                __builtin_amdgcn_wave_barrier();
                parked = 0;
            };
            auto park = [&](const Rec rcur, const bool valid, const uint32_t weight) __attribute__((always_inline)) {
                if (dbg & 1) {       // ablation: stream only
                    if (rcur.w0 == 0x123456789ULL) overflow = 1;
                    return;
                }
                const uint64_t vm = __ballot(valid);
                if (!vm) return;
                const uint32_t d = (uint32_t)__popcll(vm);
#pragma nounroll
                while (parked + d > 64u) flush_parked();                  // (once; a loop so that the body is not duplicated)
                if (valid) {
                    const uint32_t ord = parked + __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
                    const uint32_t s0 = (uint32_t)(rcur.w0 >> 32), s1 = (uint32_t)rcur.w0, s2 = (uint32_t)(rcur.w1 >> 32);
                    wrec[ord] = make_uint4(s0 >> 1, __builtin_amdgcn_alignbit(s0, s1, 1), __builtin_amdgcn_alignbit(s1, s2, 1),
                                           (uint32_t)rec_len(rcur) | (weight << 11));
                }
                parked += d;
            };
            // the record table takes the leaf's single pass; the hash-selected parts of a split leaf expand every record
            const bool agg = AGG && S == 1 && !(dbg & 64) && (agg_on || (dbg & 512));
            uint32_t hits = 0;                                            // records of this wave that found a slot (uniform)
            // count the record in the record table; false = no slot for it (or not eligible): expand it now
            auto place = [&](const Rec &r, bool valid) __attribute__((always_inline)) -> bool {
                if constexpr (!AGG) return false;
                if (!agg || !valid || r.w0 == EMPTY) return false;
                // (the bases behind the record's last window are whatever followed in the read: not part of its identity)
                uint32_t b = (uint32_t)(r.w1 >> 32);
                if (!(dbg & 256)) {
                    const uint32_t nw = (b & 15u) + 1u;                       // k - 1 + nw bases are the record's
                    const int used = 2 * (k - 1 + (int)nw) - 64;              // ... of them in b's 28 base bits (<= 0: none)
                    const uint32_t keep = used >= 28 ? 0xfffffff0u : used <= 0 ? 0u : ~(0xffffffffu >> used);
                    b = (b & keep) | (b & 15u);
                }
                const uint32_t h = ((uint32_t)r.w0 ^ __builtin_rotateleft32((uint32_t)(r.w0 >> 32), 13) ^ __builtin_rotateleft32(b, 7)) * 0x9E3779B1u;
                // (admitting a record only at its second sighting -- a bit per record hash -- keeps the records seen once
                // out of the table, 267 instead of 463 of its slots in use, but every record seen again is then expanded
                // twice: 39 % of the windows expanded instead of 36 %)
                uint32_t slot = ((h >> 22) * 3u) >> 2;
#pragma unroll
                for (int probe = 0; probe < RPROBE; probe++) {
                    const unsigned long long p = atomicCAS(&rA[slot], EMPTY, (unsigned long long)r.w0);
                    if (p == EMPTY || p == r.w0) {
                        const unsigned long long q = atomicCAS(&rBC[slot], EMPTY, (unsigned long long)b << 32);
                        if (q == EMPTY || (uint32_t)(q >> 32) == b) { atomicAdd((uint32_t *)&rBC[slot], 1u); return true; }
                    }
                    slot = slot + 1u == (uint32_t)RSLOTS ? 0u : slot + 1u;
                }
                return false;
            };
            auto rstat = [&](bool placed, bool direct, const Rec &r) {
                const uint32_t w = direct ? (uint32_t)rec_len(r) : 0u;
                uint32_t ws_ = w;
                for (int o = 32; o > 0; o >>= 1) ws_ += __shfl_xor(ws_, o, 64);
                const uint64_t mp = __ballot(placed), md = __ballot(direct);
                if (lane_ == 0) {
                    atomicAdd(&co->r_placed, (unsigned long long)__popcll(mp));
                    atomicAdd(&co->r_direct, (unsigned long long)__popcll(md));
                    atomicAdd(&co->w_direct, (unsigned long long)ws_);
                }
            };
            if constexpr (AGG) {
                static_assert(RPF == 1, "the first 64-record step of a wave comes from registers");
                Rec rr = cur[0];
                for (uint64_t r0 = ws; r0 < we; r0 += 64) {
                    // an overflowing pass is abandoned: stop feeding a table that is filling up
                    if (__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    const bool valid = r0 + lane_ < we;
                    // (the wave's next 64 records travel while these are counted)
                    const Rec nx = r0 + 64 + lane_ < we ? keys[r0 + 64 + lane_] : Rec{0, 0};
                    if (agg) {
                        const bool placed = place(rr, valid);
                        hits += (uint32_t)__popcll(__ballot(placed));
                        if (dbg & 128) rstat(placed, valid && !placed, rr);
                        park(rr, valid && !placed, 1u);
                    } else step(rr, valid, 1u);          // (a shallow data set, or a hash-selected part of a split leaf)
                    rr = nx;
                }
                if (agg) {
                    // every record of the leaf has been counted: the distinct ones become weighted k-mers (the wave takes
                    // its block of slots and leaves it empty, whatever became of the pass); what the wave still has
                    // parked goes with them
                    __syncthreads();
                    const uint32_t slot = (uint32_t)(wave_ * 64 + lane_);
                    const unsigned long long a_ = rA[slot], bc = rBC[slot];
                    const bool have = bc != EMPTY;
                    if (a_ != EMPTY) { rA[slot] = EMPTY; rBC[slot] = EMPTY; }
                    {
                        const uint32_t used = (uint32_t)__popcll(__ballot(have));
                        if (lane_ == 0 && hits != used) atomicAdd(&agg_saved, hits - used);      // (modulo 2^32: the sum is >= 0)
                    }
                    if (dbg & 128) {
                        uint32_t w = have ? (uint32_t)((bc >> 32) & 15) + 1u : 0u;
                        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
                        const uint64_t mh = __ballot(have);
                        if (lane_ == 0) { atomicAdd(&co->r_slots, (unsigned long long)__popcll(mh)); atomicAdd(&co->w_table, (unsigned long long)w); }
                    }
                    if (!__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                        park(Rec{(uint64_t)a_, (uint64_t)(bc >> 32) << 32}, have, (uint32_t)bc);
                }
#pragma nounroll
                while (parked) flush_parked();
            } else {
#pragma unroll
                for (int i = 0; i < RPF; i++) {
                    const uint64_t r0 = ws + 64u * i;
                    // an overflowing pass is abandoned: stop feeding a table that is filling up
                    if (r0 < we && !__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) step(cur[i], r0 + lane_ < we, 1u);
                }
                for (uint64_t r0 = ws + 64u * RPF; r0 < we; r0 += 64) {
                    if (__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    const bool valid = r0 + lane_ < we;
                    step(valid ? keys[r0 + lane_] : Rec{0, 0}, valid, 1u);
                }
                drain_queue();
            }
        } else {
            for (uint64_t base = begin; base < end; base += (uint64_t)LT * NB) {
                // an overflowing pass is abandoned: stop feeding a table that is filling up
                if (__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                Elem kc[NB];
                if (pf == base) {
#pragma unroll
                    for (int j = 0; j < NB; j++) kc[j] = kn[j];
                } else {                        // re-streaming a split leaf: load now
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const uint64_t i = base + (uint64_t)j * LT + threadIdx.x;
                        kc[j] = i < end ? keys[i] : LeafElem<ELEM>::none();
                    }
                }
                // next batch of this leaf, or the first batch of the next leaf
                const uint64_t nxt = base + (uint64_t)LT * NB < end ? base + (uint64_t)LT * NB : begin_next;
                if (nxt < stream_end && nxt != pf) prefetch(nxt);
                if (dbg & 1) {       // ablation: stream only
                    if (LeafElem<ELEM>::key(kc[0]) == 0x123456789ULL) overflow = 1;
                    continue;
                }
                static_assert(RECS || NB % 2 == 0, "pairs");
#pragma unroll
                for (int j = 0; j + 1 < NB; j += 2)
                    insert2(LeafElem<ELEM>::key(kc[j]), base + (uint64_t)j * LT + threadIdx.x < end,
                            LeafElem<ELEM>::key(kc[j + 1]), base + (uint64_t)(j + 1) * LT + threadIdx.x < end,
                            LeafElem<ELEM>::weight(kc[j]), LeafElem<ELEM>::weight(kc[j + 1]));
            }
        }
    };

    // Survivors of the pass that just ended -> LDS buffer, by a sweep over the whole table that
    // also resets it.  Thread t owns the 16-byte count chunks t and t + LT (4 slots each) and the
    // key chunks under them.  Call after a barrier; ends with a barrier.  A wave reserves room for
    // its survivors with one LDS add; when the buffer is full the wave writes straight to the
    // output instead (one global add per wave), so no barrier depends on how many keys survive.
    auto emit_pass = [&]() __attribute__((always_inline)) {
        static_assert(LT % 64 == 0 && (LCAP / 4) % 64 == 0, "whole waves sweep whole chunks");
#pragma unroll
        for (int h = 0; h < (LCAP / 4 + LT - 1) / LT; h++) {
            const uint32_t c4 = threadIdx.x + h * LT;                    // chunk of 4 slots
            if (c4 >= (uint32_t)(LCAP / 4)) break;                       // (wave-uniform)
            const uint4 cv = *(const uint4 *)&tcnt[4 * c4];
            const uint32_t cs[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t slot = 4 * c4 + q;
                const int32_t c = (int32_t)cs[q];
                my_distinct += c != 0;
                const bool keep = c != 0 && (!apply_filter || (c >= min_cov && c <= max_cov));
                const uint64_t km = __ballot(keep);
                if (km) {
                    const uint32_t cntw = (uint32_t)__popcll(km);
                    const int leader = __ffsll((unsigned long long)km) - 1;
                    const uint32_t r = (uint32_t)__popcll(km & ((1ULL << lane_) - 1));
                    uint32_t base = 0;
                    if (pair_out) {
                        // room for a whole table is guaranteed (see the leaf loop): positions past the end
                        // of the current block continue in the next one
                        if (lane_ == leader) base = atomicAdd(&blk_pos, cntw);
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader) + r;
                        const unsigned long long pos = base < (uint32_t)PBLOCK ? blk_base + base : blk_next + (base - PBLOCK);
                        if (keep && pos < cap) ((Rec *)out_keys)[pos] = Rec{tkey[slot], (uint64_t)(uint32_t)c};
                        continue;
                    }
                    if (lane_ == leader) base = atomicAdd(&ob_n, cntw);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                    if (base + cntw <= (uint32_t)OBT) {
                        if (keep) ob_put(base + r, tkey[slot], c);
                    } else {
                        uint32_t glo = 0, ghi = 0;
                        if (lane_ == leader) {
                            atomicMin(&ob_lim, base);       // buffer entries at and after `base` are holes
                            const unsigned long long g = atomicAdd(&co->n_out, (unsigned long long)cntw);
                            glo = (uint32_t)g; ghi = (uint32_t)(g >> 32);
                        }
                        glo = (uint32_t)__builtin_amdgcn_readlane((int)glo, leader);
                        ghi = (uint32_t)__builtin_amdgcn_readlane((int)ghi, leader);
                        const unsigned long long pos = (((unsigned long long)ghi << 32) | glo) + r;
                        if (keep && pos < cap) {
                            if (pair_out) ((Rec *)out_keys)[pos] = Rec{tkey[slot], (uint64_t)(uint32_t)c};
                            else { out_keys[pos] = tkey[slot]; out_counts[pos] = c; }
                        }
                    }
                }
            }
            *(uint4 *)&tcnt[4 * c4] = make_uint4(0, 0, 0, 0);
            *(ulonglong2 *)&tkey[4 * c4] = make_ulonglong2(EMPTY, EMPTY);
            *(ulonglong2 *)&tkey[4 * c4 + 2] = make_ulonglong2(EMPTY, EMPTY);
        }
        if (dbg & 32) { const long long t0 = clock64(); __syncthreads(); t_wait += clock64() - t0; }
        else __syncthreads();
        const uint32_t raw = ob_n, lim = ob_lim;      // stable until the next emit_pass
        if (raw >= (uint32_t)OB1 / 2 || lim != 0xffffffffu) flush();
    };

    for (int64_t leaf = l0; leaf < l1; leaf++) {
        // the offset after the next leaf travels while this leaf is processed
        uint64_t begin_next = stream_end, end_next = stream_end;
        if (leaf + 1 < l1) range(leaf + 1, begin_next, end_next);
        // Passes of this leaf.  Normally one: (S, s) = (1, 0), two barriers.  A leaf whose table
        // fills up is re-streamed in 2, 4, ... hash-selected parts off a small stack (rare; the
        // stack is empty between leaves).
        uint32_t S = 1, s = 0;
        bool first = true;
        // a leaf with this many elements will not fit one table: start it in 2, 4, ... hash-selected parts
        // instead of finding that out from an abandoned pass (uniform: the bounds come from leaf_off)
        if (end - begin > (uint64_t)ps_eff) {
            while (end - begin > (uint64_t)ps_eff * S && S < 16) S *= 2;
            __syncthreads();
            if (threadIdx.x == 0) for (uint32_t q = S - 1; q >= 1; q--) { stackS[sp] = S; stacks[sp] = q; sp++; }
            __syncthreads();
        }
        while (true) {
            run_pass(S, s, first, begin_next, end_next);        // (an empty leaf still hands the prefetch chain on)
            first = false;
            if (begin == end) break;
            if (dbg & 32) { const long long t0 = clock64(); __syncthreads(); t_wait += clock64() - t0; }
            else __syncthreads();
            const bool ov = overflow != 0;
            if (threadIdx.x == 0) my_passes++;
            if (!ov) {
                // (need_grab, not blk_pos: waves already in emit_pass move blk_pos while slower ones still
                // decide here -- the decision must be the same in every wave or the barriers go out of step)
                if (pair_out && need_grab) {
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        blk_next = atomicAdd(&co->n_out, (unsigned long long)PBLOCK);
                        have_next = 1; need_grab = 0;
                    }
                    __syncthreads();
                }
                // (before emit_pass, whose closing barrier stands between this store and the next leaf's read of ps_eff:
                // after it, a workgroup whose threads saw different thresholds would disagree on S and on its barriers)
                if (threadIdx.x == 0 && S == 1 && end - begin > (uint64_t)ps_eff * 3 / 4 && ps_eff < (1u << 24)) ps_eff += ps_eff / 64 + 1;
                if constexpr (AGG) {
                    if (threadIdx.x == 0 && S == 1 && agg_on) {
                        agg_total += (uint32_t)(end - begin);
                        if (agg_total >= 8192u) {
                            if (agg_saved * 2u < agg_total) agg_on = 0;
                            agg_total = 0; agg_saved = 0;
                        }
                    }
                }
                emit_pass();
                if (pair_out && threadIdx.x == 0) {         // (all emission done; next read: after a later barrier)
                    if (blk_pos >= (uint32_t)PBLOCK) { blk_base = blk_next; blk_pos -= PBLOCK; have_next = 0; }
                    need_grab = blk_pos + (uint32_t)LCAP > (uint32_t)PBLOCK && !have_next;
                }
                if (S == 1) break;
            } else {
                // wipe the abandoned table, push the two halves of (S, s)
                for (int i = threadIdx.x; i < LCAP; i += LT) { tkey[i] = EMPTY; tcnt[i] = 0; }
                __syncthreads();                    // everyone has read `overflow`
                if (threadIdx.x == 0) {
                    my_overflows++;
                    overflow = 0;
                    // what one table took too much of: later leaves of this workgroup start in parts sooner
                    // (the distinct-per-element ratio is a property of the data set: coverage, error rate)
                    // (additive increase / multiplicative decrease: one odd leaf must not decide for all)
                    if ((end - begin) / S < 2ull * ps_eff && ps_eff > 64) ps_eff -= ps_eff / 8;
                    if (sp + 2 <= LSTACK && S < (1u << 16)) {
                        stackS[sp] = 2 * S; stacks[sp] = s + S; sp++;
                        stackS[sp] = 2 * S; stacks[sp] = s;     sp++;
                    } else {
                        atomicAdd(&co->n_failed, 1ULL);
                    }
                }
                __syncthreads();
            }
            if (sp == 0) break;                     // uniform: sp only changes between barriers
            S = stackS[sp - 1]; s = stacks[sp - 1];
            __syncthreads();
            if (threadIdx.x == 0) sp--;
        }
        begin = begin_next;
        end = end_next;
    }
    flush();
    if (pair_out) {
        // the unused tail of this workgroup's last block(s) reads as holes (count 0)
        if (blk_base != NOBLK)
            for (uint32_t i = blk_pos + threadIdx.x; i < (uint32_t)PBLOCK; i += LT)
                if (blk_base + i < cap) ((Rec *)out_keys)[blk_base + i] = Rec{0, 0};
        if (have_next)
            for (uint32_t i = threadIdx.x; i < (uint32_t)PBLOCK; i += LT)
                if (blk_next + i < cap) ((Rec *)out_keys)[blk_next + i] = Rec{0, 0};
    }
    {
        // distinct keys: wave sums, one global add per wave
        uint32_t d = my_distinct;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane_ == 0 && d) atomicAdd(&co->n_distinct, (unsigned long long)d);
    }
    if (threadIdx.x == 0) {
        atomicAdd(&co->n_passes, my_passes);
        if (my_overflows) atomicAdd(&co->n_overflow, my_overflows);
    }
    if ((dbg & 32) && lane_ == 0) {
        atomicAdd(&co->t_wait, (unsigned long long)t_wait);
        atomicAdd(&co->t_all, (unsigned long long)(clock64() - t_begin));
    }
}

// ---- leaves of the k > 32 path: two-word keys.  No 128-bit LDS atomic exists, so a slot is
// claimed through its count word: 0 = empty, WLOCK = being written, otherwise the count.  The
// claimer writes both key words and then publishes count 1; a lane that meets WLOCK tries the same
// slot again on the next trip round a loop that the WHOLE WAVE leaves together (ballot) -- so the
// publish of a neighbouring lane can never sit on an exit path the spinning lane keeps it from
// reaching -- and a lane that meets a count compares the (now immutable) key words.  Same structure otherwise: persistent workgroups over
// contiguous leaf chunks, two barriers per leaf, table sweep, split on overflow.
#ifndef RFX_WCAP_BITS
#define RFX_WCAP_BITS 12
#endif
#ifndef RFX_WLT
#define RFX_WLT 1024
#endif
#ifndef RFX_WCAP_X3
#define RFX_WCAP_X3 0
#endif
constexpr int WCAP_BITS = RFX_WCAP_BITS;
constexpr int WCAP = RFX_WCAP_X3 ? 3 << (WCAP_BITS - 1) : 1 << WCAP_BITS;     // 2^bits slots, or three halves of that
__device__ __forceinline__ uint32_t wide_slot(uint32_t g) {
    return RFX_WCAP_X3 ? ((g >> (31 - WCAP_BITS)) * 3u) >> 2 : g >> (32 - WCAP_BITS);
}
__device__ __forceinline__ uint32_t wide_step(uint32_t g) {          // prime to the slot count
    return RFX_WCAP_X3 ? ((g >> 8) & ((1u << (WCAP_BITS - 2)) - 1u)) * 6u + 1u : (((g >> 8) & (WCAP - 1)) | 1u);
}
__device__ __forceinline__ uint32_t wide_next(uint32_t slot, uint32_t step) {
    if (!RFX_WCAP_X3) return (slot + step) & (WCAP - 1);
    const uint32_t x = slot + step;
    return x >= (uint32_t)WCAP ? x - WCAP : x;
}
// 1024 threads, 4096 slots, a queue of probe attempts per wave (RFX_WIDE_QUEUE) and a 64-entry survivor buffer -- all the
// LDS there is: 31.1 ms at k = 63 (768 threads: 34.0, 896: 33.1).  Measured on the
// way: 1024 threads and 4096 slots 44.2 ms, 768 threads and 6144 slots (RFX_WCAP_X3: the site-laden leaves split less
// often) 39.5 ms, two workgroups of 2048 slots per CU 55 ms, 512 threads with the queue 41.5 ms
constexpr int WLT = RFX_WLT;            // threads per workgroup
#ifndef RFX_WOBUF
#define RFX_WOBUF 64
#endif
constexpr int WOBUF = RFX_WOBUF;        // survivors buffered in LDS between flushes
constexpr uint32_t WLOCK = 0xFFFFFFFFu;

// canonical two-word k-mer (counter layout: word0 = bases 0..31, word1 = the last t = k - 32 bases,
// right-aligned) of window j of a record's base string -- compareLongArrayBlocks' order
// (P/ReflexivDataFrameCounter64.java:652-687): smaller (word0, word1) wins, ties keep the forward strand
__device__ __forceinline__ void wrec_kmer(const WRec &r, uint32_t j, int t, uint64_t *k0, uint64_t *k1) {
    const uint32_t sh = 2u * j;                                  // <= 30
    // ((x >> 1) >> (63 - sh): the funnel shift that is also right at sh = 0, without a compare and two selects per word)
    const uint64_t f0 = (r.b0 << sh) | ((r.b1 >> 1) >> (63u - sh));
    const uint64_t x1 = (r.b1 << sh) | ((r.b2 >> 1) >> (63u - sh));
    const int t2 = 2 * t;                                        // 2..62
    const uint64_t f1 = x1 >> (64 - t2);
    const uint64_t last32 = (f0 << t2) | f1;                     // the k-mer's last 32 bases
    const uint64_t r0 = revcomp(last32, 32);
    const uint64_t r1 = revcomp(f0, 32) & ((1ULL << t2) - 1);    // reverse complement of its first t bases
    const bool fwd = f0 < r0 || (f0 == r0 && f1 <= r1);
    *k0 = fwd ? f0 : r0;
    *k1 = fwd ? f1 : r1;
}

// RFX_WIDE_LOCKFREE (round 3): the two key words of a slot are claimed by two 64-bit compare-and-swaps, one after the
// other -- word 0 first; a key that meets its own word 0 (or an empty one) goes on to word 1, one that meets another
// value in either moves to its next slot -- so whatever order the claims land in a slot names ONE key, nothing is ever
// locked, nobody spins, and a compare-and-swap's return value IS the key word (no 16-byte read to compare).  EMPTY = all
// ones in both words: word 1 holds k - 32 <= 31 bases, and word 0 of a CANONICAL k-mer is never 32 T's (its reverse
// complement would begin with at most 31 T's and an A, and be the smaller one).  With no lock there is no loop the wave
// must leave together and no need for the queue of attempts that replaced it.
#ifndef RFX_WIDE_LOCKFREE
#define RFX_WIDE_LOCKFREE 1
#endif
#ifndef RFX_WIDE_QUEUE
#define RFX_WIDE_QUEUE (RFX_WIDE_LOCKFREE ? 0 : 1)
#endif
constexpr int WQCAP = RFX_WIDE_QUEUE ? 128 : 0;   // probe attempts a wave has pending (record leaves)
// RFX_WIDE_AGG: the record table of the two-word leaf (see RFX_LEAF_AGG): a slot is (bases 0..31, bases 32..63, the
// rest of the record's own bases | windows - 1 | count), claimed field by field -- a record that meets a slot whose
// fields are not all its own moves on, so whatever order the claims land in, a slot names one record.
#ifndef RFX_WIDE_AGG
#define RFX_WIDE_AGG 1
#endif
#ifndef RFX_WRSLOTS
#define RFX_WRSLOTS 1024
#endif
constexpr int WRSLOTS = RFX_WRSLOTS;    // record slots: 1024 (every wave sweeps a block of 64; 768: 26.9 ms instead of 26.1 at k = 63), 768 or 512
constexpr int WRMAX = 8191;             // records of a leaf that goes through the table (a weight takes 13 bits of a queue entry)
#ifndef RFX_WPARK
#define RFX_WPARK (RFX_WIDE_AGG && RFX_WIDE_QUEUE ? 32 : 64)
#endif
constexpr int WPARK = RFX_WPARK;        // records a wave parks before it expands them
constexpr int WWS0 = WPARK * 4 + 32;    // u64 words of a wave's expansion area: the parked records + head bits + prefix counts
constexpr int WWS = WWS0 + WQCAP * 2 + WQCAP / 2;   // ... + the queue: 16-byte keys, then 4-byte (slot | probes << 16)

// RECS: the leaf's elements are super-k-mer records (WRec) expanded here, k-mer by k-mer, balanced over the
// lanes as in k_leaf_count<1>; else two-word k-mers (Rec = {word0, word1}).
template <bool RECS>
__global__ __launch_bounds__(WLT) void k_leaf_count_wide(const std::conditional_t<RECS, WRec, Rec> *__restrict__ elems,
                                                        const uint64_t *__restrict__ leaf_off, const uint64_t *__restrict__ leaf_end,
                                                        int64_t nleaf, int k, int min_cov, int max_cov,
                                                        uint64_t *__restrict__ out_keys, int64_t *__restrict__ out_counts,
                                                        unsigned long long cap, CountOut *__restrict__ co, uint32_t presplit) {
    __shared__ __attribute__((aligned(32))) uint64_t wstage[RECS ? WWS * (WLT / 64) : 4];
    const bool dh = !(presplit & 0x20000000u);
    __shared__ uint32_t ps_eff;              // records one table takes (starts at `presplit`, shrinks on overflow)
    constexpr bool LF = RFX_WIDE_LOCKFREE != 0;
    __shared__ __attribute__((aligned(16))) ulonglong2 tk[LF ? 1 : WCAP];     // (lock form) both key words of a slot in one 16-byte LDS access
    __shared__ unsigned long long tk0[LF ? WCAP : 1], tk1[LF ? WCAP : 1];       // (lock-free form) a plane per key word
    __shared__ uint32_t tcnt[WCAP];
    constexpr bool WAGG = RECS && RFX_WIDE_AGG != 0;
    static_assert(!WAGG || ((WQCAP > 0 || LF) && WRSLOTS % 64 == 0 && WRSLOTS <= WLT), "record table");
    __shared__ unsigned long long rA[WAGG ? WRSLOTS : 1], rB[WAGG ? WRSLOTS : 1], rCC[WAGG ? WRSLOTS : 1];
    __shared__ uint32_t agg_on, agg_saved, agg_total;       // (as in k_leaf_count)
    __shared__ unsigned long long obh[WOBUF], obl[WOBUF];
    __shared__ uint32_t obc[WOBUF];
    __shared__ uint32_t stackS[LSTACK], stacks[LSTACK];
    __shared__ int sp;
    __shared__ uint32_t overflow, ob_n, ob_lim;
    __shared__ unsigned long long g_emit;
    uint32_t my_distinct = 0;
    unsigned long long my_passes = 0, my_overflows = 0;
    const int lane_ = threadIdx.x & 63;
    const int64_t l0 = (int64_t)(((unsigned long long)blockIdx.x * (unsigned long long)nleaf) / gridDim.x);
    const int64_t l1 = (int64_t)(((unsigned long long)(blockIdx.x + 1) * (unsigned long long)nleaf) / gridDim.x);
    if (l0 >= l1) return;
    for (int i = threadIdx.x; i < WCAP; i += WLT) tcnt[i] = 0;
    if constexpr (LF) for (int i = threadIdx.x; i < WCAP; i += WLT) { tk0[i] = EMPTY; tk1[i] = EMPTY; }
    if constexpr (WAGG) for (int i = threadIdx.x; i < WRSLOTS; i += WLT) { rA[i] = EMPTY; rB[i] = EMPTY; rCC[i] = EMPTY; }
    if (threadIdx.x == 0) {
        ob_n = 0; ob_lim = 0xffffffffu; overflow = 0; sp = 0;
        ps_eff = (presplit & 0xffffffu) ? (presplit & 0xffffffu) : 0xffffffffu;
        agg_on = 1; agg_saved = 0; agg_total = 0;
    }
    __syncthreads();

    auto flush = [&]() {
        __syncthreads();
        const uint32_t cntv = ob_n < ob_lim ? ob_n : ob_lim;
        if (ob_n == 0) return;
        if (threadIdx.x == 0 && cntv) g_emit = atomicAdd(&co->n_out, (unsigned long long)cntv);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cntv; i += WLT) {
            const unsigned long long pos = g_emit + i;
            if (pos < cap) { out_keys[2 * pos] = obh[i]; out_keys[2 * pos + 1] = obl[i]; out_counts[pos] = (int64_t)obc[i]; }
        }
        __syncthreads();
        if (threadIdx.x == 0) { ob_n = 0; ob_lim = 0xffffffffu; }
        __syncthreads();
    };

    // (records: the wave's first 64 records of a leaf are loaded while the leaf before it is counted -- 2048 leaves per
    // workgroup, and after every one of them all sixteen waves stood waiting for memory at the same moment)
    using ElemT = std::conditional_t<RECS, WRec, Rec>;
    ElemT pre{};
    bool first_pass = true;
    auto wave_first = [&](uint64_t b, uint64_t e) __attribute__((always_inline)) -> ElemT {
        constexpr int NWV0 = WLT / 64;
        const uint64_t n0 = e - b, w0 = b + n0 * (threadIdx.x >> 6) / NWV0, w1 = b + n0 * ((threadIdx.x >> 6) + 1) / NWV0;
        return w0 + lane_ < w1 ? elems[w0 + lane_] : ElemT{};
    };
    if constexpr (RECS) pre = wave_first(leaf_off[l0], leaf_end[l0]);       // (leaf_end = leaf_off + 1 unless the last level left slack)
    for (int64_t leaf = l0; leaf < l1; leaf++) {
        const uint64_t begin = leaf_off[leaf], end = leaf_end[leaf];
        uint32_t S = 1, s = 0;
        first_pass = true;
        if constexpr (RECS) {
            // a leaf with many records will not fit one table: start it in 2, 4, ... hash-selected parts
            // instead of finding that out from an abandoned pass (presplit = records one table takes)
            if (end - begin > (uint64_t)ps_eff) {
                while ((end - begin) > (uint64_t)ps_eff * S && S < 16) S *= 2;
                if (S > 1) {
                    __syncthreads();
                    if (threadIdx.x == 0) for (uint32_t q = S - 1; q >= 1; q--) { stackS[sp] = S; stacks[sp] = q; sp++; }
                    __syncthreads();
                }
            }
        }
        while (begin != end) {
            // one pass: the keys selected by (S, s) go into the table
            // one key into the table.  The first probe is straight-line code (nine in ten meet the key itself);
            // the rest runs in a loop the WHOLE WAVE leaves together (ballot): a lane that leaves a loop early
            // waits at the exit for the others, and code on an exit path -- a `break` after the publish -- runs
            // only then, so a lane spinning on a lock its neighbour holds would spin for ever.  Here every
            // publish sits inside an iteration every lane completes, and the claim of a first probe is
            // published before anyone waits.
            // lock-free form: the first probe straight-line (most keys are in the table already), the rest in a loop of the
            // lane's own
            auto insertw = [&](const uint64_t w0, const uint64_t w1, bool v, const uint32_t wgt) __attribute__((always_inline)) {
                if constexpr (LF) {
                    const uint32_t g = ((uint32_t)w0 ^ __builtin_rotateleft32((uint32_t)(w0 >> 32), 13) ^
                                        ((uint32_t)w1 * 0x85EBCA6Bu) ^ (uint32_t)(w1 >> 32)) * 0x9E3779B1u;
                    v = v && ((g >> 4) & ((S - 1u) & 0xffffu)) == s;       // (one part: mask 0, s = 0)
                    uint32_t slot = wide_slot(g);
                    bool done = !v;
                    if (v) {
                        const unsigned long long p0 = atomicCAS(&tk0[slot], EMPTY, (unsigned long long)w0);
                        if (p0 == EMPTY || p0 == w0) {
                            const unsigned long long p1 = atomicCAS(&tk1[slot], EMPTY, (unsigned long long)w1);
                            if (p1 == EMPTY || p1 == w1) { atomicAdd(&tcnt[slot], wgt); done = true; }
                        }
                    }
                    if (!done) {
                        const uint32_t step = dh ? wide_step(g) : 1u;
#pragma nounroll
                        for (int probe = 1;; probe++) {
                            slot = wide_next(slot, step);
                            const unsigned long long p0 = atomicCAS(&tk0[slot], EMPTY, (unsigned long long)w0);
                            if (p0 == EMPTY || p0 == w0) {
                                const unsigned long long p1 = atomicCAS(&tk1[slot], EMPTY, (unsigned long long)w1);
                                if (p1 == EMPTY || p1 == w1) { atomicAdd(&tcnt[slot], wgt); break; }
                            }
                            if (probe >= LPROBE) { overflow = 1; break; }
                        }
                    }
                }
            };
            auto insert1 = [&](const uint64_t w0, const uint64_t w1, bool v) __attribute__((always_inline)) {
                if constexpr (LF) { insertw(w0, w1, v, 1u); return; }
                const uint32_t g = ((uint32_t)w0 ^ __builtin_rotateleft32((uint32_t)(w0 >> 32), 13) ^
                                    ((uint32_t)w1 * 0x85EBCA6Bu) ^ (uint32_t)(w1 >> 32)) * 0x9E3779B1u;
                v = v && ((g >> 4) & ((S - 1u) & 0xffffu)) == s;       // (one part: mask 0, s = 0)
                uint32_t slot = wide_slot(g);
                // double hashing: an odd step from other hash bits (the whole wave waits for its longest probe
                // sequence, and linear probing's clusters make that one long when a leaf fills its table)
                const uint32_t step = dh ? wide_step(g) : 1u;
                uint32_t c = 1u;
                if (v) c = atomicCAS(&tcnt[slot], 0u, WLOCK);
                if (v && c == 0u) {                                   // claimed: write the key, publish count 1
                    tk[slot] = make_ulonglong2(w0, w1);
                    __threadfence_block();
                    atomicExch(&tcnt[slot], 1u);
                }
                const bool h = v && c != 0u && c != WLOCK;
                const ulonglong2 t01 = h ? tk[slot] : make_ulonglong2(0, 0);
                const bool hit = h && t01.x == w0 && t01.y == w1;
                if (hit) atomicAdd(&tcnt[slot], 1u);
                int probe = 0;
                bool done = !v || c == 0u || hit;
                while (__ballot(!done)) {
                    if (!done) {
                        if (c == 0u) {
                            tk[slot] = make_ulonglong2(w0, w1);
                            __threadfence_block();
                            atomicExch(&tcnt[slot], 1u);
                            done = true;
                        } else if (c != WLOCK) {
                            const ulonglong2 t = tk[slot];
                            if (t.x == w0 && t.y == w1) { atomicAdd(&tcnt[slot], 1u); done = true; }
                            else {
                                slot = wide_next(slot, step);
                                if (++probe >= LPROBE) { overflow = 1; done = true; }
                            }
                        }
                        if (!done) c = atomicCAS(&tcnt[slot], 0u, WLOCK);   // next slot, or the same one while it is being written
                    }
                }
            };
            // Record leaves, RFX_WIDE_QUEUE: probing as a queue of ATTEMPTS.  An attempt looks at one slot: empty -> claim,
            // write the key, publish; the key itself -> count; another key -> the next slot of the sequence; a slot
            // being written -> the same slot again.  An attempt that did not finish the key goes (back) to the wave's
            // queue, and whenever 64 wait they are made together, one per lane.  No lane ever waits for another, so there is
            // no loop the wave must leave together, and the wave pays the AVERAGE number of probes per key instead of
            // the longest sequence among 64 in every round (site-laden leaves run their tables at 60-80 %).
            uint32_t qn = 0;
            ulonglong2 *wqk = (ulonglong2 *)(wstage + (size_t)(threadIdx.x >> 6) * WWS + WWS0);
            uint32_t *wqs = (uint32_t *)(wqk + WQCAP);
            // (wgt: the k-mer stands for that many instances -- the count of the record it was cut from)
            auto attempt = [&](const uint64_t w0, const uint64_t w1, uint32_t slot, uint32_t probe, bool v, uint32_t wgt) __attribute__((always_inline)) {
                uint32_t c = 1u;
                if (v) c = atomicCAS(&tcnt[slot], 0u, WLOCK);
                bool done = !v;
                if (v) {
                    if (c == 0u) {
                        tk[slot] = make_ulonglong2(w0, w1);
                        __threadfence_block();
                        atomicExch(&tcnt[slot], wgt);
                        done = true;
                    } else if (c != WLOCK) {
                        const ulonglong2 t = tk[slot];
                        if (t.x == w0 && t.y == w1) { atomicAdd(&tcnt[slot], wgt); done = true; }
                        else {
                            const uint32_t g = ((uint32_t)w0 ^ __builtin_rotateleft32((uint32_t)(w0 >> 32), 13) ^
                                                ((uint32_t)w1 * 0x85EBCA6Bu) ^ (uint32_t)(w1 >> 32)) * 0x9E3779B1u;
                            slot = wide_next(slot, dh ? wide_step(g) : 1u);
                            if (++probe >= (uint32_t)LPROBE) { overflow = 1; done = true; }
                        }
                    }
                }
                const uint64_t m = __ballot(!done);
                if (m) {
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (!done) { wqk[qn + r] = make_ulonglong2(w0, w1); wqs[qn + r] = slot | (probe << 13) | (wgt << 19); }
                    qn += (uint32_t)__popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
            };
            auto pending = [&](uint32_t n) __attribute__((always_inline)) {      // the top n <= 64 entries of the queue
                qn -= n;
                const bool v = (uint32_t)lane_ < n;
                const ulonglong2 key = v ? wqk[qn + lane_] : make_ulonglong2(0, 0);
                const uint32_t st = v ? wqs[qn + lane_] : 0u;
                __builtin_amdgcn_wave_barrier();
                attempt(key.x, key.y, st & 0x1fffu, (st >> 13) & 63u, v, st >> 19);
            };
            static_assert(WCAP <= 8192 && LPROBE < 64, "a queue entry: 13 bits of slot, 6 of probes, 13 of weight");
            auto insertq = [&](const uint64_t w0, const uint64_t w1, bool v, uint32_t wgt) __attribute__((always_inline)) {
                const uint32_t g = ((uint32_t)w0 ^ __builtin_rotateleft32((uint32_t)(w0 >> 32), 13) ^
                                    ((uint32_t)w1 * 0x85EBCA6Bu) ^ (uint32_t)(w1 >> 32)) * 0x9E3779B1u;
                v = v && ((g >> 4) & ((S - 1u) & 0xffffu)) == s;       // (one part: mask 0, s = 0)
                attempt(w0, w1, wide_slot(g), 0u, v, wgt);
#pragma nounroll
                while (qn >= 64u) pending(64u);
            };
            if constexpr (RECS) {
                // every wave takes an equal contiguous share of the leaf, 64 records at a time.  A record is first counted
                // in the record table; what finds no slot there is PARKED in the wave's LDS area, and whenever the area is
                // full the parked records are expanded: a bit per output position marks where each record's windows
                // start, and lane j extracts the k-mer at position j (its record by a prefix popcount of those bits).
                // After the leaf's last record (one barrier) the table's slots are parked and expanded the same way, every
                // k-mer weighing what its record counted.
                constexpr int NWV = WLT / 64;
                const int wave_ = threadIdx.x >> 6;
                const uint64_t n = end - begin;
                const uint64_t ws = begin + n * wave_ / NWV, we = begin + n * (wave_ + 1) / NWV;
                WRec *wrec = (WRec *)(wstage + (size_t)wave_ * WWS);
                uint32_t *wbits = (uint32_t *)(wrec + WPARK);
                uint32_t *wcum = wbits + 32;
                const int t = k - 32;
                const bool agg = WAGG && n <= (uint64_t)WRMAX && !(presplit & 0x10000000u) && agg_on;      // (every hash-selected part of a split leaf too)
                uint32_t hits = 0;
                uint32_t parked = 0;                                 // wave-uniform
                auto kmer_at_pos = [&](uint32_t j, uint64_t *k0, uint64_t *k1, uint32_t *wgt) __attribute__((always_inline)) {
                    const uint32_t wd = wbits[j >> 5];
                    const uint32_t ri = wcum[j >> 5] + (uint32_t)__popc(wd & (0xffffffffu >> (31 - (j & 31)))) - 1u;
                    const WRec rr = wrec[ri];
                    *wgt = (uint32_t)rr.hd >> 11;
                    wrec_kmer(rr, j - ((uint32_t)rr.hd & 2047u), t, k0, k1);
                };
                auto flush_parked = [&]() __attribute__((always_inline)) {
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t w4 = (uint32_t)lane_ < parked ? (uint32_t)wrec[lane_].hd : 0u;      // windows | weight << 11
                    const uint32_t nwin = w4 & 31u;
                    const uint32_t x = wave_incl_scan(nwin);
                    const uint32_t off = x - nwin;
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                    if (lane_ < 32) wbits[lane_] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (nwin) { atomicOr(&wbits[off >> 5], 1u << (off & 31)); *(uint32_t *)&wrec[lane_].hd = off | (w4 & ~2047u); }
                    __builtin_amdgcn_wave_barrier();
                    {
                        const uint32_t c = (uint32_t)__popc(wbits[lane_ & 31]);
                        const uint32_t y = wave_incl_scan(c);
                        if (lane_ < 32) wcum[lane_] = y - c;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (one k-mer per lane and round: pairing them, as the k <= 31 leaf does, measured 3 ms slower here)
                    for (uint32_t wb = 0; wb < total; wb += 64) {
                        const uint32_t ja = wb + lane_;
                        const bool va = ja < total;
                        uint64_t a0, a1;
                        uint32_t wg;
                        kmer_at_pos(va ? ja : 0, &a0, &a1, &wg);
                        if (presplit & 0x40000000u) {                       // ablation (RFX_WIDE_DBG=1): expand, no table
                            if (va && (a0 ^ a1) == 0x123456789ULL) overflow = 1;
                            continue;
                        }
                        if constexpr (LF) insertw(a0, a1, va, wg);
                        else if constexpr (WQCAP > 0) insertq(a0, a1, va, wg);
                        else insert1(a0, a1, va);
                    }
                    __builtin_amdgcn_wave_barrier();
                    parked = 0;
                };
                // (valid in at most WPARK lanes)
                auto park = [&](const WRec &rc, const bool valid, const uint32_t weight) __attribute__((always_inline)) {
                    const uint64_t vm = __ballot(valid);
                    if (!vm) return;
                    const uint32_t d = (uint32_t)__popcll(vm);
#pragma nounroll
                    while (parked + d > (uint32_t)WPARK) flush_parked();      // (once; a loop so that the body is not duplicated)
                    if (valid) {
                        const uint32_t ord = parked + __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
                        WRec r2 = rc;
                        r2.hd = (rc.hd & ~0xffffffffULL) | (uint64_t)((uint32_t)rec_len(rc) | (weight << 11));
                        wrec[ord] = r2;
                    }
                    parked += d;
                };
                // the record's own bases: k - 1 + windows of them; what follows in b1 / b2 is whatever followed in the read
                auto place = [&](const WRec &r, bool valid) __attribute__((always_inline)) -> bool {
                    if constexpr (!WAGG) return false;
                    if (!agg || !valid) return false;
                    const uint32_t nw1 = (uint32_t)(r.hd >> 32) & 15u;
                    const int ub = 2 * (k + (int)nw1);                             // bits of the base string in use (66..156)
                    const uint64_t b1 = ub >= 128 ? r.b1 : r.b1 & ~(~0ULL >> (ub - 64));
                    const uint32_t c = (ub > 128 ? (uint32_t)(r.b2 >> 32) & ~(0xffffffffu >> (ub - 128)) & 0xfffffff0u : 0u) | nw1;
                    if (r.b0 == EMPTY || b1 == EMPTY) return false;
                    const uint32_t h = ((uint32_t)r.b0 ^ __builtin_rotateleft32((uint32_t)(r.b0 >> 32), 13) ^ __builtin_rotateleft32((uint32_t)b1, 7) ^
                                        __builtin_rotateleft32((uint32_t)(b1 >> 32), 19) ^ (c * 0x85EBCA6Bu)) * 0x9E3779B1u;
                    uint32_t slot = WRSLOTS == 768 ? ((h >> 22) * 3u) >> 2 : WRSLOTS == 1024 ? h >> 22 : h >> 23;
                    static_assert(WRSLOTS == 768 || WRSLOTS == 512 || WRSLOTS == 1024, "slot = three quarters of ten hash bits, ten, or nine");
#pragma unroll
                    for (int probe = 0; probe < RPROBE; probe++) {
                        const unsigned long long pa = atomicCAS(&rA[slot], EMPTY, (unsigned long long)r.b0);
                        if (pa == EMPTY || pa == r.b0) {
                            const unsigned long long pb = atomicCAS(&rB[slot], EMPTY, (unsigned long long)b1);
                            if (pb == EMPTY || pb == b1) {
                                const unsigned long long pc = atomicCAS(&rCC[slot], EMPTY, (unsigned long long)c << 32);
                                if (pc == EMPTY || (uint32_t)(pc >> 32) == c) { atomicAdd((uint32_t *)&rCC[slot], 1u); return true; }
                            }
                        }
                        slot = slot + 1u == (uint32_t)WRSLOTS ? 0u : slot + 1u;
                    }
                    return false;
                };
                WRec nxt;
                if (first_pass) {
                    nxt = pre;
                    if (leaf + 1 < l1) pre = wave_first(leaf_off[leaf + 1], leaf_end[leaf + 1]);      // travels while this leaf is counted
                } else {
                    nxt = ws + lane_ < we ? elems[ws + lane_] : WRec{0, 0, 0, 0};          // (a later part of a split leaf)
                }
                first_pass = false;
                for (uint64_t r0 = ws; r0 < we; r0 += 64) {
                    if (__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    const bool valid = r0 + lane_ < we;
                    const WRec rc = nxt;
                    nxt = r0 + 64 + lane_ < we ? elems[r0 + 64 + lane_] : WRec{0, 0, 0, 0};   // travels during the expansion
                    const bool un = valid && !place(rc, valid);
                    hits += (uint32_t)__popcll(__ballot(valid && !un));
                    if (presplit & 0x08000000u) {                           // RFX_WIDE_DBG=... statistics
                        const uint64_t mp = __ballot(valid && !un), md = __ballot(un);
                        if (lane_ == 0) { atomicAdd(&co->r_placed, (unsigned long long)__popcll(mp)); atomicAdd(&co->r_direct, (unsigned long long)__popcll(md)); }
                    }
                    if constexpr (WPARK >= 64) park(rc, un, 1u);
                    else {
#pragma nounroll
                        for (int hf = 0; hf < 64 / WPARK; hf++) park(rc, un && lane_ / WPARK == hf, 1u);
                    }
                }
                if constexpr (WAGG) {
                    if (agg) {
                        __syncthreads();                             // every record of the leaf is counted
                        const uint32_t slot = (uint32_t)(wave_ * 64 + lane_);
                        const bool mine = slot < (uint32_t)WRSLOTS;
                        const unsigned long long a_ = mine ? rA[slot] : EMPTY, b_ = mine ? rB[slot] : EMPTY, cc = mine ? rCC[slot] : EMPTY;
                        const bool have = cc != EMPTY;
                        if (a_ != EMPTY) { rA[slot] = EMPTY; rB[slot] = EMPTY; rCC[slot] = EMPTY; }
                        {
                            const uint32_t used = (uint32_t)__popcll(__ballot(have));
                            if (lane_ == 0 && hits != used) atomicAdd(&agg_saved, hits - used);
                        }
                        if (presplit & 0x08000000u) {
                            const uint64_t mh = __ballot(have);
                            if (lane_ == 0) atomicAdd(&co->r_slots, (unsigned long long)__popcll(mh));
                        }
                        if (!__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            const uint32_t c = (uint32_t)(cc >> 32);
                            const WRec rr{(uint64_t)a_, (uint64_t)b_, (uint64_t)(c & 0xfffffff0u) << 32, (uint64_t)(c & 15u) << 32};
#pragma nounroll
                            for (int hf = 0; hf < 64 / WPARK; hf++) park(rr, have && lane_ / WPARK == hf, (uint32_t)cc);
                        }
                    }
                }
#pragma nounroll
                while (parked) flush_parked();
                if constexpr (WQCAP > 0) {
#pragma nounroll
                    while (qn > 0u && !__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) pending(qn < 64u ? qn : 64u);
                }
            } else {
                for (uint64_t i = begin + threadIdx.x; i < end; i += WLT) {
                    if (__hip_atomic_load(&overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    const Rec e = elems[i];
                    insert1(e.w0, e.w1, true);
                }
            }
            __syncthreads();
            const bool ov = overflow != 0;
            if (threadIdx.x == 0) my_passes++;
            if (!ov) {
                // (the threshold moves BEFORE the sweep, whose closing barrier stands between this store and the next
                // leaf's read of ps_eff: threads that saw different thresholds would disagree on S and on their barriers)
                if (RECS && threadIdx.x == 0 && S == 1 && end - begin > (uint64_t)ps_eff * 3 / 4 && ps_eff < (1u << 24)) ps_eff += ps_eff / 64 + 1;
                if constexpr (WAGG) {
                    if (threadIdx.x == 0 && agg_on && end - begin <= (uint64_t)WRMAX) {
                        agg_total += (uint32_t)(end - begin);
                        if (agg_total >= 8192u) {
                            if (agg_saved * 2u < agg_total) agg_on = 0;
                            agg_total = 0; agg_saved = 0;
                        }
                    }
                }
                // sweep: survivors -> LDS buffer (one add per wave), slots reset
                for (int base = 0; base < WCAP; base += WLT) {
                    const int slot = base + threadIdx.x;
                    if (slot >= WCAP) break;                         // (wave-uniform: WCAP is a multiple of 64)
                    const uint32_t c = tcnt[slot];
                    my_distinct += c != 0;
                    bool keep = c != 0;                              // counter64 filters :197-205
                    if (min_cov > 1 && c < (uint32_t)min_cov) keep = false;
                    if (max_cov < 10000000 && c > (uint32_t)max_cov) keep = false;
                    const uint64_t km = __ballot(keep);
                    if (km) {
                        const uint32_t cntw = (uint32_t)__popcll(km);
                        const int leader = __ffsll((unsigned long long)km) - 1;
                        const uint32_t r = (uint32_t)__popcll(km & ((1ULL << lane_) - 1));
                        uint32_t b0 = 0;
                        if (lane_ == leader) b0 = atomicAdd(&ob_n, cntw);
                        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
                        if (b0 + cntw <= (uint32_t)WOBUF) {
                            if (keep) { const ulonglong2 t = LF ? make_ulonglong2(tk0[slot], tk1[slot]) : tk[slot]; obh[b0 + r] = t.x; obl[b0 + r] = t.y; obc[b0 + r] = c; }
                        } else {
                            uint32_t glo = 0, ghi = 0;
                            if (lane_ == leader) {
                                atomicMin(&ob_lim, b0);
                                const unsigned long long gg = atomicAdd(&co->n_out, (unsigned long long)cntw);
                                glo = (uint32_t)gg; ghi = (uint32_t)(gg >> 32);
                            }
                            glo = (uint32_t)__builtin_amdgcn_readlane((int)glo, leader);
                            ghi = (uint32_t)__builtin_amdgcn_readlane((int)ghi, leader);
                            const unsigned long long pos = (((unsigned long long)ghi << 32) | glo) + r;
                            if (keep && pos < cap) {
                                const ulonglong2 t = LF ? make_ulonglong2(tk0[slot], tk1[slot]) : tk[slot];
                                out_keys[2 * pos] = t.x; out_keys[2 * pos + 1] = t.y; out_counts[pos] = (int64_t)c;
                            }
                        }
                    }
                    tcnt[slot] = 0;
                    if constexpr (LF) { if (c) { tk0[slot] = EMPTY; tk1[slot] = EMPTY; } }
                }
                __syncthreads();
                const uint32_t raw = ob_n, lim = ob_lim;
                if (raw >= (uint32_t)WOBUF / 2 || lim != 0xffffffffu) flush();
                if (S == 1) break;
            } else {
                for (int i = threadIdx.x; i < WCAP; i += WLT) { tcnt[i] = 0; if constexpr (LF) { tk0[i] = EMPTY; tk1[i] = EMPTY; } }
                __syncthreads();
                if (threadIdx.x == 0) {
                    my_overflows++;
                    overflow = 0;
                    if constexpr (RECS) {
                        if ((end - begin) / S < 2ull * ps_eff && ps_eff > 64) ps_eff -= ps_eff / 8;
                    }
                    if (sp + 2 <= LSTACK && S < (1u << 16)) {
                        stackS[sp] = 2 * S; stacks[sp] = s + S; sp++;
                        stackS[sp] = 2 * S; stacks[sp] = s;     sp++;
                    } else {
                        atomicAdd(&co->n_failed, 1ULL);
                    }
                }
                __syncthreads();
            }
            if (sp == 0) break;
            S = stackS[sp - 1]; s = stacks[sp - 1];
            __syncthreads();
            if (threadIdx.x == 0) sp--;
        }
        if constexpr (RECS) {
            if (first_pass && leaf + 1 < l1) pre = wave_first(leaf_off[leaf + 1], leaf_end[leaf + 1]);     // (an empty leaf hands the chain on)
        }
    }
    flush();
    {
        uint32_t d = my_distinct;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane_ == 0 && d) atomicAdd(&co->n_distinct, (unsigned long long)d);
    }
    if (threadIdx.x == 0) {
        atomicAdd(&co->n_passes, my_passes);
        if (my_overflows) atomicAdd(&co->n_overflow, my_overflows);
    }
}

// ---- heavy leaves: slices for the second launch, and the merge of the slices' partial counts

__global__ void k_heavy_count(const uint64_t *__restrict__ off, const uint64_t *__restrict__ end, int64_t nleaf, uint64_t heavy, uint64_t slice,
                              uint64_t *__restrict__ nsl) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaf) return;
    const uint64_t n = end[l] - off[l];
    nsl[l] = n > heavy ? (n + slice - 1) / slice : 0;
}

__global__ void k_heavy_fill(const uint64_t *__restrict__ off, const uint64_t *__restrict__ end, int64_t nleaf, const uint64_t *__restrict__ pos,
                             uint64_t *__restrict__ sb, uint64_t *__restrict__ se) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaf) return;
    const uint64_t c = pos[l + 1] - pos[l];
    if (!c) return;
    const uint64_t b = off[l], n = end[l] - b;
    for (uint64_t j = 0; j < c; j++) {            // equal slices
        sb[pos[l] + j] = b + n * j / c;                 // n < 2^40, j < 2^24: no overflow
        se[pos[l] + j] = b + n * (j + 1) / c;
    }
}

// partial (key, count) pairs of the slices, sorted by key: sum every run, filter, append
__global__ void k_reduce_partials(const uint64_t *__restrict__ pk, const uint32_t *__restrict__ pc, int64_t np,
                                  int min_cov, int max_cov, int apply_filter, uint64_t *__restrict__ out_keys,
                                  int32_t *__restrict__ out_counts, unsigned long long cap, CountOut *__restrict__ co,
                                  int pair_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const uint64_t key = pk[i];
    if (i > 0 && pk[i - 1] == key) return;
    uint32_t sum = 0;                              // Integer sums wrap in the reference too (:2895-2899)
    for (int64_t j = i; j < np && pk[j] == key; j++) sum += pc[j];
    atomicAdd(&co->n_distinct, 1ULL);
    const int32_t c = (int32_t)sum;
    if (apply_filter && !(c >= min_cov && c <= max_cov)) return;
    const unsigned long long o = atomicAdd(&co->n_out, 1ULL);
    if (o < cap) {
        if (pair_out) ((Rec *)out_keys)[o] = Rec{key, (uint64_t)(uint32_t)c};
        else { out_keys[o] = key; out_counts[o] = c; }
    }
}

// ------------------------------------------------------- super-k-mer records

// Level 1 from reads, second form: instead of one 8-byte k-mer per window, a thread emits one
// 16-byte record per RUN of consecutive windows (inside its 16-window segment) that share a
// minimiser -- the canonical 13-mer with the smallest hash among the window's W = k-12 m-mers.
// The minimiser is a function of the canonical k-mer (the set of canonical m-mers is the same on
// both strands), so every instance of a k-mer follows the same bucket path; buckets are radix
// digits of the minimiser's hash.  ~6 windows per record -> ~2.6 B per instance through the
// partition levels instead of 8, and the leaf expands records straight into its LDS table, so the
// 8-byte instance array never exists in HBM.  (KMC / Gerbil-style, re-cut for wave64 + LDS.)
constexpr int SK_M = 13;
constexpr int SK_MIN_W = 9;           // record path for k = 21..31 (W = k - 12 m-mers per window)
constexpr int SKT = 1024;             // threads per workgroup of the reads -> records kernels
#ifndef RFX_SK_ROLL
#define RFX_SK_ROLL 0               // 1: the m-mers rolled a base at a time (rounds 1-2)
#endif
#ifndef SK_HIST_RUNLOOP
#define SK_HIST_RUNLOOP false
#endif
#ifndef SK_SCATTER_RUNLOOP
#define SK_SCATTER_RUNLOOP true
#endif
#ifndef SK_DESC_RUNLOOP
#define SK_DESC_RUNLOOP true
#endif

// order of the m-mers: a bijection of the canonical 26-bit m-mer onto 32 bits (odd multiplier,
// xor-shift), so two different m-mers never tie and a plain 32-bit minimum picks the minimiser;
// the value itself (not the m-mer) names the bucket
// (RFX_MMER_KEY24, round 3: ONE full-rate instruction, v_mad_u32_u24 -- the low 24 bits of the m-mer times a 24-bit odd
// constant, plus the m-mer (which brings in its first base, bits 24-25) -- instead of a quarter-rate 32-bit multiply, a
// shift and an xor: 34 m-mers per segment make this 15 % of level 1's issue cycles.  Not a bijection any more, and it need
// not be one: the value, not the m-mer, names the run and its bucket, so two m-mers that tie simply share a run -- the
// bucket is still a function of the canonical k-mer.)
#ifndef RFX_MMER_KEY24
#define RFX_MMER_KEY24 1
#endif
__device__ __forceinline__ uint32_t mmer_key(uint32_t canon) {
#if RFX_MMER_KEY24
    return __umul24(canon, 0x9E3779u) + canon;
#else
    uint32_t h = canon * 0x9E3779B1u;
    return h ^ (h >> 15);
#endif
}

// digit of a record at a level that follows `used` radix bits
__device__ __forceinline__ unsigned rec_digit(uint32_t hdr, int used, int bits) {
    return bits ? (unsigned)((hdr << used) >> (32 - bits)) : 0u;
}

// Walks the runs of one segment; calls emit(first_window, n_windows, minimiser_key).
// SEG: windows per segment, 16 or 32 (round 3).  A thread that owns 32 windows evaluates 50 m-mers where two threads of 16
// evaluate 68, pays the per-segment work once, and -- what counts downstream -- a read is cut into records at half as
// many places that only depend on where the read happened to start: 19 % fewer records, and more of them identical from
// read to read.  Runs stay <= 16 windows (four bits in the record): a longer one is cut 16 windows after its start.
template <int W, bool RUNLOOP, int SEG = 16, class F>
__device__ __forceinline__ void seg_runs(const ReadSrc &s, int nk_r, int sgm, const uint64_t (&w)[3], F &&emit,
                                         uint64_t *hi_out, uint64_t *lo_out, uint32_t *wm_lds = nullptr) {
    static_assert(SEG == 16 || SEG == 32, "segment");
    constexpr int PK = SEG;                        // (shadows the file-wide 16 inside this function)
    constexpr int NM = PK + W - 1;                 // m-mers a segment can touch (<= 34, or <= 50)
    static_assert(NM + SK_M - 1 <= 64, "the segment's bases lie in (hi, lo)");
    const int p0 = sgm * PK;
    int v = nk_r - p0;
    v = v > PK ? PK : v;
    if (v <= 0) return;                            // a short read of a ragged set: no window in this segment
    const int sh = 2 * ((s.fc + p0) & 31);
    const uint64_t hi = sh ? (w[0] << sh) | (w[1] >> (64 - sh)) : w[0];
    const uint64_t lo = sh ? (w[1] << sh) | (w[2] >> (64 - sh)) : w[1];
    *hi_out = hi; *lo_out = lo;
    constexpr int M2 = 2 * SK_M;
    const uint32_t mmask = (1u << M2) - 1;
    uint32_t val[NM];
#if RFX_SK_ROLL
    uint32_t fm = (uint32_t)(hi >> (64 - M2));
    uint32_t rm = (uint32_t)revcomp((uint64_t)fm, SK_M);
#pragma unroll
    for (int j = 0; j < NM; j++) {
        val[j] = mmer_key(fm < rm ? fm : rm);
        // base M + j of the 64-base stream (hi, lo): compile-time position
        const int pos = SK_M + j;
        const uint32_t b = pos < 32 ? (uint32_t)(hi >> (62 - 2 * pos)) & 3u : (uint32_t)(lo >> (62 - 2 * (pos - 32))) & 3u;
        fm = ((fm << 2) | b) & mmask;
        rm = (rm >> 2) | ((b ^ 3u) << (M2 - 2));
    }
#else
    // Every m-mer is CUT out of the 64-base stream, and its reverse complement out of the stream's reverse complement
    // (made once: the complement of m-mer j is the m-mer at base 51 - j of it), at compile-time positions: a bit-field
    // extract when the 26 bits lie in one 32-bit word, a funnel shift and a shift otherwise -- 3.5 instructions for the
    // pair instead of the 6 of rolling both through a base at a time.
    static_assert(SK_M == 13 && NM <= 50, "positions below");
    const uint64_t rhi = revcomp(lo, 32), rlo = revcomp(hi, 32);
    const uint32_t FW[4] = {(uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(lo >> 32), (uint32_t)lo};
    const uint32_t RW[4] = {(uint32_t)(rhi >> 32), (uint32_t)rhi, (uint32_t)(rlo >> 32), (uint32_t)rlo};
    auto cut = [&](const uint32_t (&Wd)[4], const int pos) __attribute__((always_inline)) -> uint32_t {
        const int wi = pos >> 4, off = 2 * (pos & 15);
        if (off + M2 <= 32) return (Wd[wi] >> (32 - M2 - off)) & mmask;
        return __builtin_amdgcn_alignbit(Wd[wi], Wd[wi < 3 ? wi + 1 : 3], 32 - off) >> (32 - M2);
    };
#pragma unroll
    for (int j = 0; j < NM; j++) {
        const uint32_t fm = cut(FW, j), rm = cut(RW, 64 - SK_M - j);
        val[j] = mmer_key(fm < rm ? fm : rm);
    }
#endif
    // per-window minimiser = sliding minimum over W m-mers (van Herk: suffix minima and prefix minima
    // inside blocks of W m-mers; a window spans at most two blocks)
    uint32_t wm[PK];
    if constexpr (W >= PK - 1) {
        // two blocks cover the segment: suffixes of [0, W) and prefixes of [W, NM), in place
#pragma unroll
        for (int j = W - 2; j >= 0; j--) val[j] = val[j] < val[j + 1] ? val[j] : val[j + 1];
#pragma unroll
        for (int j = W + 1; j < NM; j++) val[j] = val[j] < val[j - 1] ? val[j] : val[j - 1];
        wm[0] = val[0];
#pragma unroll
        for (int i = 1; i < PK; i++) wm[i] = val[i] < val[W + i - 1] ? val[i] : val[W + i - 1];
    } else {
        uint32_t sfx[NM], pfx[NM];
#pragma unroll
        for (int j = NM - 1; j >= 0; j--)
            sfx[j] = (j % W == W - 1 || j == NM - 1) ? val[j] : (val[j] < sfx[j + 1] ? val[j] : sfx[j + 1]);
#pragma unroll
        for (int j = 0; j < NM; j++) pfx[j] = (j % W == 0) ? val[j] : (val[j] < pfx[j - 1] ? val[j] : pfx[j - 1]);
#pragma unroll
        for (int i = 0; i < PK; i++) wm[i] = sfx[i] < pfx[i + W - 1] ? sfx[i] : pfx[i + W - 1];
    }
    if constexpr (!RUNLOOP) {
        // cheap emit(): call it where the run ends (16 divergent call sites)
        uint32_t cur = wm[0];
        int start = 0;
#pragma unroll
        for (int i = 1; i < PK; i++) {
            if (i < v && (wm[i] != cur || (PK > 16 && i - start == 16))) {
                emit(start, i - start, cur);
                cur = wm[i]; start = i;
            }
        }
        emit(start, v - start, cur);
        return;
    }
    // a bit per window that starts a run.  The runs are walked in a loop of their own so that emit()
    // -- the expensive part -- runs once per run of the busiest lane (~6 times) instead of once per
    // window position (16 divergent call sites).
    uint32_t starts = 1u;
#pragma unroll
    for (int i = 1; i < PK; i++) starts |= (uint32_t)(wm[i] != wm[i - 1]) << i;
    starts &= 0xffffffffu >> (32 - v);                 // (1 <= v <= PK: the windows the read has in this segment)
    if constexpr (PK > 16) {
        // no run longer than 16 windows: `cov` = the windows within 15 of a start at or before them; the lowest window
        // that is not lies exactly 16 after a start and becomes one (bit 0 is set, so whatever is not covered lies in
        // the upper half, and the new start covers all that is left of it)
        uint32_t cov = starts;
        cov |= cov << 1; cov |= cov << 2; cov |= cov << 4; cov |= cov << 8;
        const uint32_t unc = ~cov & (0xffffffffu >> (32 - v));
        starts |= unc & (0u - unc);
    }
    if (wm_lds) {
#pragma unroll
        for (int i = 0; i < PK; i++) wm_lds[i * SKT + threadIdx.x] = wm[i];
    }
    // (the loop below picks wm[i0] with a tree of selects.  A select between two ELEMENTS of an array is turned into
    // an indexed load by the compiler, and the array into scratch memory: the minima become sixteen scalars, results of
    // an empty asm, which are no elements of anything)
    uint32_t m0 = wm[0], m1 = wm[1], m2 = wm[2], m3 = wm[3], m4 = wm[4], m5 = wm[5], m6 = wm[6], m7 = wm[7], m8 = wm[8],
             m9 = wm[9], m10 = wm[10], m11 = wm[11], m12 = wm[12], m13 = wm[13], m14 = wm[14], m15 = wm[15];
    if (!wm_lds)
        asm("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7), "+v"(m8), "+v"(m9), "+v"(m10),
                 "+v"(m11), "+v"(m12), "+v"(m13), "+v"(m14), "+v"(m15));
    constexpr int UP = PK > 16 ? 16 : 0;               // (SEG = 16: the upper sixteen alias the lower and are never selected)
    uint32_t n0 = wm[UP + 0], n1 = wm[UP + 1], n2 = wm[UP + 2], n3 = wm[UP + 3], n4 = wm[UP + 4], n5 = wm[UP + 5], n6 = wm[UP + 6],
             n7 = wm[UP + 7], n8 = wm[UP + 8], n9 = wm[UP + 9], n10 = wm[UP + 10], n11 = wm[UP + 11], n12 = wm[UP + 12],
             n13 = wm[UP + 13], n14 = wm[UP + 14], n15 = wm[UP + 15];
    if (!wm_lds && PK > 16)
        asm("" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7), "+v"(n8), "+v"(n9), "+v"(n10),
                 "+v"(n11), "+v"(n12), "+v"(n13), "+v"(n14), "+v"(n15));
    while (starts) {
        const int i0 = __ffs((int)starts) - 1;
        starts &= starts - 1;
        const int i1 = starts ? __ffs((int)starts) - 1 : v;
        // wm[i0]: a register array cannot be indexed at run time -- through the thread's own LDS column
        // when the caller has one (16 conflict-free stores, one load per run), else a compare-select chain
        uint32_t key;
        if (wm_lds) {
            key = wm_lds[i0 * SKT + threadIdx.x];
        } else {
            // (a tree of selects on the bits of i0: 15 selects under 4 masks, where a chain of 15 compares each waits
            // for its own mask)
            const bool b0 = i0 & 1, b1 = i0 & 2, b2 = i0 & 4, b3 = i0 & 8;
            const uint32_t a0 = b0 ? m1 : m0, a1 = b0 ? m3 : m2, a2 = b0 ? m5 : m4, a3 = b0 ? m7 : m6, a4 = b0 ? m9 : m8,
                           a5 = b0 ? m11 : m10, a6 = b0 ? m13 : m12, a7 = b0 ? m15 : m14;
            const uint32_t c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2, c2 = b1 ? a5 : a4, c3 = b1 ? a7 : a6;
            const uint32_t d0 = b2 ? c1 : c0, d1 = b2 ? c3 : c2;
            key = b3 ? d1 : d0;
            if constexpr (PK > 16) {                   // a fifth level: the same tree over the upper sixteen
                const uint32_t e0 = b0 ? n1 : n0, e1 = b0 ? n3 : n2, e2 = b0 ? n5 : n4, e3 = b0 ? n7 : n6, e4 = b0 ? n9 : n8,
                               e5 = b0 ? n11 : n10, e6 = b0 ? n13 : n12, e7 = b0 ? n15 : n14;
                const uint32_t f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
                const uint32_t g0 = b2 ? f1 : f0, g1 = b2 ? f3 : f2;
                const uint32_t ku = b3 ? g1 : g0;
                key = (i0 & 16) ? ku : key;
            }
        }
        emit(i0, i1 - i0, key);
    }
}

__device__ __forceinline__ uint64_t mmer_hash64(uint32_t canon) { return kmer_hash((uint64_t)canon); }

__device__ __forceinline__ unsigned sk_digit(uint32_t canon, const Level &lv) {
    const uint64_t h = mmer_hash64(canon);
    if (lv.n_owners > 0) {
        const unsigned o = (unsigned)__umul64hi(h, (uint64_t)lv.n_owners);
        return lv.sub_bits ? (o << lv.sub_bits) | ((uint32_t)((h << OWNER_BITS) >> 32) >> (32 - lv.sub_bits)) : o;
    }
    return rec_digit((uint32_t)((h << OWNER_BITS) >> 32), 0, lv.bits);
}

// Run descriptors: what the scatter needs to know about a segment without redoing the minimiser
// arithmetic (two thirds of both kernels' instructions).  desc[g] = run-start bits of segment g |
// runs << 16, desc[(1 + r) * n_threads + g] = header word (32 hash bits) of its r-th run, r < SKD.
// Word-major, so the lanes of a wave store / load consecutive words; a segment with more than SKD
// runs (0.02 %) makes its wave recompute.
constexpr int SKD = 8;

template <int W, bool DESC>
__global__ __launch_bounds__(SKT) void k_sk_hist(ReadSrc s, Level lv, uint64_t *__restrict__ blockhist,
                                                 uint32_t *__restrict__ desc) {
    __shared__ uint32_t h[1 << MAX_BITS];
    __shared__ uint32_t wmcol[DESC && SK_DESC_RUNLOOP ? PK * SKT : 1];      // per-window minimisers, one column per thread
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    for (int i = threadIdx.x; i < nb; i += SKT) h[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * SKT;
    const int64_t dq = stride / s.segs;
    const int dr = (int)(stride - dq * s.segs);
    int64_t g = (int64_t)blockIdx.x * SKT + threadIdx.x;
    SegPos q;
    q.r = g / s.segs;
    q.sgm = (int)(g - q.r * s.segs);
    for (; g < s.n_threads; g += stride) {
        uint64_t w[3], hi, lo;
        seg_load(s, q, w);
        if constexpr (DESC) {
            uint32_t mask = 0, nr = 0;
            // run loop: the r-th run of every lane is handled in the same iteration, so the header
            // stores of a wave go to consecutive words
            uint32_t own[2] = {0, 0};                    // owner mode: the runs' owners, 8 bits each
            seg_runs<W, SK_DESC_RUNLOOP>(s, read_nk(s, q.r), q.sgm, w, [&](int i0, int, uint32_t canon) {
                const uint64_t h64 = mmer_hash64(canon);
                const uint32_t hdr = (uint32_t)((h64 << OWNER_BITS) >> 32);
                const unsigned d = lv.n_owners > 0 ? (unsigned)__umul64hi(h64, (uint64_t)lv.n_owners) : rec_digit(hdr, 0, lv.bits);
                atomicAdd(&h[d], 1u);
                if (nr < (uint32_t)SKD) {
                    desc[(int64_t)(1 + nr) * s.n_threads + g] = hdr;
                    if (nr < 4) own[0] |= d << (8 * nr); else own[1] |= d << (8 * (nr - 4));
                }
                mask |= 1u << i0; nr++;
            }, &hi, &lo, SK_DESC_RUNLOOP ? wmcol : nullptr);
            desc[g] = mask | (nr << 16);
            if (lv.n_owners > 0) {
                desc[(int64_t)(1 + SKD) * s.n_threads + g] = own[0];
                desc[(int64_t)(2 + SKD) * s.n_threads + g] = own[1];
            }
        } else {
            seg_runs<W, SK_HIST_RUNLOOP>(s, read_nk(s, q.r), q.sgm, w, [&](int, int, uint32_t canon) { atomicAdd(&h[sk_digit(canon, lv)], 1u); }, &hi, &lo);
        }
        q.r += dq; q.sgm += dr;
        if (q.sgm >= s.segs) { q.sgm -= s.segs; q.r++; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += SKT) blockhist[(int64_t)i * gridDim.x + blockIdx.x] = h[i];
}

// Records leave through write-combining rings (see k_rec_scatter_wc): every private per-digit
// stream has SKB record slots in LDS indexed by the record's final position, and only whole
// aligned 64-byte lines are stored; a digit that overruns its ring in one round stores directly.
constexpr int SKB = 8, SKA = 4;
// WIDE: the 32-byte records of the k = 33..63 path (s.k = the central window, s.wfl = the flank)
template <int W, bool DESC, bool WIDE = false>
__global__ __launch_bounds__(SKT, WIDE ? 4 : 8) void k_sk_scatter(ReadSrc s, Level lv, const uint64_t *__restrict__ scanned,
                                                              std::conditional_t<WIDE, WRec, Rec> *__restrict__ out,
                                                              const uint32_t *__restrict__ desc) {
    using RT = std::conditional_t<WIDE, WRec, Rec>;
    extern __shared__ __attribute__((aligned(32))) unsigned char sk_smem[];
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    // (the rings are addressed through these expressions, not through pointer variables: a pointer
    // captured by the lambdas below decays to a generic one and the LDS atomics with it)
#define buf ((RT *)sk_smem)
#define tail ((unsigned long long *)(sk_smem + (size_t)nb * SKB * sizeof(RT)))
#define head (tail + nb)
    for (int i = threadIdx.x; i < nb; i += SKT) tail[i] = head[i] = scanned[(int64_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    auto drain = [&](bool final) __attribute__((always_inline)) {
        for (int d = threadIdx.x / SKB; d < nb; d += SKT / SKB) {
            const int j = threadIdx.x % SKB;
            const unsigned long long h = head[d], t = tail[d];
            unsigned long long e, nh;
            if (t - h > (unsigned long long)SKB) { e = h + SKB; nh = t; }     // the excess went out directly
            else {
                e = final ? t : (t & ~(unsigned long long)(SKA - 1));
                if (e < h) e = h;
                nh = e;
            }
            const unsigned long long g = h + j;
            if (g < e) out[g] = buf[(size_t)d * SKB + (g & (SKB - 1))];
            if (j == 0) head[d] = nh;
        }
    };
    const int64_t stride = (int64_t)gridDim.x * SKT;
    const int64_t dq = stride / s.segs;
    const int dr = (int)(stride - dq * s.segs);
    const int64_t g0 = (int64_t)blockIdx.x * SKT;
    int64_t g = g0 + threadIdx.x;
    SegPos q;
    q.r = g / s.segs;
    q.sgm = (int)(g - q.r * s.segs);
    // every thread of the workgroup runs the same number of rounds (the barriers are uniform)
    for (int64_t gb = g0; gb < s.n_threads; gb += stride, g += stride) {
        if (g < s.n_threads) {
            uint64_t w[3], hi = 0, lo = 0;
            seg_load(s, q, w);
            // one record: windows [i0, i0 + n) of the segment, header word hdr, digit d
            auto put = [&](int i0, int n, uint32_t hdr, unsigned d) __attribute__((always_inline)) {
                RT r;
                if constexpr (WIDE) {
                    // 96 bases from the first base of the run's first k-mer: the central window starts
                    // s.wfl bases further on (words past the read's end are never used: k + n - 1 bases are)
                    const uint64_t *gw = s.words + q.r * s.wpr;
                    const int p = s.fc - s.wfl + q.sgm * PK + i0;
                    const int wi = p >> 5, sft = 2 * (p & 31), last = s.wpr - 1;
                    const uint64_t a0 = gw[wi < last ? wi : last], a1 = gw[wi + 1 < last ? wi + 1 : last];
                    const uint64_t a2 = gw[wi + 2 < last ? wi + 2 : last], a3 = gw[wi + 3 < last ? wi + 3 : last];
                    r.b0 = sft ? (a0 << sft) | (a1 >> (64 - sft)) : a0;
                    r.b1 = sft ? (a1 << sft) | (a2 >> (64 - sft)) : a1;
                    r.b2 = sft ? (a2 << sft) | (a3 >> (64 - sft)) : a2;
                    r.hd = ((uint64_t)(n - 1) << 32) | (uint64_t)hdr;
                } else {
                    const int sft = 2 * i0;
                    r.w0 = sft ? (hi << sft) | (lo >> (64 - sft)) : hi;
                    r.w1 = ((lo << sft) & 0xFFFFFFF000000000ULL) | ((uint64_t)(n - 1) << 32) | (uint64_t)hdr;
                }
                const unsigned long long pos = atomicAdd(&tail[d], 1ULL);
                if (pos - head[d] < (unsigned long long)SKB) buf[(size_t)d * SKB + (pos & (SKB - 1))] = r;
                else out[pos] = r;
            };
            // `hi`/`lo` are written before the first emit() runs (seg_runs stores them first)
            auto recompute = [&]() __attribute__((always_inline)) {
                seg_runs<W, SK_SCATTER_RUNLOOP>(s, read_nk(s, q.r), q.sgm, w, [&](int i0, int n, uint32_t canon) {
                    const uint64_t h = mmer_hash64(canon);
                    const unsigned d = lv.n_owners > 0 ? (unsigned)__umul64hi(h, (uint64_t)lv.n_owners)
                                                       : rec_digit((uint32_t)((h << OWNER_BITS) >> 32), 0, lv.bits);
                    put(i0, n, (uint32_t)((h << OWNER_BITS) >> 32), d);
                }, &hi, &lo);
            };
            if constexpr (DESC) {
                const uint32_t m = desc[g];
                const uint32_t nr = m >> 16;
                if (nr > (uint32_t)SKD) {
                    recompute();
                } else if (nr) {
                    // the 64 bases from the segment's first window on (as seg_runs forms them)
                    const int p0 = q.sgm * PK;
                    int v = read_nk(s, q.r) - p0;
                    v = v > PK ? PK : v;
                    const int sh = 2 * ((s.fc + p0) & 31);
                    hi = sh ? (w[0] << sh) | (w[1] >> (64 - sh)) : w[0];
                    lo = sh ? (w[1] << sh) | (w[2] >> (64 - sh)) : w[1];
                    // all header words of the segment at once: SKD independent loads in flight instead of
                    // one exposed memory latency per run
                    uint32_t hd[SKD];
#pragma unroll
                    for (int r = 0; r < SKD; r++) hd[r] = (uint32_t)r < nr ? desc[(int64_t)(1 + r) * s.n_threads + g] : 0u;
                    uint32_t own[2] = {0, 0};
                    if (lv.n_owners > 0) {
                        own[0] = desc[(int64_t)(1 + SKD) * s.n_threads + g];
                        own[1] = desc[(int64_t)(2 + SKD) * s.n_threads + g];
                    }
                    uint32_t starts = m & 0xffffu;
#pragma unroll
                    for (int r = 0; r < SKD; r++) {
                        if (starts) {
                            const int i0 = __ffs((int)starts) - 1;
                            starts &= starts - 1;
                            const int i1 = starts ? __ffs((int)starts) - 1 : v;
                            put(i0, i1 - i0, hd[r], lv.n_owners > 0 ? (own[r >> 2] >> (8 * (r & 3))) & 255u : rec_digit(hd[r], 0, lv.bits));
                        }
                    }
                }
            } else {
                recompute();
            }
            q.r += dq; q.sgm += dr;
            if (q.sgm >= s.segs) { q.sgm -= s.segs; q.r++; }
        }
        __syncthreads();
        drain(false);
        __syncthreads();
    }
    drain(true);
}
#undef buf
#undef tail
#undef head

// ---- level 1 in ONE sweep over the reads.  The two-pass form above computes every minimiser in the histogram pass
// and carries 5.4 GB of run descriptors to the scatter so as not to compute them twice.  Here nothing is counted first:
// a sampled histogram (one tile of 1024 segments in every `sample`) sizes a REGION per bucket with room to spare, and a
// workgroup takes its output positions from the bucket's cursor in EXTENTS of OSE records (one global atomic per
// extent).  A workgroup always owns the extent it is filling and the next one, so the write-combining rings and the
// direct stores of a busy round never wait for an allocation; extents change hands only at the round's barrier.
// What a workgroup leaves unused of its last two extents are HOLES; k_fix_holes moves the records at the end of every
// bucket into the holes before it, so the next level reads a gap-free [seg_begin, seg_end) per bucket.  A region that
// runs out (a sample that missed the skew) raises `overflow` and the caller falls back to the two-pass form.
#ifndef RFX_DRAIN_UNROLL
#define RFX_DRAIN_UNROLL 1
#endif
#ifndef RFX_OS_A32
#define RFX_OS_A32 4                 // records per aligned burst of the 16-slot rings (4 = 64 bytes, 8 = 128)
#endif
#ifndef RFX_OS_T32
#define RFX_OS_T32 512               // threads per workgroup of the sweep with 32 windows per thread
#endif
#ifndef RFX_OS_LPB
#define RFX_OS_LPB 4                 // lanes that drain a bin there (8: a lane per ring slot)
#endif
#ifndef RFX_OS_LPB_WIDE
#define RFX_OS_LPB_WIDE 4            // ... of the 32-byte records of k = 33..63 (8: part1 9.95 instead of 9.80 ms)
#endif
constexpr int OSE_MIN = 64, OSE_MAX = 256;   // records per extent: a round must not put more than one extent of one
                                         // workgroup into one bucket (5 records on average at 512 buckets); the sampled
                                         // histogram picks 64, 128 or 256 by the busiest bucket, or no sweep at all
constexpr int OS_CSTRIDE = 16;           // cursors 128 bytes apart: one atomic unit each
constexpr int OS_MAXG = 512;             // workgroups of the sweep
constexpr int OS_HOLES = 3;              // extents a workgroup has in hand per bucket: filling, next, requested
struct OneSweep {
    const uint64_t *reg_start;           // [nb + 1] first record of every region (multiples of the extent); [nb] = a dump extent
    const uint32_t *reg_cap;             // [nb] records a region holds
    unsigned long long *cursor;          // [nb * OS_CSTRIDE] next record to hand out (absolute; starts at reg_start)
    uint64_t *holes;                     // [nb][OS_HOLES * G] (absolute start << 16) | length, 0 = none
    int *overflow;
    uint64_t total;                      // records allocated (regions + the dump extent)
    unsigned long long *tile_counter;    // tiles of SKT segments handed out beyond every workgroup's first
    uint32_t ose, ose_shift;             // records per extent (a power of two)
};

template <int W, int SEG = 16>
__global__ __launch_bounds__(SKT) void k_sk_sample_hist(ReadSrc s, Level lv, int sample, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[1 << MAX_BITS];
    const int nb = 1 << lv.bits;
    for (int i = threadIdx.x; i < nb; i += SKT) h[i] = 0;
    __syncthreads();
    const int64_t ntile = (s.n_threads + SKT - 1) / SKT;
    for (int64_t T = (int64_t)blockIdx.x * sample; T < ntile; T += (int64_t)gridDim.x * sample) {
        const int64_t g = T * SKT + threadIdx.x;
        if (g < s.n_threads) {
            SegPos q;
            q.r = g / s.segs;
            q.sgm = (int)(g - q.r * s.segs);
            uint64_t w[3], hi, lo;
            seg_load<SEG>(s, q, w);
            seg_runs<W, SK_HIST_RUNLOOP, SEG>(s, read_nk(s, q.r), q.sgm, w, [&](int, int, uint32_t canon) { atomicAdd(&h[sk_digit(canon, lv)], 1u); }, &hi, &lo);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += SKT) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// regions from the sampled histogram: estimate + six standard deviations of the sample + 1/64 + what the workgroups
// hold in hand; totals[0] = records to allocate (regions + the dump extent), totals[2] = records per extent: four times
// what a tile is expected to put into the busiest bucket, or 0 = no sweep (low-complexity reads on one minimiser)
__global__ __launch_bounds__(1024) void k_plan_regions(const unsigned long long *__restrict__ hist, int nb, double scale, int G, int cap_pct,
                                                       double tiles,
                                                       uint64_t *__restrict__ reg_start, uint32_t *__restrict__ reg_cap,
                                                       unsigned long long *__restrict__ cursor, unsigned long long *__restrict__ totals) {
    __shared__ uint64_t caps[1 << MAX_BITS];
    __shared__ unsigned long long busiest;
    __shared__ uint32_t ose_s;
    const int d = threadIdx.x;
    if (d == 0) busiest = 0;
    __syncthreads();
    if (d < nb) atomicMax(&busiest, hist[d]);
    __syncthreads();
    if (d == 0) {
        const double per_tile = (double)busiest * scale / tiles;       // records a tile puts into the busiest bucket
        ose_s = per_tile <= 16.0 ? 64u : per_tile <= 32.0 ? 128u : per_tile <= 64.0 ? 256u : 0u;
    }
    __syncthreads();
    const uint32_t E = ose_s ? ose_s : OSE_MIN;
    if (d < nb) {
        const double c = (double)hist[d];
        const double est = c * scale;
        double room = (est + 6.0 * scale * sqrt(c + 1.0) + est / 64.0 + 1024.0) * (double)cap_pct / 100.0;
        uint64_t cap = (uint64_t)room + (uint64_t)G * OS_HOLES * E;
        cap = (cap + E - 1) / E * E;
        if (cap > 0xFFFFFF00ULL) cap = 0xFFFFFF00ULL / E * E;          // bucket-local positions are 32-bit
        caps[d] = cap;
        reg_cap[d] = (uint32_t)cap;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t at = 0;
        for (int i = 0; i < nb; i++) { reg_start[i] = at; cursor[(size_t)i * OS_CSTRIDE] = at; at += caps[i]; }
        reg_start[nb] = at;
        totals[0] = at + E;
        totals[1] = 0;
        totals[2] = ose_s;
    }
}

// WIDE: the 32-byte records of the k = 33..63 path (as in k_sk_scatter)
// SEG: windows a thread owns (seg_runs).  32: one workgroup per CU (the minima of 32 windows and 50 m-mers live in
// registers) with rings of OSB = 16 records, since a round now brings a bucket 8 records on average.
// (first form: 1024 threads, one workgroup per CU, rings of 16: no faster than 16 windows per thread -- the instructions
// saved went into the CU standing still at the round's two barriers.  Hence T = 512 threads: two workgroups per CU again,
// a tile brings a bucket what it did before, and the rings stay at 8 slots.)
template <int SEG, bool WIDE> struct OsGeo {
    static constexpr int T = SEG > 16 && !WIDE ? RFX_OS_T32 : SKT;           // threads per workgroup = segments per tile
    static constexpr int B = SEG > 16 && !WIDE && T == SKT ? 16 : SKB, A = SEG > 16 && !WIDE && T == SKT ? RFX_OS_A32 : SKA;
};
template <int W, bool WIDE = false, int SEG = 16>
__global__ __launch_bounds__((OsGeo<SEG, WIDE>::T), WIDE ? 4 : SEG > 16 ? (OsGeo<SEG, WIDE>::T == SKT ? 4 : OsGeo<SEG, WIDE>::T / 128) : 8) void k_sk_onesweep(ReadSrc s, Level lv, OneSweep os,
                                                               std::conditional_t<WIDE, WRec, Rec> *__restrict__ out) {
    using RT = std::conditional_t<WIDE, WRec, Rec>;
    constexpr int SKB = OsGeo<SEG, WIDE>::B, SKA = OsGeo<SEG, WIDE>::A;      // (shadow the file-wide ring geometry
    constexpr int SKT = OsGeo<SEG, WIDE>::T;                                  //  and workgroup size)
    extern __shared__ __attribute__((aligned(32))) unsigned char sk_smem[];
    const int nb = 1 << lv.bits;
#define buf ((RT *)sk_smem)
#define tail ((uint32_t *)(sk_smem + (size_t)nb * SKB * sizeof(RT)))        /* records given a (workgroup-local) position */
#define head (tail + nb)                                                    /* ... of them stored */
#define cstart (tail + 2 * nb)                                              /* local position the current extent starts at */
#define cbase (tail + 3 * nb)                                               /* the current extent (its number: record / OSE) */
#define nbase (tail + 4 * nb)                                               /* the next one */
#define pbase (tail + 5 * nb)                                               /* the one after that, or NONE while it is on order */
    constexpr uint32_t DUMP = 0xFFFFFFFFu;
    const uint32_t OSE = os.ose, OSH = os.ose_shift;
    const uint64_t dump_at = os.total - OSE;
    // An extent off bucket d's cursor.  Only the END OF THE ALLOCATION is checked here (no load on this path): an extent
    // past the bucket's own region lands in its neighbour's, which k_fix_holes sees from the cursor and declares the
    // whole sweep void.
    auto grab = [&](int d) __attribute__((always_inline)) -> uint32_t {
        const unsigned long long b = atomicAdd(&os.cursor[(size_t)d * OS_CSTRIDE], (unsigned long long)OSE);
        return b + OSE <= dump_at ? (uint32_t)(b >> OSH) : DUMP;
    };
    // where local position v of a bucket lives (v - cs < 2 * OSE)
    auto phys = [&](uint32_t v, uint32_t cs, uint32_t cb, uint32_t nx) __attribute__((always_inline)) -> uint64_t {
        const uint32_t o = v - cs;
        const uint32_t b = o < (uint32_t)OSE ? cb : nx;
        return (b == DUMP ? dump_at : (uint64_t)b << OSH) + (o & (OSE - 1));
    };
    // The extent after the next one is requested a round ahead: thread d asks for bucket d's at the top of a round and
    // files it (pbase) at the end of the round's arithmetic, so the atomic has the whole round to come back -- waited
    // for inside the drain it would hold up the wave's stores behind it, round after round.
    constexpr uint32_t NONE = 0xFFFFFFFEu;
    for (int i = threadIdx.x; i < nb; i += SKT) {
        tail[i] = 0; head[i] = 0; cstart[i] = 0;
        cbase[i] = grab(i);
        nbase[i] = grab(i);
        pbase[i] = grab(i);
    }
    __syncthreads();
    // LPB lanes drain a bin, each every LPB-th record of what waits: eight lanes (one record each) in rounds 1-2; four
    // with 32 windows per thread -- a bin's bookkeeping is done by half as many lanes, and four lanes are one aligned
    // 64-byte line, the unit the rings hand out anyway
    constexpr int LPB = SEG > 16 && !WIDE ? RFX_OS_LPB : WIDE ? RFX_OS_LPB_WIDE : SKB;
    auto drain = [&](bool final) __attribute__((always_inline)) {
#pragma unroll RFX_DRAIN_UNROLL
        for (int d = threadIdx.x / LPB; d < nb; d += SKT / LPB) {
            const int j = threadIdx.x % LPB;
            const uint32_t h = head[d], t = tail[d], cs = cstart[d], cb = cbase[d], nx = nbase[d];
            uint32_t e, nh;
            if (t - h > (uint32_t)SKB) { e = h + SKB; nh = t; }               // the excess went out directly
            else {
                e = final ? t : (t & ~(uint32_t)(SKA - 1));
                if (e < h) e = h;
                nh = e;
            }
#ifdef RFX_OS_ABL_NOSTORE
            { const uint32_t g = h + j; if (g < e && g == 0xFFFFFFF0u) out[phys(g, cs, cb, nx)] = buf[(size_t)d * SKB + (g & (SKB - 1))]; }
#else
#pragma unroll
            for (int q = 0; q < SKB / LPB; q++) {
                const uint32_t g = h + j + q * LPB;
                if (g < e) out[phys(g, cs, cb, nx)] = buf[(size_t)d * SKB + (g & (SKB - 1))];
            }
#endif
            if (j == 0) {
                head[d] = nh;
                // extents that lie wholly behind the stored position are done with: move up (the 8 lanes of this
                // bucket read the old values above, in lock-step)
                if (nh - cs >= (uint32_t)OSE) {
                    uint32_t c = cs, b0 = cb, b1 = nx, b2 = pbase[d];
                    while (nh - c >= (uint32_t)OSE) { c += OSE; b0 = b1; b1 = b2 != NONE ? b2 : grab(d); b2 = NONE; }
                    cstart[d] = c; cbase[d] = b0; nbase[d] = b1; pbase[d] = b2;
                }
            }
        }
    };
    // Tiles of SKT segments are handed out by a global counter (the first one is the workgroup's number): the
    // workgroups stay as few as the holes allow and still finish together.  The next tile is asked for at the top of a
    // round, like the extents.
    __shared__ long long tile_lds[2];
    const int64_t ntile = (s.n_threads + SKT - 1) / SKT;
    const int my_d = (int)threadIdx.x;                  // the bucket this thread keeps supplied (nb <= SKT)
    int64_t T = blockIdx.x;
    for (int rnd = 0; T < ntile; rnd++) {
        long long t_next = 0;
        if (threadIdx.x == SKT - 1) t_next = (long long)gridDim.x + (long long)atomicAdd(os.tile_counter, 1ULL);
        const bool ask = my_d < nb && pbase[my_d] == NONE;
        uint32_t req = DUMP;
        if (ask) req = grab(my_d);
        const int64_t g = T * SKT + threadIdx.x;
        if (g < s.n_threads) {
            SegPos q;
            q.r = g / s.segs;
            q.sgm = (int)(g - q.r * s.segs);
            uint64_t w[3], hi = 0, lo = 0;
            seg_load<SEG>(s, q, w);
            seg_runs<W, SK_SCATTER_RUNLOOP, SEG>(s, read_nk(s, q.r), q.sgm, w, [&](int i0, int n, uint32_t canon) {
                const uint64_t hh = mmer_hash64(canon);
                const uint32_t hdr = (uint32_t)((hh << OWNER_BITS) >> 32);
                // (by owner: the owner's bucket is 2^sub_bits bins here, so that a round still brings a bin a handful of
                // records -- the receiver does not care which of them a record came through)
                const unsigned d = lv.n_owners > 0 ? (((unsigned)__umul64hi(hh, (uint64_t)lv.n_owners) << lv.sub_bits) | (hdr >> (32 - lv.sub_bits)))
                                                   : rec_digit(hdr, 0, lv.bits);
                RT r;
                if constexpr (WIDE) {
                    // 96 bases from the first base of the run's first k-mer (see k_sk_scatter)
                    const uint64_t *gw = s.words + q.r * s.wpr;
                    const int p = s.fc - s.wfl + q.sgm * SEG + i0;
                    const int wi = p >> 5, sft = 2 * (p & 31), last = s.wpr - 1;
                    const uint64_t a0 = gw[wi < last ? wi : last], a1 = gw[wi + 1 < last ? wi + 1 : last];
                    const uint64_t a2 = gw[wi + 2 < last ? wi + 2 : last], a3 = gw[wi + 3 < last ? wi + 3 : last];
                    r.b0 = sft ? (a0 << sft) | (a1 >> (64 - sft)) : a0;
                    r.b1 = sft ? (a1 << sft) | (a2 >> (64 - sft)) : a1;
                    r.b2 = sft ? (a2 << sft) | (a3 >> (64 - sft)) : a2;
                    r.hd = ((uint64_t)(n - 1) << 32) | (uint64_t)hdr;
                } else {
                    const int sft = 2 * i0;
                    r.w0 = sft ? (hi << sft) | (lo >> (64 - sft)) : hi;
                    r.w1 = ((lo << sft) & 0xFFFFFFF000000000ULL) | ((uint64_t)(n - 1) << 32) | (uint64_t)hdr;
                }
                const uint32_t pos = atomicAdd(&tail[d], 1u);
                if (pos - head[d] < (uint32_t)SKB) buf[(size_t)d * SKB + (pos & (SKB - 1))] = r;
                else {
                    const uint32_t cs = cstart[d];
                    if (pos - cs < 2u * OSE) out[phys(pos, cs, cbase[d], nbase[d])] = r;
                    else *os.overflow = 1;             // more than the two extents in hand take: the caller starts over
                }
            }, &hi, &lo);
        }
        if (threadIdx.x == SKT - 1) tile_lds[rnd & 1] = t_next;
        if (ask) pbase[my_d] = req;
        __syncthreads();
        drain(false);
        T = tile_lds[rnd & 1];
        __syncthreads();
    }
    drain(true);
    __syncthreads();
    // what is left of the extents in hand (after the last drain: head == tail, tail - cstart < OSE)
    for (int d = threadIdx.x; d < nb; d += SKT) {
        const uint32_t used = tail[d] - cstart[d], cb = cbase[d], nx = nbase[d], px = pbase[d];
        uint64_t *hl = os.holes + ((size_t)d * gridDim.x + blockIdx.x) * OS_HOLES;
        hl[0] = cb != DUMP && used < OSE ? ((((uint64_t)cb << OSH) + used) << 16) | (uint64_t)(OSE - used) : 0;
        hl[1] = nx != DUMP ? (((uint64_t)nx << OSH) << 16) | (uint64_t)OSE : 0;
        hl[2] = px != DUMP && px != NONE ? (((uint64_t)px << OSH) << 16) | (uint64_t)OSE : 0;
    }
}
#undef buf
#undef tail
#undef head
#undef cstart
#undef cbase
#undef nbase
#undef pbase

// One workgroup per bucket: the bucket's cursor says how far extents were handed out (alloc), the holes sum to H, so
// the bucket holds size = alloc - H records and the last H positions [size, alloc) -- the tail zone -- hold as many
// records as there are hole positions before `size`.  Tail record i goes to hole position i.
constexpr int FH_T = 1024;
template <class RT>
__global__ __launch_bounds__(FH_T) void k_fix_holes(OneSweep os, int G, RT *__restrict__ out, uint64_t *__restrict__ seg_begin,
                                                   uint64_t *__restrict__ seg_end, unsigned long long *__restrict__ totals) {
    constexpr int NHMAX = 2048;                                // holes of a bucket (>= OS_HOLES * OS_MAXG, a multiple of FH_T)
    static_assert(NHMAX >= OS_HOLES * OS_MAXG && NHMAX % FH_T == 0, "holes per bucket");
    constexpr int NWMAX = NHMAX * OSE_MAX / 32;                // words of the tail-zone bitmap
    __shared__ uint32_t hstart[NHMAX], hoff[NHMAX + 1];
    __shared__ uint32_t bitmap[NWMAX];
    __shared__ uint32_t wsum[FH_T / 64];
    const int d = blockIdx.x;
    const int NH = OS_HOLES * G;
    const uint64_t base = os.reg_start[d];
    const unsigned long long alloc = os.cursor[(size_t)d * OS_CSTRIDE] - base;
    if (alloc > (unsigned long long)os.reg_cap[d]) {           // ran over its region: the sweep is void
        if (threadIdx.x == 0) { seg_begin[d] = base; seg_end[d] = base; *os.overflow = 1; }
        return;
    }
    auto block_excl_scan = [&](uint32_t v, uint32_t *total) __attribute__((always_inline)) -> uint32_t {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t b = 0, tot = 0;
        for (int i = 0; i < FH_T / 64; i++) { const uint32_t sv = wsum[i]; if (i < wave) b += sv; tot += sv; }
        __syncthreads();
        *total = tot;
        return b + x - v;
    };
    constexpr int HPT = NHMAX / FH_T;
    uint32_t hs[HPT], hl[HPT];
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < HPT; r++) {
        const int t = threadIdx.x * HPT + r;
        const uint64_t x = t < NH ? os.holes[(size_t)d * NH + t] : 0;
        hl[r] = (uint32_t)(x & 0xFFFFu);
        hs[r] = hl[r] ? (uint32_t)((x >> 16) - base) : 0u;
        mine += hl[r];
    }
    uint32_t H;
    block_excl_scan(mine, &H);
    const uint32_t size = (uint32_t)alloc - H;
    const uint32_t nw = (H + 31) / 32;                         // words of the tail zone's bitmap (<= NWMAX)
    const uint32_t wpt = (nw + FH_T - 1) / FH_T;               // ... a thread looks after: [tid * wpt, tid * wpt + wpt)
    for (uint32_t i = threadIdx.x; i < nw; i += FH_T) bitmap[i] = 0;
    __syncthreads();
    // hole positions before `size` (to be filled) / inside the tail zone (marked)
    uint32_t fill[HPT], fsum = 0;
#pragma unroll
    for (int r = 0; r < HPT; r++) {
        fill[r] = hs[r] < size ? (hs[r] + hl[r] <= size ? hl[r] : size - hs[r]) : 0u;
        fsum += fill[r];
        for (uint32_t p = hs[r] + fill[r]; p < hs[r] + hl[r]; p++) atomicOr(&bitmap[(p - size) >> 5], 1u << ((p - size) & 31));
    }
    uint32_t F;
    uint32_t fo = block_excl_scan(fsum, &F);
#pragma unroll
    for (int r = 0; r < HPT; r++) {
        const int t = threadIdx.x * HPT + r;
        hstart[t] = hs[r]; hoff[t] = fo; fo += fill[r];
    }
    if (threadIdx.x == 0) hoff[NHMAX] = F;
    __syncthreads();
    // records of the tail zone, in position order
    auto valid_word = [&](uint32_t w) __attribute__((always_inline)) -> uint32_t {
        if (w >= nw) return 0u;
        const uint32_t lim = H - w * 32 >= 32 ? 0xFFFFFFFFu : (1u << (H - w * 32)) - 1u;
        return ~bitmap[w] & lim;
    };
    uint32_t vsum = 0;
    for (uint32_t r = 0; r < wpt; r++) vsum += (uint32_t)__popc(valid_word(threadIdx.x * wpt + r));
    uint32_t V;
    uint32_t vo = block_excl_scan(vsum, &V);
    for (uint32_t r = 0; r < wpt; r++) {
        const uint32_t w = threadIdx.x * wpt + r;
        uint32_t m = valid_word(w);
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            const uint32_t i = vo++;                           // i-th record of the tail zone -> i-th hole position
            if (i < F) {
                int lo = 0, hi = NHMAX;                        // the entry with hoff <= i < hoff of the next
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (hoff[mid] <= i) lo = mid; else hi = mid; }
                out[base + hstart[lo] + (i - hoff[lo])] = out[base + size + w * 32 + b];
            }
        }
    }
    if (threadIdx.x == 0) {
        seg_begin[d] = base; seg_end[d] = base + size;
        atomicAdd(&totals[1], (unsigned long long)size);
        if (V != F) *os.overflow = 2;                          // cannot happen: the books of the sweep do not balance
    }
}

// 16-byte elements of the k > 32 path (two-word canonical k-mers, rfx_wide.hip) go through the same
// level kernels; their digits come from a hash of both words
__device__ __forceinline__ uint64_t wide_hash(uint64_t hi, uint64_t lo) {
    uint64_t x = hi ^ (lo * 0x9E3779B97F4A7C15ULL);
    x *= 0xD6E8FEB86659FD93ULL;
    return x ^ (x >> 32);
}
// (the top OWNER_BITS of the hash pick the owning GPU -- mulhi(hash, n_owners) -- and are skipped by the
// local digits, as on the k <= 31 path)
// MODE 0: super-k-mer records (digit from the header), 1: two-word k-mers (hash of both words),
// 2: (k-mer, partial count) pairs (kmer_hash of the key; with lv.n_owners > 0 the digit is the owner)
template <int MODE> struct LevelElem { using T = Rec; };
template <> struct LevelElem<3> { using T = WRec; };          // MODE 3: super-k-mer records of the k = 33..63 path
template <int MODE, class E>
__device__ __forceinline__ unsigned level_digit(const E &r, int used, const Level &lv) {
    const int bits = lv.bits;
    if constexpr (MODE == 3) return rec_digit((uint32_t)r.hd, used, bits);
    else if constexpr (MODE == 1) return bits ? (unsigned)(((wide_hash(r.w0, r.w1) << OWNER_BITS) << used) >> (64 - bits)) : 0u;
    else if constexpr (MODE == 2) {
        if (lv.n_owners > 0) return (unsigned)__umul64hi(kmer_hash(r.w0), (uint64_t)lv.n_owners);
        return bits ? (unsigned)((local_hash(r.w0) << used) >> (64 - bits)) : 0u;
    } else return rec_digit(rec_hdr(r), used, bits);
}
template <int MODE>
__device__ __forceinline__ int level_bins(const Level &lv) {
    if constexpr (MODE == 2) return lv.n_owners > 0 ? lv.n_owners : 1 << lv.bits;
    else return 1 << lv.bits;
}
__device__ __forceinline__ unsigned wide_level1_digit(const Rec &r, const Level &lv) {
    if (lv.n_owners > 0) return (unsigned)__umul64hi(wide_hash(r.w0, r.w1), (uint64_t)lv.n_owners);
    return level_digit<1>(r, 0, lv);
}

// levels >= 2 on records: virtual workgroups as for k-mers, digit from the record header
template <int MODE>
__global__ __launch_bounds__(PT) void k_rec_hist(const typename LevelElem<MODE>::T *__restrict__ recs, VbMap m, Level lv, int used,
                                                 uint32_t *__restrict__ table) {
    __shared__ uint32_t h[1 << MAX_BITS];
    VbPos q;
    if (!locate_vb(m, blockIdx.x, &q)) return;
    const int nb = level_bins<MODE>(lv);
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    for (uint64_t i = q.begin + threadIdx.x; i < q.end; i += PT) {
        const typename LevelElem<MODE>::T r = recs[i];
        if constexpr (MODE == 2) { if (r.w1 == 0) continue; }   // a hole of the combine output (count 0)
        atomicAdd(&h[level_digit<MODE>(r, used, lv)], 1u);
    }
    __syncthreads();
    const int64_t tb = (int64_t)nb * (int64_t)m.vb_start[q.p];
    for (int i = threadIdx.x; i < nb; i += PT) table[tb + (int64_t)i * q.G + q.g] = h[i];
}

template <int MODE>
__global__ __launch_bounds__(PT) void k_rec_scatter(const typename LevelElem<MODE>::T *__restrict__ recs, VbMap m, Level lv, int used,
                                                    const uint64_t *__restrict__ scanned,
                                                    typename LevelElem<MODE>::T *__restrict__ out) {
    __shared__ unsigned long long cur[1 << MAX_BITS];
    VbPos q;
    if (!locate_vb(m, blockIdx.x, &q)) return;
    const int nb = level_bins<MODE>(lv);
    const int64_t tb = (int64_t)nb * (int64_t)m.vb_start[q.p];
    for (int i = threadIdx.x; i < nb; i += PT) cur[i] = scanned[tb + (int64_t)i * q.G + q.g];
    __syncthreads();
    for (uint64_t i = q.begin + threadIdx.x; i < q.end; i += PT) {
        const typename LevelElem<MODE>::T r = recs[i];
        if constexpr (MODE == 2) { if (r.w1 == 0) continue; }
        out[atomicAdd(&cur[level_digit<MODE>(r, used, lv)], 1ULL)] = r;
    }
}

// Write-combining scatter.  A 16-byte store per record into 512+ private streams runs the memory
// system at a third of its rate (tools/scatter_bench.hip: 10.1 ms for 10.7 GB read + scattered
// 16-B writes, 3.7 ms when every stream is written in aligned 128-byte bursts).  So each stream
// gets a ring of B record slots in LDS, indexed by the record's final position in the output, and
// only whole aligned lines of B/2 records leave the ring (partial lines stay for the next round;
// a bin that receives more than the ring holds in one round writes the excess directly).
constexpr int WCT = 1024;             // threads per workgroup (one workgroup per CU: the rings fill the LDS)
constexpr int WC_PER = 4;             // records per thread per round

template <int B, int MODE>
__global__ __launch_bounds__(WCT) void k_rec_scatter_wc(const typename LevelElem<MODE>::T *__restrict__ recs, VbMap m, Level lv,
                                                        int used, const uint64_t *__restrict__ scanned,
                                                        typename LevelElem<MODE>::T *__restrict__ out) {
    using Rec = typename LevelElem<MODE>::T;          // (16-byte elements, or the 32-byte records of MODE 3)
    extern __shared__ __attribute__((aligned(32))) unsigned char wc_smem[];
    constexpr int A = B / 2;          // records per aligned output line (128 B for B = 16)
    VbPos q;
    if (!locate_vb(m, blockIdx.x, &q)) return;
    const int nb = level_bins<MODE>(lv);
    Rec *buf = (Rec *)wc_smem;
    unsigned long long *tail = (unsigned long long *)(buf + (size_t)nb * B);
    unsigned long long *head = tail + nb;
    const int64_t tb = (int64_t)nb * (int64_t)m.vb_start[q.p];
    for (int i = threadIdx.x; i < nb; i += WCT) tail[i] = head[i] = scanned[tb + (int64_t)i * q.G + q.g];
    __syncthreads();
    // B adjacent lanes drain one bin: everything up to the last aligned boundary (all of it at the end)
    auto drain = [&](bool final) __attribute__((always_inline)) {
#pragma unroll RFX_DRAIN_UNROLL
        for (int d = threadIdx.x / B; d < nb; d += WCT / B) {
            const int j = threadIdx.x % B;
            const unsigned long long h = head[d], t = tail[d];
            unsigned long long e, nh;
            if (t - h > (unsigned long long)B) { e = h + B; nh = t; }         // the excess went out directly
            else {
                e = final ? t : (t & ~(unsigned long long)(A - 1));
                if (e < h) e = h;
                nh = e;
            }
            const unsigned long long g = h + j;
            if (g < e) out[g] = buf[(size_t)d * B + (g & (B - 1))];
            if (j == 0) head[d] = nh;
        }
    };
    for (uint64_t base = q.begin; base < q.end; base += (uint64_t)WCT * WC_PER) {
        Rec r[WC_PER];
        // (unconditional loads -- a lane past the end re-reads the last element: with the assignment under a condition
        // the 32-byte records of MODE 3 stayed in scratch memory, 128 bytes per lane)
#pragma unroll
        for (int i = 0; i < WC_PER; i++) {
            const uint64_t idx = base + (uint64_t)i * WCT + threadIdx.x;
            r[i] = recs[idx < q.end ? idx : q.end - 1];
        }
#pragma unroll
        for (int i = 0; i < WC_PER; i++) {
            const uint64_t idx = base + (uint64_t)i * WCT + threadIdx.x;
            bool live = idx < q.end;
            if constexpr (MODE == 2) live = live && r[i].w1 != 0;
            if (live) {
                const unsigned d = level_digit<MODE>(r[i], used, lv);
                const unsigned long long g = atomicAdd(&tail[d], 1ULL);
                if (g - head[d] < (unsigned long long)B) buf[(size_t)d * B + (g & (B - 1))] = r[i];
                else out[g] = r[i];
            }
        }
        __syncthreads();
        drain(false);
        __syncthreads();
    }
    drain(true);
}

// ---- a FIRST level over records without a histogram pass (round 4).  What a rank receives from the exchange is one unpartitioned
// array, and its first level re-read all of it for an exact histogram (hist1 of the multi-GPU step: 2.5 of 39.6 ms) only to give
// every (virtual workgroup, digit) a private output range.  Here a workgroup takes a CHUNK of the array, counts the chunk's digits in
// LDS, CLAIMS a contiguous range per digit with one atomic add on the bucket's cursor (256 atomics a chunk), and scatters the chunk
// -- read a second time, from the cache it has just been brought into -- through the same write-combining rings as the exact form, to
// exact positions: no extents, no holes.  What the global histogram was needed for, the bucket STARTS, a sample gives with room to
// spare (k_rec_sample_hist / k_claim_plan: regions as for level 1's sweep); a bucket that outgrows its region raises a flag, nothing
// is written beyond a region, and the exact form runs.  Buckets come out as [begin, end) with the slack behind them.
// (First form, measured and dropped: level 1's extent machinery with records as input -- as much slower in the scatter, 2.3 ms a
// step, as the histogram it saved.)
constexpr int RCL_TILES = 1;               // tiles of WCT x WC_PER records a workgroup counts, claims for and scatters at a time
template <int MODE>
__global__ __launch_bounds__(PT) void k_rec_sample_hist(const typename LevelElem<MODE>::T *__restrict__ recs, uint64_t n, Level lv, int used, int sample,
                                                        unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[1 << MAX_BITS];
    const int nb = 1 << lv.bits;
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    // chunks of PT records, every `sample`-th of them (cf. k_l2_sample: fine-grained, the array is ordered by sender and bin)
    for (uint64_t c = (uint64_t)blockIdx.x * sample; c * PT < n; c += (uint64_t)gridDim.x * sample) {
        const uint64_t i = c * PT + threadIdx.x;
        if (i < n) atomicAdd(&h[level_digit<MODE>(recs[i], used, lv)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += PT) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
// regions from the sampled histogram: estimate + six standard deviations of the sample + 1/64 + 1024 (a multiple of 8 records)
struct ClaimPlan { uint64_t *reg_start; /* [nb + 1] */ unsigned long long *cursor; /* [nb] records claimed */ int *overflow; };
__global__ __launch_bounds__(1024) void k_claim_plan(const unsigned long long *__restrict__ hist, int nb, double scale, ClaimPlan pl) {
    __shared__ uint64_t caps[1 << MAX_BITS];
    const int d = threadIdx.x;
    if (d < nb) {
        const double c = (double)hist[d], est = c * scale;
        const uint64_t cap = (uint64_t)(est + 6.0 * scale * sqrt(c + 1.0) + est / 64.0 + 1024.0);
        caps[d] = (cap + 7) & ~(uint64_t)7;
        pl.cursor[d] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t at = 0;
        for (int i = 0; i < nb; i++) { pl.reg_start[i] = at; at += caps[i]; }
        pl.reg_start[nb] = at;
        *pl.overflow = 0;
    }
}
__global__ void k_claim_ends(ClaimPlan pl, int nb, uint64_t *__restrict__ seg_begin, uint64_t *__restrict__ seg_end) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nb) return;
    const uint64_t b = pl.reg_start[d], cap = pl.reg_start[d + 1] - b;
    const unsigned long long c = pl.cursor[d];
    seg_begin[d] = b;
    seg_end[d] = b + (c <= cap ? c : cap);
    if (c > cap) *pl.overflow = 1;
}

template <int B, int MODE>
__global__ __launch_bounds__(WCT) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_rec_claim_scatter(const typename LevelElem<MODE>::T *__restrict__ recs, uint64_t n, Level lv, int used, ClaimPlan pl,
                                                           typename LevelElem<MODE>::T *__restrict__ out) {
    using Rec = typename LevelElem<MODE>::T;
    extern __shared__ __attribute__((aligned(32))) unsigned char wc_smem[];
    constexpr int A = B / 2;
    constexpr uint64_t CHUNK = (uint64_t)RCL_TILES * WCT * WC_PER;
    const int nb = 1 << lv.bits;
    Rec *buf = (Rec *)wc_smem;
    unsigned long long *tail = (unsigned long long *)(buf + (size_t)nb * B);
    unsigned long long *head = tail + nb;
    unsigned long long *lim = head + nb;
    uint32_t *h = (uint32_t *)(lim + nb);
    auto drain = [&](bool final) __attribute__((always_inline)) {
#pragma unroll RFX_DRAIN_UNROLL
        for (int d = threadIdx.x / B; d < nb; d += WCT / B) {
            const int j = threadIdx.x % B;
            const unsigned long long hd = head[d], lm = lim[d];
            unsigned long long t = tail[d];
            if (t > lm) t = lm;                                // (what ran over was not stored: k_claim_ends raises the flag)
            if (t < hd) t = hd;
            unsigned long long e, nh;
            if (t - hd > (unsigned long long)B) { e = hd + B; nh = t; }
            else {
                e = final ? t : (t & ~(unsigned long long)(A - 1));
                if (e < hd) e = hd;
                nh = e;
            }
            const unsigned long long g = hd + j;
            if (g < e) out[g] = buf[(size_t)d * B + (g & (B - 1))];
            if (j == 0) head[d] = nh;
        }
    };
    // a chunk = RCL_TILES tiles of WCT x WC_PER records, the tile held in registers between its count and its scatter
    for (uint64_t c0 = (uint64_t)blockIdx.x * CHUNK; c0 < n; c0 += (uint64_t)gridDim.x * CHUNK) {
        const uint64_t qe = c0 + CHUNK < n ? c0 + CHUNK : n;
        static_assert(RCL_TILES == 1, "the tile stays in registers: one tile per claim");
        for (int i = threadIdx.x; i < nb; i += WCT) h[i] = 0;
        __syncthreads();
        // (named 16-byte vectors -- a record is one or two of them -- not `Rec r[WC_PER]` and not 32-byte vectors: either way the
        // 32-byte records of MODE 3 lived in scratch memory across the two barriers, 128 bytes a lane)
        using V = ulonglong2;
        constexpr int NV = sizeof(Rec) / 16;
        static_assert(NV == 1 || NV == 2, "16- or 32-byte records");
        static_assert(WC_PER == 4, "four named records below");
        V q0, q1, q2, q3, w0 = {}, w1 = {}, w2 = {}, w3 = {};       // (w: the second half of a 32-byte record)
        unsigned d0, d1, d2, d3;
#define RCL_LOAD(i) { const uint64_t idx = c0 + (uint64_t)(i) * WCT + threadIdx.x; const uint64_t at = (idx < qe ? idx : qe - 1) * NV; \
            q##i = ((const V *)recs)[at]; if constexpr (NV == 2) w##i = ((const V *)recs)[at + 1]; }
        RCL_LOAD(0) RCL_LOAD(1) RCL_LOAD(2) RCL_LOAD(3)
#undef RCL_LOAD
        auto digit_of = [&](const V a, const V b2) __attribute__((always_inline)) -> unsigned {
            if constexpr (NV == 2) return level_digit<MODE>(Rec{a.x, a.y, b2.x, b2.y}, used, lv);
            else return level_digit<MODE>(Rec{a.x, a.y}, used, lv);
        };
#define RCL_COUNT(i) { const uint64_t idx = c0 + (uint64_t)(i) * WCT + threadIdx.x; d##i = digit_of(q##i, w##i); if (idx < qe) atomicAdd(&h[d##i], 1u); }
        RCL_COUNT(0) RCL_COUNT(1) RCL_COUNT(2) RCL_COUNT(3)
#undef RCL_COUNT
        __syncthreads();
        // one claim per digit the tile holds: [base, base + count) of the bucket's region; what does not fit is not written
        for (int d = threadIdx.x; d < nb; d += WCT) {
            const uint32_t c = h[d];
            const uint64_t rs = pl.reg_start[d], re = pl.reg_start[d + 1];
            unsigned long long base = 0;
            if (c) base = atomicAdd(&pl.cursor[d], (unsigned long long)c);
            tail[d] = head[d] = rs + base;
            lim[d] = re;
        }
        __syncthreads();
#define RCL_PUT(i) { const uint64_t idx = c0 + (uint64_t)(i) * WCT + threadIdx.x; \
            if (idx < qe) { \
                const unsigned d = d##i; \
                const unsigned long long g = atomicAdd(&tail[d], 1ULL); \
                if (g >= lim[d]) { /* the bucket's region is full */ } \
                else if (g - head[d] < (unsigned long long)B) { V *to = (V *)buf + ((size_t)d * B + (g & (B - 1))) * NV; to[0] = q##i; if constexpr (NV == 2) to[1] = w##i; } \
                else { V *to = (V *)out + g * NV; to[0] = q##i; if constexpr (NV == 2) to[1] = w##i; } \
            } }
        RCL_PUT(0) RCL_PUT(1) RCL_PUT(2) RCL_PUT(3)
#undef RCL_PUT
        __syncthreads();
        drain(true);
        __syncthreads();
    }
}

// ---- the LAST level in one sweep (round 4): no histogram pass.  What the exact form needs its histogram for -- private output
// ranges per (parent, digit, virtual workgroup) -- ONE workgroup per parent bucket does not need: it owns every child of its
// parent, so the children's cursors are its own (LDS), and a child only needs a REGION that is large enough.  How large, a
// SAMPLE says: every L2S_STRIDE-th chunk of 512 records of every parent is counted (k_l2_sample: 1/16 of the records read, 0.6 of 9.5 GB), a
// child that shows s records there holds about STRIDE x s, and its region takes STRIDE x (s + 6 sqrt(s + 1) + 2) -- six standard
// deviations of the sampling (+ 50 % at s = 150; a child is not a Poisson draw of records but a handful of minimiser SITES of
// ~1000 records each, which is why the sizes come from a sample and not from n / 2^bits).  A child that outgrows its region all
// the same raises a flag: writes stay inside the regions, the sweep is void, the exact form runs; so it does when the plan sees
// a parent much larger than the mean (skew: one workgroup per parent would not balance) or regions beyond the buffer.  The
// slack is address space, not traffic: the leaves read [begin, end) of every child.  Saves the re-read of k_rec_hist.
constexpr int L2S_STRIDE = 16;
struct L2Plan {
    const uint64_t *seg_begin, *seg_end; const uint64_t *child_start; int *flags;      // flags[0] not ok, [1] overflow beyond the spill
    // a child that outgrows its region does not void the sweep any more (a handful of 131,072 did in every generation of the
    // multi-GPU rehearsal once the parents were filled by claims): what does not fit goes to a SPILL list (record + child), and
    // k_l2_spill_fix moves those few children, region and spill, to free room behind the regions
    void *spill_rec; uint32_t *spill_child; unsigned long long *spill_cursor; uint32_t spill_cap;
    unsigned long long *tail_cursor; uint64_t out_cap;
};
template <int MODE>
__global__ __launch_bounds__(PT) void k_l2_sample(const typename LevelElem<MODE>::T *__restrict__ recs, L2Plan pl, Level lv, int used,
                                                  uint32_t *__restrict__ sampled) {
    __shared__ uint32_t h[1 << MAX_BITS];
    const int p = blockIdx.x, nb = 1 << lv.bits;
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    const uint64_t qb = pl.seg_begin[p], qe = pl.seg_end[p];
    // sampled chunks of this parent: chunk c (PT records, one per thread: 8 KB of 16-byte elements) with c % STRIDE == 0;
    // workgroup y of gridDim.y takes every gridDim.y-th of them.  (Until the multi-GPU rehearsal of round 4 the unit was a
    // tile of 8192 records: what a rank RECEIVES is ordered by sender and owner bin, a minimiser site lies in one of
    // them -- 11 tiles of a parent there -- and every 16th tile missed whole bins: 10 % of the children ran over.)
    for (uint64_t c = (uint64_t)blockIdx.y * L2S_STRIDE; c * PT < qe - qb; c += (uint64_t)gridDim.y * L2S_STRIDE) {
        const uint64_t i = qb + c * PT + threadIdx.x;
        if (i < qe) {
            const typename LevelElem<MODE>::T r = recs[i];
            if constexpr (MODE == 2) { if (r.w1 == 0) continue; }
            atomicAdd(&h[level_digit<MODE>(r, used, lv)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += PT) if (h[i]) atomicAdd(&sampled[(int64_t)p * nb + i], h[i]);
}
// region of a child from its sampled count (records, a multiple of 8: whole aligned lines)
__global__ void k_l2_caps(const uint32_t *__restrict__ sampled, int64_t nchild, uint32_t *__restrict__ ccap, int pct) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchild) return;
    const float s = (float)sampled[c];
    // (+ 32: a child of a few hundred records may show nothing in the sample -- measured: 10 of 262144 children of config 2 held
    // ~190 records with none sampled, against a floor of 128)
    uint32_t C = (uint32_t)((float)L2S_STRIDE * (s + 6.0f * sqrtf(s + 1.0f) + 32.0f));
    if (pct != 100) C = (uint32_t)((uint64_t)C * (uint64_t)pct / 100u);      // (RFX_L2_SQUEEZE: the tests make children spill)
    ccap[c] = (C + 7u) & ~7u;
}
__global__ __launch_bounds__(1024) void k_l2_check(L2Plan pl, int nseg, int64_t nchild, uint64_t out_cap) {
    __shared__ unsigned long long sum_n, mx_n;
    if (threadIdx.x == 0) { sum_n = 0; mx_n = 0; }
    __syncthreads();
    if ((int)threadIdx.x < nseg) {
        const unsigned long long n = pl.seg_end[threadIdx.x] - pl.seg_begin[threadIdx.x];
        atomicAdd(&sum_n, n);
        atomicMax(&mx_n, n);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long mean = sum_n / (unsigned long long)(nseg > 0 ? nseg : 1);
        pl.flags[0] = (pl.child_start[nchild] + 8ULL * pl.spill_cap + 65536ULL > out_cap || mx_n > mean + mean / 2 + 4096) ? 1 : 0;
        pl.flags[1] = 0;
        *pl.spill_cursor = 0;
        *pl.tail_cursor = (pl.child_start[nchild] + 7ULL) & ~7ULL;
    }
}

template <int B, int MODE>
__global__ __launch_bounds__(WCT) void k_rec_l2sweep(const typename LevelElem<MODE>::T *__restrict__ recs, L2Plan pl, Level lv, int used,
                                                     typename LevelElem<MODE>::T *__restrict__ out, uint64_t *__restrict__ leaf_end) {
    using Rec = typename LevelElem<MODE>::T;
    extern __shared__ __attribute__((aligned(32))) unsigned char wc_smem[];
    constexpr int A = B / 2;
    if (pl.flags[0]) return;                                   // (the plan saw skew or no room: the exact form runs)
    const int p = blockIdx.x;
    const int nb = 1 << lv.bits;
    const uint64_t qb = pl.seg_begin[p], qe = pl.seg_end[p];
    Rec *buf = (Rec *)wc_smem;
    unsigned long long *tail = (unsigned long long *)(buf + (size_t)nb * B);
    unsigned long long *head = tail + nb;
    unsigned long long *lim = head + nb;
    for (int i = threadIdx.x; i < nb; i += WCT) {
        tail[i] = head[i] = pl.child_start[(int64_t)p * nb + i];
        lim[i] = pl.child_start[(int64_t)p * nb + i + 1];
    }
    __syncthreads();
    auto drain = [&](bool final) __attribute__((always_inline)) {
#pragma unroll RFX_DRAIN_UNROLL
        for (int d = threadIdx.x / B; d < nb; d += WCT / B) {
            const int j = threadIdx.x % B;
            const unsigned long long h = head[d], lm = lim[d];
            unsigned long long t = tail[d];
            if (t > lm) t = lm;                                // (what ran over was not stored: the overflow flag is up)
            unsigned long long e, nh;
            if (t - h > (unsigned long long)B) { e = h + B; nh = t; }
            else {
                e = final ? t : (t & ~(unsigned long long)(A - 1));
                if (e < h) e = h;
                nh = e;
            }
            const unsigned long long g = h + j;
            if (g < e) out[g] = buf[(size_t)d * B + (g & (B - 1))];
            if (j == 0) head[d] = nh;
        }
    };
    for (uint64_t b0 = qb; b0 < qe; b0 += (uint64_t)WCT * WC_PER) {
        Rec r[WC_PER];
#pragma unroll
        for (int i = 0; i < WC_PER; i++) {
            const uint64_t idx = b0 + (uint64_t)i * WCT + threadIdx.x;
            r[i] = recs[idx < qe ? idx : qe - 1];
        }
#pragma unroll
        for (int i = 0; i < WC_PER; i++) {
            const uint64_t idx = b0 + (uint64_t)i * WCT + threadIdx.x;
            bool live = idx < qe;
            if constexpr (MODE == 2) live = live && r[i].w1 != 0;
            if (live) {
                const unsigned d = level_digit<MODE>(r[i], used, lv);
                const unsigned long long g = atomicAdd(&tail[d], 1ULL);
                if (g >= lim[d]) {                             // the child's region is full: the record waits in the spill list
                    const unsigned long long sk = atomicAdd(pl.spill_cursor, 1ULL);
                    if (sk < (unsigned long long)pl.spill_cap) { ((Rec *)pl.spill_rec)[sk] = r[i]; pl.spill_child[sk] = (uint32_t)(p * nb + (int)d); }
                    else pl.flags[1] = 1;                      // (more than the list takes: the sweep is void after all)
                }
                else if (g - head[d] < (unsigned long long)B) buf[(size_t)d * B + (g & (B - 1))] = r[i];
                else out[g] = r[i];
            }
        }
        __syncthreads();
        drain(false);
        __syncthreads();
    }
    drain(true);
    __syncthreads();
    // (a child that ran over keeps its raw count here: the sweep is void then, and RFX_TRACE reads the need from it)
    for (int d = threadIdx.x; d < nb; d += WCT) leaf_end[(int64_t)p * nb + d] = tail[d];
}

// the children that spilled (see L2Plan): each gets room behind the regions for everything it holds -- its full region and its
// spilled records -- and its [begin, end) is rewritten.  One workgroup: a few children, a few thousand records.
constexpr int L2F_T = 1024, L2F_SLOTS = 2048, L2F_MAXC = 1536;
template <class RT>
__global__ __launch_bounds__(L2F_T) void k_l2_spill_fix(L2Plan pl, uint64_t *__restrict__ cstart, uint64_t *__restrict__ lend, RT *__restrict__ out) {
    __shared__ uint32_t tkey[L2F_SLOTS], tcnt[L2F_SLOTS], tfill[L2F_SLOTS], tcap[L2F_SLOTS];
    __shared__ unsigned long long tbase[L2F_SLOTS];
    __shared__ uint32_t ndistinct;
    __shared__ int bad;
    if (pl.flags[0] || pl.flags[1]) return;
    const unsigned long long ns = *pl.spill_cursor;
    if (ns == 0) return;
    constexpr uint32_t NOKEY = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < L2F_SLOTS; i += L2F_T) { tkey[i] = NOKEY; tcnt[i] = 0; tfill[i] = 0; }
    if (threadIdx.x == 0) { ndistinct = 0; bad = 0; }
    __syncthreads();
    auto find = [&](uint32_t c, bool insert) __attribute__((always_inline)) -> int {
        uint32_t slot = (c * 0x9E3779B1u) >> 21;                 // 11 bits
        for (int probe = 0; probe < L2F_SLOTS; probe++) {
            uint32_t k = tkey[slot];
            if (k == c) return (int)slot;
            if (k == NOKEY) {
                if (!insert) return -1;
                k = atomicCAS(&tkey[slot], NOKEY, c);
                if (k == NOKEY) { atomicAdd(&ndistinct, 1u); return (int)slot; }
                if (k == c) return (int)slot;
            }
            slot = (slot + 1) & (L2F_SLOTS - 1);
        }
        return -1;
    };
    for (unsigned long long i = threadIdx.x; i < ns; i += L2F_T) {
        if (ndistinct > (uint32_t)L2F_MAXC) { bad = 1; break; }
        const int sl = find(pl.spill_child[i], true);
        if (sl < 0) { bad = 1; break; }
        atomicAdd(&tcnt[sl], 1u);
    }
    __syncthreads();
    if (bad || ndistinct > (uint32_t)L2F_MAXC) { if (threadIdx.x == 0) pl.flags[1] = 1; return; }
    __shared__ uint16_t used_slot[L2F_MAXC + 1];
    __shared__ uint32_t n_used;
    if (threadIdx.x == 0) n_used = 0;
    __syncthreads();
    for (int sl = threadIdx.x; sl < L2F_SLOTS; sl += L2F_T) {
        if (tkey[sl] == NOKEY) continue;
        const uint32_t c = tkey[sl];
        const uint64_t cap = pl.child_start[c + 1] - pl.child_start[c];          // (a child that spilled filled its region)
        tcap[sl] = (uint32_t)cap;
        const unsigned long long size = cap + tcnt[sl];
        const unsigned long long at = atomicAdd(pl.tail_cursor, (size + 7ULL) & ~7ULL);
        if (at + size > pl.out_cap) bad = 1;
        tbase[sl] = at;
        used_slot[atomicAdd(&n_used, 1u)] = (uint16_t)sl;
    }
    __syncthreads();
    if (bad) { if (threadIdx.x == 0) pl.flags[1] = 1; return; }
    for (uint32_t u = 0; u < n_used; u++) {                      // the children's regions, one after the other, by the whole workgroup
        const int sl = used_slot[u];
        const uint64_t from = pl.child_start[tkey[sl]];
        const unsigned long long to = tbase[sl];
        for (uint32_t j = threadIdx.x; j < tcap[sl]; j += L2F_T) out[to + j] = out[from + j];
    }
    for (unsigned long long i = threadIdx.x; i < ns; i += L2F_T) {
        const int sl = find(pl.spill_child[i], false);
        out[tbase[sl] + tcap[sl] + atomicAdd(&tfill[sl], 1u)] = ((const RT *)pl.spill_rec)[i];
    }
    __syncthreads();
    for (int sl = threadIdx.x; sl < L2F_SLOTS; sl += L2F_T) {
        if (tkey[sl] == NOKEY) continue;
        cstart[tkey[sl]] = tbase[sl];
        lend[tkey[sl]] = tbase[sl] + tcap[sl] + tcnt[sl];
    }
}

// ---- level 1 of the k = 33..63 path straight from the packed reads (uniform length): a thread owns
// 16 consecutive windows of one read and rolls the two-word k-mer and its reverse complement through
// them; the histogram kernel counts digits, the scatter kernel feeds write-combining rings (drained
// every W2_STEPS windows).  The 16-byte elements are never written unpartitioned.
struct WideSrc { const uint64_t *words; int64_t n_reads, nk, segs, total; int wpr, k, fc; };
constexpr int W2SEG = 16, W2_STEPS = 4, W2T = 1024, W2B = 16, W2A = 4;

struct W2State { uint64_t f0, f1, r0, r1, nxt; int v, j; };

__device__ __forceinline__ void w2_init(const WideSrc &s, int64_t g, W2State &st) {
    st.v = 0; st.j = 0;
    if (g >= s.total) return;
    const int64_t r = g / s.segs;
    const int p0 = (int)(g - r * s.segs) * W2SEG;
    int v = (int)(s.nk - p0);
    v = v > W2SEG ? W2SEG : v;
    const int res = s.k - 32;
    const uint64_t *w = s.words + r * s.wpr;
    const int b = s.fc + p0;
    st.f0 = kmer_at(w, b, 32); st.f1 = kmer_at(w, b + 32, res);
    st.r0 = revcomp(kmer_at(w, b + s.k - 32, 32), 32); st.r1 = revcomp(kmer_at(w, b, res), res);
    st.nxt = v > 1 ? kmer_at(w, b + s.k, v - 1) : 0;
    st.v = v;
}
// canonical element of the current window, then roll one base on
__device__ __forceinline__ Rec w2_step(W2State &st, int res, uint64_t mres) {
    const bool use_f = st.f0 != st.r0 ? st.f0 < st.r0 : st.f1 <= st.r1;          // ties -> forward
    const Rec e{use_f ? st.f0 : st.r0, use_f ? st.f1 : st.r1};
    if (st.j + 1 < st.v) {
        const uint64_t nb = (st.nxt >> (2 * (st.v - 2 - st.j))) & 3;
        const uint64_t cf = st.f1 >> (2 * (res - 1));
        st.f1 = ((st.f1 << 2) | nb) & mres;
        st.f0 = (st.f0 << 2) | cf;
        const uint64_t cr = st.r0 & 3;
        st.r0 = (st.r0 >> 2) | ((nb ^ 3) << 62);
        st.r1 = (st.r1 >> 2) | (cr << (2 * (res - 1)));
    }
    st.j++;
    return e;
}

__global__ __launch_bounds__(W2T) void k_w2_hist(WideSrc s, Level lv, uint64_t *__restrict__ blockhist) {
    __shared__ uint32_t h[1 << MAX_BITS];
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    for (int i = threadIdx.x; i < nb; i += W2T) h[i] = 0;
    __syncthreads();
    const int res = s.k - 32;
    const uint64_t mres = low_mask(res);
    for (int64_t g = (int64_t)blockIdx.x * W2T + threadIdx.x; g < s.total; g += (int64_t)gridDim.x * W2T) {
        W2State st;
        w2_init(s, g, st);
        while (st.j < st.v) atomicAdd(&h[wide_level1_digit(w2_step(st, res, mres), lv)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += W2T) blockhist[(int64_t)i * gridDim.x + blockIdx.x] = h[i];
}

__global__ __launch_bounds__(W2T) void k_w2_scatter(WideSrc s, Level lv, const uint64_t *__restrict__ scanned,
                                                    Rec *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char w2_smem[];
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
#define buf ((Rec *)w2_smem)
#define tail ((unsigned long long *)(w2_smem + (size_t)nb * W2B * sizeof(Rec)))
#define head (tail + nb)
    for (int i = threadIdx.x; i < nb; i += W2T) tail[i] = head[i] = scanned[(int64_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    const int res = s.k - 32;
    const uint64_t mres = low_mask(res);
    const int64_t stride = (int64_t)gridDim.x * W2T;
    for (int64_t gb = (int64_t)blockIdx.x * W2T; gb < s.total; gb += stride) {
        W2State st;
        w2_init(s, gb + threadIdx.x, st);
        for (int blk = 0; blk < W2SEG / W2_STEPS; blk++) {
#pragma unroll
            for (int q = 0; q < W2_STEPS; q++) {
                if (st.j < st.v) {
                    const Rec e = w2_step(st, res, mres);
                    const unsigned d = wide_level1_digit(e, lv);
                    const unsigned long long pos = atomicAdd(&tail[d], 1ULL);
                    if (pos - head[d] < (unsigned long long)W2B) buf[(size_t)d * W2B + (pos & (W2B - 1))] = e;
                    else out[pos] = e;
                }
            }
            __syncthreads();
            const bool final = gb + stride >= s.total && blk == W2SEG / W2_STEPS - 1;
            for (int d = threadIdx.x / W2B; d < nb; d += W2T / W2B) {
                const int j = threadIdx.x % W2B;
                const unsigned long long h = head[d], t = tail[d];
                unsigned long long e, nh;
                if (t - h > (unsigned long long)W2B) { e = h + W2B; nh = t; }
                else {
                    e = final ? t : (t & ~(unsigned long long)(W2A - 1));
                    if (e < h) e = h;
                    nh = e;
                }
                const unsigned long long g = h + j;
                if (g < e) out[g] = buf[(size_t)d * W2B + (g & (W2B - 1))];
                if (j == 0) head[d] = nh;
            }
            __syncthreads();
        }
    }
#undef buf
#undef tail
#undef head
}

// ------------------------------------------------------------ synthetic reads

constexpr uint64_t TAG_GENOME = 0x47454E4F4D45ULL, TAG_PAIRS = 0x5041495253ULL, TAG_ERRORS = 0x4552524F5253ULL;

__global__ void k_synth_genome(uint64_t sg, int64_t nw, uint64_t *__restrict__ g) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nw) g[j] = splitmix64(sg + (uint64_t)j);
}

__device__ __forceinline__ unsigned genome_base(const uint64_t *__restrict__ g, int64_t i) {
    return (unsigned)((g[i >> 5] >> (62 - 2 * (i & 31))) & 3);
}

// one thread per output word of a read
__global__ void k_synth_reads(uint64_t sp, uint64_t se, const uint64_t *__restrict__ genome,
                              int64_t genome_len, int64_t first_read, int64_t n_reads, int read_len,
                              uint32_t err, int wpr, uint64_t *__restrict__ words) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads * wpr) return;
    int64_t rl = t / wpr;
    int w = (int)(t % wpr);
    int64_t r = first_read + rl;
    uint64_t pair = (uint64_t)r >> 1;
    int mate = (int)(r & 1);
    uint64_t u = splitmix64(sp + pair);
    int64_t s = (int64_t)(u & 0xFFFF) + (int64_t)((u >> 16) & 0xFFFF) + (int64_t)((u >> 32) & 0xFFFF) +
                (int64_t)((u >> 48) & 0xFFFF);
    int64_t frag = 350 + ((s - 131070) * 35) / 37837;
    if (frag < read_len) frag = read_len;
    if (frag > genome_len) frag = genome_len;
    uint64_t v = splitmix64(u);
    int64_t start = (int64_t)((v >> 1) % (uint64_t)(genome_len - frag + 1));
    int strand = (int)(v & 1);
    int is_rc = mate ^ strand;
    int64_t pos = is_rc ? start + frag - read_len : start;
    uint64_t x = 0;
    for (int jj = 0; jj < 32; jj++) {
        int j = w * 32 + jj;
        unsigned b = 0;
        if (j < read_len) {
            b = is_rc ? 3u - genome_base(genome, pos + read_len - 1 - j) : genome_base(genome, pos + j);
            uint64_t e = splitmix64(se + (uint64_t)r * (uint64_t)read_len + (uint64_t)j);
            if ((uint32_t)e < err) b = (b + 1u + (unsigned)((e >> 32) % 3u)) & 3u;
        }
        x = (x << 2) | b;
    }
    words[t] = x;
}

size_t scatter_lds_bytes(int nb) { return (size_t)PTILE * 8 + (size_t)nb * 8 + (size_t)(nb + 1) * 4 + 16; }

}  // namespace

namespace rfx {

int64_t kmers_per_read(int read_len, int k, int front_clip, int end_clip) {
    return nk_of(read_len, k, front_clip, end_clip);
}

int encode_reads(rfx_ctx *ctx, const uint8_t *d_bases, const int64_t *d_read_off, int64_t n_reads,
                 int words_per_read, uint64_t *d_words, uint32_t *d_read_len) {
    int64_t total = n_reads * words_per_read;
    if (total <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_encode, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, d_bases,
                       d_read_off, n_reads, words_per_read, d_words, d_read_len);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int kmer_counts_per_read(rfx_ctx *ctx, const int64_t *d_read_off, int64_t n_reads, int k, int front_clip,
                         int end_clip, uint64_t *d_nk) {
    if (n_reads <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_nk_per_read, dim3((unsigned)ceil_div(n_reads, 256)), dim3(256), 0, ctx->stream,
                       d_read_off, n_reads, k, front_clip, end_clip, d_nk);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int extract_ordered_packed(rfx_ctx *ctx, const uint64_t *d_words, int wpr, const uint64_t *d_kmer_off,
                           int64_t n_reads, int k, int front_clip, uint64_t *d_out) {
    if (n_reads <= 0) return RFX_OK;
    int64_t threads = n_reads * 64;
    hipLaunchKernelGGL(k_extract_ordered, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, ctx->stream,
                       d_words, wpr, d_kmer_off, n_reads, k, front_clip, d_out);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int ragged_instances(rfx_ctx *ctx, const uint32_t *d_read_len, int64_t n_reads, int k, int front_clip,
                         int end_clip, int64_t *out_total) {
    *out_total = 0;
    if (n_reads <= 0) return RFX_OK;
    DevBuf nk, off;
    RFX_HIP(nk.alloc((size_t)n_reads * 8, ctx->stream));
    RFX_HIP(off.alloc((size_t)(n_reads + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_nk_from_len, dim3((unsigned)ceil_div(n_reads, 256)), dim3(256), 0, ctx->stream, d_read_len, n_reads,
                       k, front_clip, end_clip, nk.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u64(ctx, nk.as<uint64_t>(), off.as<uint64_t>(), n_reads));
    uint64_t t = 0;
    RFX_HIP(hipMemcpyAsync(&t, off.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    *out_total = (int64_t)t;
    return RFX_OK;
}

int64_t count_workspace_bytes(int64_t n_kmers) { return 2 * n_kmers * 8 + (int64_t)(64 << 20); }

static void plan_levels(int64_t n, bool from_reads, std::vector<int> &bits, double target = 16384.0) {
    bits.clear();
    if (const char *e = getenv("RFX_LEVEL_BITS")) {          // tuning override, e.g. "9,9"
        for (const char *q = e; *q;) {
            int v = atoi(q);
            if (v >= 0 && v <= MAX_BITS) bits.push_back(v);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        if (!bits.empty()) return;
    }
    if (const char *e = getenv("RFX_LEAF_TARGET")) target = atof(e) > 0 ? atof(e) : target;
    int B = 0;
    if ((double)n > target) B = (int)std::ceil(std::log2((double)n / target));
    if (B > 3 * MAX_BITS) B = 3 * MAX_BITS;
    int L = (B + MAX_BITS - 1) / MAX_BITS;
    for (int l = 0; l < L; l++) bits.push_back(B / L + (l >= L - B % L ? 1 : 0));   // later levels take the extra bit
    if (bits.empty() && from_reads) bits.push_back(0);   // materialise the instances once
}

static ReadSrc make_read_src(const ReadStore *reads) {
    ReadSrc s{};
    s.words = reads->words; s.n_reads = reads->n_reads; s.wpr = reads->words_per_read;
    s.fc = reads->front_clip; s.k = reads->k;
    s.nk = (int)kmers_per_read(reads->read_len, reads->k, reads->front_clip, reads->end_clip);
    s.segs = s.nk > 0 ? (s.nk + PK - 1) / PK : 1;
    s.n_threads = s.n_reads * s.segs;
    s.len_arr = reads->read_len_arr; s.ec = reads->end_clip;
    return s;
}

// k-mer instances of a read set
static int64_t instances_of(const ReadStore *reads, const ReadSrc &s) {
    return reads->n_instances >= 0 ? reads->n_instances : (int64_t)s.nk * reads->n_reads;
}


static int set_scatter_attrs(rfx_ctx *ctx) {
    static bool done = false;
    if (done) return RFX_OK;
    RFX_HIP(hipFuncSetAttribute((const void *)k_vb_scatter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)scatter_lds_bytes(1 << MAX_BITS)));
    RFX_HIP(hipFuncSetAttribute((const void *)k_reads_scatter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)scatter_lds_bytes(1 << MAX_BITS)));
    done = true;
    return RFX_OK;
}

// grid of the persistent reads kernels: a few workgroups per CU, never more than tiles
static unsigned reads_grid(rfx_ctx *ctx, const ReadSrc &s, int per_cu) {
    int64_t tiles = ceil_div(s.n_threads, PT);
    int64_t g = (int64_t)ctx->num_cu * per_cu;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, g));
}

// leaf count + filter + ascending sort of the survivors (shared tail of every source kind)
// pair_out: the survivors leave as 16-byte {k-mer, count} pairs in d_out_keys, in no particular order
// (the local combine of the multi-GPU count: every distinct k-mer, no filter, no sort)
template <int ELEM>
static int finish_leaves(rfx_ctx *ctx, const typename LeafElem<ELEM>::T *elems, int64_t elem_count,
                         const uint64_t *d_leaf_off, int64_t nleaf, int k, int min_cov, int max_cov, int twin, int key_bits,
                         uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n,
                         int64_t *out_distinct, bool pair_out = false, const uint64_t *d_leaf_end_in = nullptr) {
    constexpr bool RECS = ELEM == 1;
    // leaf l = elements [d_leaf_off[l], d_leaf_end[l]): back to back (the exact levels) or with slack between them (level 2 in
    // one sweep, l2_sweep below)
    const uint64_t *d_leaf_end = d_leaf_end_in ? d_leaf_end_in : d_leaf_off + 1;
    DevBuf co_buf;
    RFX_HIP(co_buf.alloc(sizeof(CountOut), ctx->stream));
    RFX_HIP(hipMemsetAsync(co_buf.p, 0, sizeof(CountOut), ctx->stream));
    const int apply = !pair_out && !(twin == RFX_TWIN_RDD && min_cov <= 1);    // P/ReflexivMain.java:160
    // heavy leaves (low-complexity sequence: millions of instances of a few k-mers in one bucket) are
    // left out of the first launch, cut into slices and counted by the whole grid in a second one
    // (record leaves hold ~2500 records; one of up to 16 x the pre-split threshold still goes through the
    // main launch in hash-selected parts)
    uint64_t heavy = RECS ? 65536 : 131072, slice = RECS ? 2048 : 16384, pcap_min = 1 << 20;
    if (const char *e = getenv("RFX_HEAVY")) {          // test knob: "heavy,slice,pcap" in elements
        unsigned long long a = 0, b = 0, c = 0;
        if (sscanf(e, "%llu,%llu,%llu", &a, &b, &c) == 3 && a && b && c) { heavy = a; slice = b; pcap_min = c; }
    }
    const int dbg = getenv("RFX_LEAF_DBG") ? atoi(getenv("RFX_LEAF_DBG")) : 0;
    // records beyond which a leaf starts in 2, 4, .. parts (a record holds ~0.5 distinct k-mers at high coverage,
    // a table takes ~3300 keys before probe sequences run long); measured neutral for pairs and not used there
    const uint32_t presplit = getenv("RFX_PRESPLIT") ? (uint32_t)atoi(getenv("RFX_PRESPLIT")) : ELEM == 1 ? 8000u : 0u;
    DevBuf nsl, spos;
    RFX_HIP(nsl.alloc((size_t)nleaf * 8, ctx->stream));
    RFX_HIP(spos.alloc((size_t)(nleaf + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_heavy_count, dim3((unsigned)ceil_div(nleaf, 256)), dim3(256), 0, ctx->stream, d_leaf_off, d_leaf_end, nleaf,
                       heavy, slice, nsl.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u64(ctx, nsl.as<uint64_t>(), spos.as<uint64_t>(), nleaf));
    uint64_t n_slices = 0;
    RFX_HIP(hipMemcpyAsync(&n_slices, spos.as<uint64_t>() + nleaf, 8, hipMemcpyDeviceToHost, ctx->stream));
    {
        ScopedTimer t(ctx, "leaf");
        // many more workgroups than fit (2 per CU are resident): a workgroup's contiguous chunk of leaves is
        // short, and the chunks even out what the leaves' sizes do not (leaf 14.6 -> 13.3 ms against 2 per CU)
        // (pair output: every workgroup leaves part of its last block(s) as holes, so fewer of them)
        const int leaf_per_cu = getenv("RFX_LEAF_PER_CU") ? std::max(1, atoi(getenv("RFX_LEAF_PER_CU"))) : pair_out ? 4 : 32;
        int64_t grid = std::min<int64_t>(nleaf, (int64_t)ctx->num_cu * leaf_per_cu);      // persistent, <= 78 KB LDS each
        static const bool kc_off = getenv("RFX_LEAF_KC") && atoi(getenv("RFX_LEAF_KC")) == 0;
        auto *kern = k_leaf_count<ELEM, 0>;
        if (RECS && k == 31 && !kc_off) kern = k_leaf_count<ELEM, RECS ? 31 : 0>;
        static const int extra_lds = getenv("RFX_LEAF_EXTRA_LDS") ? atoi(getenv("RFX_LEAF_EXTRA_LDS")) : 0;   // occupancy experiment
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(LT), (size_t)extra_lds, ctx->stream, elems, d_leaf_off, d_leaf_end, nleaf,
                           (const uint64_t *)nullptr, (const uint64_t *)nullptr, heavy, (uint64_t)elem_count, k, min_cov,
                           max_cov, apply, d_out_keys, d_out_counts, (unsigned long long)cap, co_buf.as<CountOut>(), dbg,
                           (int)pair_out, presplit);
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(sync_checked(ctx));
    if (n_slices > 0) {
        DevBuf sb, se, co2, pk, pc, tk, tv;
        RFX_HIP(sb.alloc((size_t)n_slices * 8, ctx->stream));
        RFX_HIP(se.alloc((size_t)n_slices * 8, ctx->stream));
        RFX_HIP(co2.alloc(sizeof(CountOut), ctx->stream));
        hipLaunchKernelGGL(k_heavy_fill, dim3((unsigned)ceil_div(nleaf, 256)), dim3(256), 0, ctx->stream, d_leaf_off, d_leaf_end, nleaf,
                           (const uint64_t *)spos.as<uint64_t>(), sb.as<uint64_t>(), se.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        // partial counts: every distinct key of every slice; grow on demand
        uint64_t pcap = std::max<uint64_t>(pcap_min, n_slices * 64);
        CountOut c2{};
        for (;;) {
            RFX_HIP(pk.alloc((size_t)pcap * 8, ctx->stream));
            RFX_HIP(pc.alloc((size_t)pcap * 4, ctx->stream));
            RFX_HIP(hipMemsetAsync(co2.p, 0, sizeof(CountOut), ctx->stream));
            {
                ScopedTimer t(ctx, "leaf");
                int64_t grid = std::min<int64_t>((int64_t)n_slices, (int64_t)ctx->num_cu * 2);
                hipLaunchKernelGGL(k_leaf_count<ELEM>, dim3((unsigned)grid), dim3(LT), 0, ctx->stream, elems, d_leaf_off, d_leaf_end,
                                   (int64_t)n_slices, (const uint64_t *)sb.as<uint64_t>(), (const uint64_t *)se.as<uint64_t>(),
                                   (uint64_t)0, (uint64_t)elem_count, k, min_cov, max_cov, 0, pk.as<uint64_t>(),
                                   pc.as<int32_t>(), (unsigned long long)pcap, co2.as<CountOut>(), dbg, 0, 0u);
                RFX_HIP(hipGetLastError());
            }
            RFX_HIP(hipMemcpyAsync(&c2, co2.p, sizeof c2, hipMemcpyDeviceToHost, ctx->stream));
            RFX_TRY(sync_checked(ctx));
            if (c2.n_failed) { ctx->last_error = "leaf split depth exhausted"; ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
            if (c2.n_out <= pcap) break;
            pcap = c2.n_out;
        }
        if (c2.n_out >= (1ULL << 32)) { ctx->last_error = "too many partial counts"; ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
        RFX_HIP(tk.alloc((size_t)c2.n_out * 8, ctx->stream));
        RFX_HIP(tv.alloc((size_t)c2.n_out * 4, ctx->stream));
        ScopedTimer t(ctx, "leaf");
        RFX_TRY(sort_pairs(ctx, pk.as<uint64_t>(), pc.as<uint32_t>(), (int64_t)c2.n_out, key_bits, tk.as<uint64_t>(),
                           tv.as<uint32_t>()));
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)ceil_div((int64_t)c2.n_out, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)pk.as<uint64_t>(), (const uint32_t *)pc.as<uint32_t>(), (int64_t)c2.n_out,
                           min_cov, max_cov, apply, d_out_keys, d_out_counts, (unsigned long long)cap, co_buf.as<CountOut>(),
                           (int)pair_out);
        RFX_HIP(hipGetLastError());
        t.stop();
        if (getenv("RFX_TRACE"))
            fprintf(stderr, "heavy leaves: %llu slices, %llu partial counts\n", (unsigned long long)n_slices, c2.n_out);
    }
    CountOut co{};
    RFX_HIP(hipMemcpyAsync(&co, co_buf.p, sizeof co, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (out_n) *out_n = (int64_t)co.n_out;
    if (out_distinct) *out_distinct = (int64_t)co.n_distinct;
    if (getenv("RFX_TRACE"))
        fprintf(stderr, "leaves: %lld buckets, %llu table passes, %llu overflowed\n", (long long)nleaf, co.n_passes, co.n_overflow);
    // table statistics of the last count call, read back by tests through rfx_last_count_timing (launches = the number)
    ctx->timing["stat_leaves"].launches += nleaf; ctx->timing["stat_passes"].launches += (int64_t)co.n_passes;
    ctx->timing["stat_overflows"].launches += (int64_t)co.n_overflow;
    if (dbg & 128)
        fprintf(stderr, "record table: %llu records counted in it, %llu expanded on the spot (%llu windows), %llu slots swept (%llu windows), %.2f per leaf\n",
                co.r_placed, co.r_direct, co.w_direct, co.r_slots, co.w_table, (double)co.r_slots / (double)std::max<int64_t>(nleaf, 1));
    if (dbg & 32)
        fprintf(stderr, "leaf waves: %.1f %% of their clocks at the leaf barriers\n", 100.0 * (double)co.t_wait / (double)std::max<unsigned long long>(co.t_all, 1));
    if (co.n_failed) { ctx->last_error = "leaf split depth exhausted"; ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
    if ((int64_t)co.n_out > cap) { ScopedTimer::collect(ctx); return RFX_E_CAP; }
    if ((int64_t)co.n_out > (int64_t)0xFFFFFFFFLL) { ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
    // ascending k-mer order (order contract B.0)
    if (!pair_out) {
        DevBuf tk, tv;
        RFX_HIP(tk.alloc((size_t)co.n_out * 8, ctx->stream));
        RFX_HIP(tv.alloc((size_t)co.n_out * 4, ctx->stream));
        ScopedTimer t(ctx, "sort");
        RFX_TRY(sort_pairs(ctx, d_out_keys, reinterpret_cast<uint32_t *>(d_out_counts), (int64_t)co.n_out, key_bits,
                           tk.as<uint64_t>(), tv.as<uint32_t>()));
        t.stop();
        RFX_TRY(sync_checked(ctx));
    }
    ScopedTimer::collect(ctx);
    return RFX_OK;
}

// super-k-mer records are used for k = 28..31 (W = k - 12 in 16..19); RFX_SUPERKMER=0 disables them
static bool superkmer_enabled(int k) {
    if (const char *e = getenv("RFX_SUPERKMER")) if (atoi(e) == 0) return false;
    return k >= SK_M + SK_MIN_W - 1 && k <= 31;
}

#define RFX_SK_W_CASES(X) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18)
template <bool DESC, class... Args>
static void launch_sk_hist(int W, dim3 grid, hipStream_t st, Args... args) {
    switch (W) {
#define X(w) case w: hipLaunchKernelGGL((k_sk_hist<w, DESC>), grid, dim3(SKT), 0, st, args...); break;
        RFX_SK_W_CASES(X)
#undef X
        default: hipLaunchKernelGGL((k_sk_hist<19, DESC>), grid, dim3(SKT), 0, st, args...); break;
    }
}

template <int W, bool DESC, bool WIDE, class... Args>
static hipError_t launch_sk_scatter_w(dim3 grid, size_t lds, hipStream_t st, Args... args) {
    hipError_t e = hipFuncSetAttribute((const void *)k_sk_scatter<W, DESC, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sk_scatter<W, DESC, WIDE>), grid, dim3(SKT), lds, st, args...);
    return hipGetLastError();
}
template <bool DESC, bool WIDE, class... Args>
static hipError_t launch_sk_scatter(int W, dim3 grid, size_t lds, hipStream_t st, Args... args) {
    if constexpr (WIDE) {                  // the central window is 30 or 31 bases: W = 18 or 19
        if (W == 18) return launch_sk_scatter_w<18, DESC, true>(grid, lds, st, args...);
        return launch_sk_scatter_w<19, DESC, true>(grid, lds, st, args...);
    } else {
        switch (W) {
#define X(w) case w: return launch_sk_scatter_w<w, DESC, false>(grid, lds, st, args...);
            RFX_SK_W_CASES(X)
#undef X
            default: return launch_sk_scatter_w<19, DESC, false>(grid, lds, st, args...);
        }
    }
}

// level 1 of the record path: reads -> records bucketed by `lv` (radix digit or owner).
// *out_recs (workspace slot `ws_slot`, or the caller's buffer d_dst of cap_dst records) receives
// the records, d_seg_off[nb+1] their bucket offsets; *n_recs the total.
template <bool WIDE = false>
static int records_from_reads(rfx_ctx *ctx, const ReadSrc &rsrc, const Level &lv, bool use_ws, int ws_slot,
                              std::conditional_t<WIDE, WRec, Rec> *d_dst, int64_t cap_dst, uint64_t *d_seg_off,
                              std::conditional_t<WIDE, WRec, Rec> **out_recs, int64_t *n_recs,
                              const char *hn, const char *pn) {
    using Rec = std::conditional_t<WIDE, WRec, ::Rec>;
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    const int W = rsrc.k - SK_M + 1;
    const size_t sk_lds = (size_t)nb * (SKB * sizeof(Rec) + 16);
    // two workgroups fit a CU when the rings are small; four times as many are launched then (the rest queue
    // behind the first and even out the tail: hist1 5.0 -> 4.6 ms)
    int per_cu = sk_lds <= 80 * 1024 ? 8 : 1;
    if (const char *e = getenv("RFX_SK_PER_CU")) per_cu = std::max(1, atoi(e));
    const unsigned G = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(rsrc.n_threads, SKT), (int64_t)ctx->num_cu * per_cu));
    DevBuf bh, scanned;
    RFX_HIP(bh.alloc((size_t)nb * G * 8, ctx->stream));
    RFX_HIP(scanned.alloc(((size_t)nb * G + 1) * 8, ctx->stream));
    // run descriptors (hist -> scatter) live in the workspace slot the records do not use yet
    uint32_t *desc = nullptr;
    const bool want_desc = lv.n_owners <= 64 && !(getenv("RFX_SK_DESC") && atoi(getenv("RFX_SK_DESC")) == 0);
    // (records into the caller's buffer: both slots are free; owner mode adds two words of owners)
    if (want_desc) desc = (uint32_t *)ctx->ws_get(use_ws && ws_slot == 1 ? 0 : 1, (size_t)(3 + SKD) * 4 * (size_t)rsrc.n_threads);
    {
        ScopedTimer t(ctx, hn);
        if (desc) launch_sk_hist<true>(W, dim3(G), ctx->stream, rsrc, lv, bh.as<uint64_t>(), desc);
        else launch_sk_hist<false>(W, dim3(G), ctx->stream, rsrc, lv, bh.as<uint64_t>(), (uint32_t *)nullptr);
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u64(ctx, bh.as<uint64_t>(), scanned.as<uint64_t>(), (int64_t)nb * G));
    hipLaunchKernelGGL(k_bin_offsets, dim3((unsigned)ceil_div(nb + 1, 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t *)scanned.as<uint64_t>(), nb, (int64_t)G, d_seg_off);
    RFX_HIP(hipGetLastError());
    uint64_t R = 0;
    RFX_HIP(hipMemcpyAsync(&R, scanned.as<uint64_t>() + (size_t)nb * G, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    *n_recs = (int64_t)R;
    Rec *dst = d_dst;
    if (use_ws) {
        dst = (Rec *)ctx->ws_get(ws_slot, (size_t)R * sizeof(Rec));
        if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
    } else if (!dst || (int64_t)R > cap_dst) {
        return RFX_E_CAP;                      // caller's buffer missing or too small: *n_recs holds the need
    }
    {
        ScopedTimer t(ctx, pn);
        if (desc) RFX_HIP((launch_sk_scatter<true, WIDE>(W, dim3(G), sk_lds, ctx->stream, rsrc, lv,
                                                         (const uint64_t *)scanned.as<uint64_t>(), dst, (const uint32_t *)desc)));
        else RFX_HIP((launch_sk_scatter<false, WIDE>(W, dim3(G), sk_lds, ctx->stream, rsrc, lv,
                                                     (const uint64_t *)scanned.as<uint64_t>(), dst, (const uint32_t *)nullptr)));
    }
    *out_recs = dst;
    return RFX_OK;
}

// level 1 of the record path in one sweep (k_sk_onesweep): -> records in workspace slot `ws_slot`, bucket b in
// [d_seg_begin[b], d_seg_end[b]).  *done = false: a region overflowed, nothing is valid, take the two-pass form.
template <bool WIDE = false>
static int records_onesweep(rfx_ctx *ctx, const ReadSrc &rsrc_in, const Level &lv, int ws_slot, uint64_t *d_seg_begin,
                            uint64_t *d_seg_end, std::conditional_t<WIDE, WRec, Rec> **out_recs, int64_t *n_recs, bool *done,
                            const char *hn, const char *pn, std::conditional_t<WIDE, WRec, Rec> *d_dst = nullptr, int64_t cap_dst = 0) {
    using Rec = std::conditional_t<WIDE, WRec, ::Rec>;
    *done = false;
    const int nb = 1 << lv.bits;
    const int W = rsrc_in.k - SK_M + 1;
    // 32 windows per thread (seg_runs; RFX_SK_SEG=16: the 16 of rounds 1-2).  The k = 33..63 records keep 16.
    // (its 16-slot rings fit the LDS up to 512 buckets)
    const bool seg32 = !WIDE && nb <= 512 && !(getenv("RFX_SK_SEG") && atoi(getenv("RFX_SK_SEG")) == 16);
    ReadSrc rsrc = rsrc_in;
    if (seg32) {
        rsrc.segs = rsrc.nk > 0 ? (rsrc.nk + 31) / 32 : 1;
        rsrc.n_threads = rsrc.n_reads * rsrc.segs;
    }
    const int OST = seg32 ? OsGeo<32, false>::T : SKT;            // threads per workgroup of the sweep = segments per tile
    const int64_t ntile = ceil_div(rsrc.n_threads, OST);
    const int64_t ntile_s = ceil_div(rsrc.n_threads, SKT);        // (the sampled histogram keeps tiles of SKT segments)
    // two workgroups per CU (one when a workgroup fills the CU); fewer on small inputs (every workgroup holds three extents
    // per bucket: at least 32 tiles each)
    const int per_cu = WIDE || (seg32 && OST == SKT) ? 1 : 2;
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(OS_MAXG, (int64_t)ctx->num_cu * per_cu),
                                                                std::max<int64_t>(std::min<int64_t>(ntile, 64), ntile / 32)));
    int sample = getenv("RFX_SK_SAMPLE") ? std::max(1, atoi(getenv("RFX_SK_SAMPLE"))) : 32;
    if (ntile_s < 64 * (int64_t)sample) sample = (int)std::max<int64_t>(1, ntile_s / 64);      // small inputs: at least 64 tiles
    const int64_t n_sampled = ceil_div(ntile_s, sample);
    const int cap_pct = getenv("RFX_SK_ONESWEEP_CAP") ? std::max(1, atoi(getenv("RFX_SK_ONESWEEP_CAP"))) : 100;
    DevBuf hist, reg_start, reg_cap, cursor, totals, holes;
    RFX_HIP(hist.alloc((size_t)nb * 8 + 16, ctx->stream));                 // + the overflow flag
    RFX_HIP(reg_start.alloc((size_t)(nb + 1) * 8, ctx->stream));
    RFX_HIP(reg_cap.alloc((size_t)nb * 4, ctx->stream));
    RFX_HIP(cursor.alloc((size_t)nb * OS_CSTRIDE * 8, ctx->stream));
    RFX_HIP(totals.alloc(24, ctx->stream));
    RFX_HIP(holes.alloc((size_t)nb * OS_HOLES * G * 8, ctx->stream));
    RFX_HIP(hipMemsetAsync(hist.p, 0, (size_t)nb * 8 + 16, ctx->stream));
    int *d_overflow = (int *)(hist.as<unsigned long long>() + nb);
    {
        ScopedTimer t(ctx, hn);
        const dim3 gs((unsigned)std::min<int64_t>(n_sampled, (int64_t)ctx->num_cu * 8));
        switch (W) {
#define X(w) case w: if (seg32) hipLaunchKernelGGL((k_sk_sample_hist<w, 32>), gs, dim3(SKT), 0, ctx->stream, rsrc, lv, sample, hist.as<unsigned long long>()); \
                     else hipLaunchKernelGGL((k_sk_sample_hist<w>), gs, dim3(SKT), 0, ctx->stream, rsrc, lv, sample, hist.as<unsigned long long>()); break;
            RFX_SK_W_CASES(X)
            default: X(19)
#undef X
        }
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_plan_regions, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long *)hist.as<unsigned long long>(), nb,
                           (double)ntile_s / (double)n_sampled, G, cap_pct, (double)ntile, reg_start.as<uint64_t>(), reg_cap.as<uint32_t>(),
                           cursor.as<unsigned long long>(), totals.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
    }
    unsigned long long h_tot[3] = {0, 0, 0};
    RFX_HIP(hipMemcpyAsync(h_tot, totals.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    const uint32_t ose = (uint32_t)h_tot[2];
    if (ose == 0) {                       // one bucket takes more of a tile than the largest extent: two passes straight away
        if (getenv("RFX_TRACE")) fprintf(stderr, "one-sweep level 1: not tried (the sample puts too much on one bucket)\n");
        return RFX_OK;
    }
    int ose_shift = 0;
    while ((1u << ose_shift) < ose) ose_shift++;
    Rec *dst = d_dst;
    if (d_dst) {                          // the caller's buffer (the regions with their slack must fit): *n_recs = the need
        if ((int64_t)h_tot[0] > cap_dst) { *n_recs = (int64_t)h_tot[0]; return RFX_E_CAP; }
    } else {
        dst = (Rec *)ctx->ws_get(ws_slot, (size_t)h_tot[0] * sizeof(Rec));
        if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
    }
    const OneSweep os{reg_start.as<uint64_t>(), reg_cap.as<uint32_t>(), cursor.as<unsigned long long>(), holes.as<uint64_t>(), d_overflow,
                      (uint64_t)h_tot[0], hist.as<unsigned long long>() + nb + 1, ose, (uint32_t)ose_shift};
    const size_t lds = (size_t)nb * ((seg32 ? OsGeo<32, false>::B : SKB) * sizeof(Rec) + 24);
    {
        ScopedTimer t(ctx, pn);
        if constexpr (WIDE) {                  // the central window is 30 or 31 bases: W = 18 or 19
            if (W == 18) {
                RFX_HIP(hipFuncSetAttribute((const void *)k_sk_onesweep<18, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_sk_onesweep<18, true>), dim3((unsigned)G), dim3(SKT), lds, ctx->stream, rsrc, lv, os, dst);
            } else {
                RFX_HIP(hipFuncSetAttribute((const void *)k_sk_onesweep<19, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_sk_onesweep<19, true>), dim3((unsigned)G), dim3(SKT), lds, ctx->stream, rsrc, lv, os, dst);
            }
        } else {
            switch (W) {
#define X(w) case w: if (seg32) { \
                         RFX_HIP(hipFuncSetAttribute((const void *)k_sk_onesweep<w, false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                         hipLaunchKernelGGL((k_sk_onesweep<w, false, 32>), dim3((unsigned)G), dim3(OST), lds, ctx->stream, rsrc, lv, os, dst); \
                     } else { \
                         RFX_HIP(hipFuncSetAttribute((const void *)k_sk_onesweep<w>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                         hipLaunchKernelGGL((k_sk_onesweep<w>), dim3((unsigned)G), dim3(SKT), lds, ctx->stream, rsrc, lv, os, dst); \
                     } break;
                RFX_SK_W_CASES(X)
                default: X(19)
#undef X
            }
        }
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_fix_holes<Rec>, dim3((unsigned)nb), dim3(FH_T), 0, ctx->stream, os, G, dst, d_seg_begin, d_seg_end,
                           totals.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
    }
    int h_over = 0;
    RFX_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(&h_over, d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (getenv("RFX_TRACE"))
        fprintf(stderr, "one-sweep level 1: %llu records in regions of %llu (sample 1/%d, %d workgroups, extents of %u)%s\n", h_tot[1], h_tot[0], sample, G, ose,
                h_over ? " -- a region overflowed: two passes instead" : "");
    if (h_over == 2) { ctx->last_error = "one-sweep level 1: holes and tail records do not balance"; return RFX_E_STATE; }
    if (h_over) return RFX_OK;
    *out_recs = dst;
    *n_recs = (int64_t)h_tot[1];
    *done = true;
    return RFX_OK;
}

// A first level over ONE unpartitioned record array without a histogram pass (k_rec_claim_scatter): -> records in workspace slot
// `oslot`, bucket b in [d_seg_begin[b], d_seg_end[b]).  *done = false: not tried (small input, too many bins) or a bucket outgrew
// its region -- nothing is valid and the exact form runs.  On for the super-k-mer records (k <= 31 and k = 33..63) from 2^24
// records up (RFX_REC_ONESWEEP=0: never, 1: every element kind, 2: at any size -- the tests).
// Measured in the multi-GPU rehearsal (6.25 Gbp as a rank of 8, 4 generations, 185 M records each at k = 31): the level itself is as
// fast as the exact scatter or faster and the histogram pass is gone -- k = 31: hist1 2.75 -> 0.32, 40.3 -> 37.5 ms a step; k = 63:
// hist1 4.2 -> 0.35, part1 23.5 -> 20.7, 87.9 -> 81.8.  The buckets then fill in the order the workgroups' claims land, and the NEXT
// level's 1/16 sample misjudges a handful of its 131,072 children by more than its six standard deviations in every generation (a
// child is a few minimiser sites, their records arrive in clumps) -- which voided that level's sweep (+6 ms) until its children
// could spill (L2Plan, k_l2_spill_fix).
template <int MODE>
static int records_resweep(rfx_ctx *ctx, const typename LevelElem<MODE>::T *recs, int64_t n_recs, const Level &lv, int used, int oslot,
                           uint64_t *d_seg_begin, uint64_t *d_seg_end, const typename LevelElem<MODE>::T **out_recs, bool *done,
                           const char *hn, const char *pn) {
    using RT = typename LevelElem<MODE>::T;
    *done = false;
    const int mode = getenv("RFX_REC_ONESWEEP") ? atoi(getenv("RFX_REC_ONESWEEP")) : (MODE == 0 || MODE == 3 ? 1 : 0);
    if (MODE == 2 || !mode || lv.bits < 4 || lv.bits > 9 || lv.n_owners > 0 || (mode != 2 && n_recs < ((int64_t)1 << 24))) return RFX_OK;
    const int nb = 1 << lv.bits;
    constexpr int B = MODE == 3 ? 8 : 16;
    if (nb > 256 && B * sizeof(RT) * nb + 28 * (size_t)nb > 160 * 1024) return RFX_OK;
    const size_t lds = (size_t)nb * (B * sizeof(RT) + 28);
    const int64_t nchunk = ceil_div(n_recs, PT);
    int sample = 32;
    if (nchunk < 256 * (int64_t)sample) sample = (int)std::max<int64_t>(1, nchunk / 256);      // small inputs: at least 256 chunks
    const int64_t n_sampled = ceil_div(nchunk, sample);
    DevBuf hist, reg_start, cursor;
    RFX_HIP(hist.alloc((size_t)nb * 8 + 8, ctx->stream));                  // + the overflow flag
    RFX_HIP(reg_start.alloc((size_t)(nb + 1) * 8, ctx->stream));
    RFX_HIP(cursor.alloc((size_t)nb * 8, ctx->stream));
    RFX_HIP(hipMemsetAsync(hist.p, 0, (size_t)nb * 8 + 8, ctx->stream));
    const ClaimPlan pl{reg_start.as<uint64_t>(), cursor.as<unsigned long long>(), (int *)(hist.as<unsigned long long>() + nb)};
    {
        ScopedTimer t(ctx, hn);
        hipLaunchKernelGGL(k_rec_sample_hist<MODE>, dim3((unsigned)std::min<int64_t>(n_sampled, (int64_t)ctx->num_cu * 8)), dim3(PT), 0, ctx->stream,
                           recs, (uint64_t)n_recs, lv, used, sample, hist.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_claim_plan, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long *)hist.as<unsigned long long>(), nb,
                           (double)nchunk / (double)n_sampled, pl);
        RFX_HIP(hipGetLastError());
    }
    uint64_t total = 0;
    RFX_HIP(hipMemcpyAsync(&total, reg_start.as<uint64_t>() + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if ((int64_t)total > 2 * n_recs + ((int64_t)1 << 26)) return RFX_OK;
    RT *dst = (RT *)ctx->ws_get(oslot, (size_t)total * sizeof(RT));
    if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
    {
        ScopedTimer t(ctx, pn);
        constexpr int64_t CHUNK = (int64_t)RCL_TILES * WCT * WC_PER;
        RFX_HIP(hipFuncSetAttribute((const void *)k_rec_claim_scatter<B, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // (as many workgroups as are resident, all of them moving through the array together: the buckets fill in the array's order.
        //  With four times as many, started in four shifts, a minimiser site's records lay in four short stretches of its bucket
        //  and the next level's 1/16 sample misjudged a few children in every generation)
        const int per_cu = lds <= 72 * 1024 ? 2 : 1;
        hipLaunchKernelGGL((k_rec_claim_scatter<B, MODE>), dim3((unsigned)std::min<int64_t>(ceil_div(n_recs, CHUNK), (int64_t)ctx->num_cu * per_cu)), dim3(WCT), lds, ctx->stream, recs, (uint64_t)n_recs, lv, used,
                           pl, dst);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_claim_ends, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, ctx->stream, pl, nb, d_seg_begin, d_seg_end);
        RFX_HIP(hipGetLastError());
    }
    int h_over = 0;
    RFX_HIP(hipMemcpyAsync(&h_over, pl.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (getenv("RFX_TRACE"))
        fprintf(stderr, "record level without a histogram: %lld records in regions of %llu (sample 1/%d)%s\n", (long long)n_recs, (unsigned long long)total,
                sample, h_over ? " -- a bucket outgrew its region: the exact form instead" : "");
    if (h_over) return RFX_OK;
    *out_recs = dst;
    *done = true;
    return RFX_OK;
}

// levels [first_level, ...) of the record path on records already bucketed by `used` bits
// (seg offsets in *seg_cur), then the leaves.
// the partition levels of a record array: -> the fully partitioned array and its leaf offsets
template <int MODE>
static int partition_record_levels(rfx_ctx *ctx, const typename LevelElem<MODE>::T *recs, int64_t n_recs, int ws_slot_of_recs,
                                   const std::vector<int> &bits, size_t first_level, int used, DevBuf **seg_cur_io,
                                   DevBuf **seg_next_io, int64_t *nseg_io, const typename LevelElem<MODE>::T **cur_out,
                                   const uint64_t *seg_end_first = nullptr, DevBuf *seg_end_buf = nullptr,
                                   const uint64_t **seg_end_out = nullptr) {
    using Rec = typename LevelElem<MODE>::T;
    DevBuf *seg_cur = *seg_cur_io, *seg_next = *seg_next_io;
    int64_t nseg = *nseg_io;
    const Rec *cur = recs;
    int slot = ws_slot_of_recs;          // the next level writes into the other slot
    const uint64_t *seg_end_cur = seg_end_first;     // ends of the current segments when they are not the next one's begin
    if (seg_end_out) *seg_end_out = nullptr;
    for (size_t l = first_level; l < bits.size(); l++) {
        Level lv{};
        lv.bits = bits[l];
        const int nb = 1 << lv.bits;
        const int64_t nchild = nseg << lv.bits;
        // one unpartitioned array (what a rank received): the level in one sweep, regions from a sample, when the caller can take
        // segment ends (records_resweep)
        if (nseg == 1 && seg_end_buf && seg_end_out && !seg_end_cur) {
            bool swept = false;
            const Rec *dst_s = nullptr;
            RFX_HIP(seg_next->alloc((size_t)(nchild + 1) * 8, ctx->stream));
            RFX_HIP(seg_end_buf->alloc((size_t)nchild * 8, ctx->stream));
            RFX_TRY(records_resweep<MODE>(ctx, cur, n_recs, lv, used, slot == 0 ? 1 : 0, seg_next->as<uint64_t>(), seg_end_buf->as<uint64_t>(), &dst_s,
                                          &swept, l == 0 ? "hist1" : l == 1 ? "hist2" : "hist3", l == 0 ? "part1" : l == 1 ? "part2" : "part3"));
            if (swept) {
                slot = slot == 0 ? 1 : 0;
                cur = dst_s;
                used += lv.bits;
                std::swap(seg_cur, seg_next);
                nseg = nchild;
                seg_end_cur = seg_end_buf->as<uint64_t>();
                continue;
            }
        }
        const int64_t total_tiles = ceil_div(std::max<int64_t>(n_recs, 1), PTILE);
        int tpb = (int)std::min<int64_t>(32, std::max<int64_t>(1, total_tiles / ((int64_t)ctx->num_cu * 8)));
        if (const char *e = getenv("RFX_TPB")) tpb = std::max(1, atoi(e));
        const int64_t v_bound = ceil_div(std::max<int64_t>(n_recs, 1), (int64_t)tpb * PTILE) + nseg;
        DevBuf nvb, vb_start, table, scanned;
        RFX_HIP(nvb.alloc((size_t)nseg * 8, ctx->stream));
        RFX_HIP(vb_start.alloc((size_t)(nseg + 1) * 8, ctx->stream));
        RFX_HIP(table.alloc((size_t)nb * v_bound * 4, ctx->stream));
        RFX_HIP(scanned.alloc(((size_t)nb * v_bound + 1) * 8, ctx->stream));
        RFX_HIP(seg_next->alloc((size_t)(nchild + 1) * 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(table.p, 0, (size_t)nb * v_bound * 4, ctx->stream));
        const uint64_t *seg_end = seg_end_cur;
        hipLaunchKernelGGL(k_vb_per_seg, dim3((unsigned)ceil_div(nseg, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)seg_cur->as<uint64_t>(), seg_end, nseg, tpb, nvb.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u64(ctx, nvb.as<uint64_t>(), vb_start.as<uint64_t>(), nseg));
        VbMap vm{seg_cur->as<uint64_t>(), vb_start.as<uint64_t>(), nseg, tpb, seg_end};
        const char *hn = l == 0 ? "hist1" : l == 1 ? "hist2" : "hist3";
        const char *pn = l == 0 ? "part1" : l == 1 ? "part2" : "part3";
        {
            ScopedTimer t(ctx, hn);
            hipLaunchKernelGGL(k_rec_hist<MODE>, dim3((unsigned)v_bound), dim3(PT), 0, ctx->stream, cur, vm, lv, used,
                               table.as<uint32_t>());
            RFX_HIP(hipGetLastError());
        }
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), scanned.as<uint64_t>(), (int64_t)nb * v_bound));
        hipLaunchKernelGGL(k_child_offsets, dim3((unsigned)ceil_div(nchild + 1, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)scanned.as<uint64_t>(), vm, nb, (uint64_t)n_recs, seg_next->as<uint64_t>());
        RFX_HIP(hipGetLastError());
        slot = slot == 0 ? 1 : 0;
        Rec *dst = (Rec *)ctx->ws_get(slot, (size_t)std::max<int64_t>(n_recs, 1) * sizeof(Rec));
        if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
        {
            ScopedTimer t(ctx, pn);
            const bool wc = !(getenv("RFX_WC") && atoi(getenv("RFX_WC")) == 0) && lv.bits >= 4;
            if (wc && MODE == 3) {                   // 32-byte records: 8 slots per bin fill the LDS at 512 bins
                if (lv.bits > 9) {                   // 1024 bins: 4 slots each (64-byte lines)
                    const size_t lds = (size_t)nb * (4 * sizeof(Rec) + 16);
                    RFX_HIP(hipFuncSetAttribute((const void *)k_rec_scatter_wc<4, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((k_rec_scatter_wc<4, MODE>), dim3((unsigned)v_bound), dim3(WCT), lds, ctx->stream, cur, vm, lv,
                                       used, (const uint64_t *)scanned.as<uint64_t>(), dst);
                } else {
                    const size_t lds = (size_t)nb * (8 * sizeof(Rec) + 16);
                    RFX_HIP(hipFuncSetAttribute((const void *)k_rec_scatter_wc<8, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((k_rec_scatter_wc<8, MODE>), dim3((unsigned)v_bound), dim3(WCT), lds, ctx->stream, cur, vm, lv,
                                       used, (const uint64_t *)scanned.as<uint64_t>(), dst);
                }
            } else if (wc && lv.bits <= 9 && !(getenv("RFX_WC_B") && atoi(getenv("RFX_WC_B")) == 8)) {
                const size_t lds = (size_t)nb * (16 * sizeof(Rec) + 16);
                RFX_HIP(hipFuncSetAttribute((const void *)k_rec_scatter_wc<16, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_rec_scatter_wc<16, MODE>), dim3((unsigned)v_bound), dim3(WCT), lds, ctx->stream, cur, vm, lv,
                                   used, (const uint64_t *)scanned.as<uint64_t>(), dst);
            } else if (wc) {
                const size_t lds = (size_t)nb * (8 * sizeof(Rec) + 16);
                RFX_HIP(hipFuncSetAttribute((const void *)k_rec_scatter_wc<8, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_rec_scatter_wc<8, MODE>), dim3((unsigned)v_bound), dim3(WCT), lds, ctx->stream, cur, vm, lv,
                                   used, (const uint64_t *)scanned.as<uint64_t>(), dst);
            } else {
                hipLaunchKernelGGL(k_rec_scatter<MODE>, dim3((unsigned)v_bound), dim3(PT), 0, ctx->stream, cur, vm, lv, used,
                                   (const uint64_t *)scanned.as<uint64_t>(), dst);
            }
            RFX_HIP(hipGetLastError());
        }
        cur = dst;
        used += lv.bits;
        std::swap(seg_cur, seg_next);
        nseg = nchild;
        seg_end_cur = nullptr;                       // (an exact level leaves gap-free segments)
    }
    *seg_cur_io = seg_cur; *seg_next_io = seg_next; *nseg_io = nseg; *cur_out = cur;
    if (seg_end_out) *seg_end_out = seg_end_cur;
    return RFX_OK;
}

// The last level of a record path in one sweep (k_rec_l2sweep above), when it can be: a real level (4..9 bits: the rings), at
// most 1024 parents of about the same size, enough records that the regions' floor is small.  *ok: the leaves are in *dst_out,
// leaf c = [cstart[c], lend[c]) of `total` record slots; else nothing happened that the exact form cannot undo.
// RFX_L2_ONESWEEP=0: never.
template <int MODE>
static int last_level_sweep(rfx_ctx *ctx, const typename LevelElem<MODE>::T *cur_h, int64_t n_recs, int slot, int bits_last, int used_h,
                            const uint64_t *sb, const uint64_t *seg_end, int64_t nseg_h, size_t level_index,
                            const typename LevelElem<MODE>::T **dst_out, DevBuf &cstart, DevBuf &lend, int64_t *nchild_out, uint64_t *total_out,
                            bool *ok) {
    using RT = typename LevelElem<MODE>::T;
    *ok = false;
    const bool l2s_off = getenv("RFX_L2_ONESWEEP") && atoi(getenv("RFX_L2_ONESWEEP")) == 0;
    const int nb = 1 << bits_last;
    if (l2s_off || bits_last < 4 || bits_last > 10 || nseg_h > 1024 || n_recs < (int64_t)nseg_h * nb * 256) return RFX_OK;
    const int64_t nchild = nseg_h << bits_last;
    // what the regions may take at most is fixed here, without a readback (the sampled sizes give ~1.7 x the records; a plan
    // beyond it raises flags[0] and the exact form runs)
    // (+ room behind the regions for the children that spill: k_l2_spill_fix)
    const uint32_t spill_cap = (uint32_t)std::min<int64_t>((int64_t)1 << 22, std::max<int64_t>(65536, n_recs / 256));
    const uint64_t bound = (uint64_t)n_recs * 2 + (uint64_t)nchild * 640 + 4096 + 8ULL * spill_cap + 65536;
    const int oslot = slot == 0 ? 1 : 0;
    RT *dst = (RT *)ctx->ws_get(oslot, (size_t)bound * sizeof(RT));
    if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
    DevBuf sampled, ccap, flags, spill_rec, spill_child;
    RFX_HIP(spill_rec.alloc((size_t)spill_cap * sizeof(RT), ctx->stream));
    RFX_HIP(spill_child.alloc((size_t)spill_cap * 4, ctx->stream));
    RFX_HIP(sampled.alloc((size_t)nchild * 4, ctx->stream)); RFX_HIP(ccap.alloc((size_t)nchild * 4, ctx->stream));
    RFX_HIP(cstart.alloc((size_t)(nchild + 1) * 8, ctx->stream));
    RFX_HIP(flags.alloc(32, ctx->stream));                     // two flags, the spill cursor, the cursor of the room behind the regions
    RFX_HIP(lend.alloc((size_t)nchild * 8, ctx->stream));
    RFX_HIP(hipMemsetAsync(sampled.p, 0, (size_t)nchild * 4, ctx->stream));
    L2Plan pl{sb, seg_end ? seg_end : sb + 1, (const uint64_t *)cstart.as<uint64_t>(), flags.as<int>(), spill_rec.p, spill_child.as<uint32_t>(),
              flags.as<unsigned long long>() + 1, spill_cap, flags.as<unsigned long long>() + 2, bound};
    Level lv{};
    lv.bits = bits_last;
    {
        ScopedTimer t(ctx, level_index == 0 ? "hist1" : level_index == 1 ? "hist2" : "hist3");
        hipLaunchKernelGGL(k_l2_sample<MODE>, dim3((unsigned)nseg_h, 8), dim3(PT), 0, ctx->stream, cur_h, pl, lv, used_h, sampled.as<uint32_t>());
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l2_caps, dim3((unsigned)ceil_div(nchild, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)sampled.as<uint32_t>(), nchild,
                           ccap.as<uint32_t>(), getenv("RFX_L2_SQUEEZE") ? std::max(1, atoi(getenv("RFX_L2_SQUEEZE"))) : 100);
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, ccap.as<uint32_t>(), cstart.as<uint64_t>(), nchild));
    {
        ScopedTimer t(ctx, level_index == 0 ? "part1" : level_index == 1 ? "part2" : "part3");
        hipLaunchKernelGGL(k_l2_check, dim3(1), dim3(1024), 0, ctx->stream, pl, (int)nseg_h, nchild, bound);
        RFX_HIP(hipGetLastError());
        // ring slots per child: what fills the LDS (16-byte elements: 16 at 512 children, 8 at 1024; the 32-byte records 8 and 4)
        constexpr int B9 = MODE == 3 ? 8 : 16, B10 = B9 / 2;
        if (bits_last <= 9) {
            const size_t lds = (size_t)nb * (B9 * sizeof(RT) + 24);
            RFX_HIP(hipFuncSetAttribute((const void *)k_rec_l2sweep<B9, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_rec_l2sweep<B9, MODE>), dim3((unsigned)nseg_h), dim3(WCT), lds, ctx->stream, cur_h, pl, lv, used_h, dst, lend.as<uint64_t>());
        } else {
            const size_t lds = (size_t)nb * (B10 * sizeof(RT) + 24);
            RFX_HIP(hipFuncSetAttribute((const void *)k_rec_l2sweep<B10, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_rec_l2sweep<B10, MODE>), dim3((unsigned)nseg_h), dim3(WCT), lds, ctx->stream, cur_h, pl, lv, used_h, dst, lend.as<uint64_t>());
        }
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l2_spill_fix<RT>, dim3(1), dim3(L2F_T), 0, ctx->stream, pl, cstart.as<uint64_t>(), lend.as<uint64_t>(), dst);
        RFX_HIP(hipGetLastError());
    }
    unsigned long long h_fl[3] = {0, 0, 0};                    // {flags[0] | flags[1] << 32, spilled records, end of the room in use}
    uint64_t h_tot = 0;
    RFX_HIP(hipMemcpyAsync(h_fl, flags.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(&h_tot, cstart.as<uint64_t>() + nchild, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    const int h_flags[2] = {(int)(h_fl[0] & 0xffffffffULL), (int)(h_fl[0] >> 32)};
    if (!h_flags[0] && !h_flags[1]) ctx->timing["stat_l2_spilled"].launches += (int64_t)h_fl[1];     // records moved by k_l2_spill_fix
    if (!h_flags[0] && h_flags[1]) ctx->timing["stat_l2_void"].launches += 1;                        // sweeps given up for the exact form
    if (getenv("RFX_TRACE")) {
        fprintf(stderr, "last level in one sweep: %lld records in regions of %llu%s\n", (long long)n_recs, (unsigned long long)h_tot,
                h_flags[0] ? " -- not tried (skew, or no room): the exact form instead" : h_flags[1] ? " -- more spilled than the list takes: the exact form instead" : "");
        if (!h_flags[0] && !h_flags[1] && h_fl[1])
            fprintf(stderr, "  %llu records of children that outgrew their regions spilled and were moved with them behind the regions\n", h_fl[1]);
    }
    if (h_flags[1] && getenv("RFX_TRACE")) {
        std::vector<uint64_t> hs((size_t)nchild + 1), he((size_t)nchild);
        std::vector<uint32_t> hsm((size_t)nchild);
        RFX_HIP(hipMemcpy(hs.data(), cstart.p, hs.size() * 8, hipMemcpyDeviceToHost));
        RFX_HIP(hipMemcpy(he.data(), lend.p, he.size() * 8, hipMemcpyDeviceToHost));
        RFX_HIP(hipMemcpy(hsm.data(), sampled.p, hsm.size() * 4, hipMemcpyDeviceToHost));
        int64_t nover = 0; double worst = 0; int64_t wi = 0;
        for (int64_t c = 0; c < nchild; c++) {
            const uint64_t capc = hs[(size_t)c + 1] - hs[(size_t)c], need = he[(size_t)c] - hs[(size_t)c];
            if (need > capc) { nover++; const double r = (double)need / (double)capc; if (r > worst) { worst = r; wi = c; } }
        }
        {   // how the sample misjudged: children by need / (STRIDE x sampled)
            int64_t z = 0, r2 = 0, r4 = 0, big = 0;
            for (int64_t c = 0; c < nchild; c++) {
                const uint64_t need = he[(size_t)c] - hs[(size_t)c];
                if (need > hs[(size_t)c + 1] - hs[(size_t)c]) {
                    if (hsm[(size_t)c] == 0) z++;
                    else { const double r = (double)need / (16.0 * hsm[(size_t)c]); if (r > 4) r4++; else if (r > 2) r2++; }
                    if (need > 4000) big++;
                }
            }
            fprintf(stderr, "  of the children that ran over: %lld with nothing sampled, %lld with need > 4 x estimate, %lld with 2-4 x; %lld hold > 4000 records\n",
                    (long long)z, (long long)r4, (long long)r2, (long long)big);
            std::vector<uint64_t> hb((size_t)nseg_h), hen((size_t)nseg_h);
            RFX_HIP(hipMemcpy(hb.data(), sb, hb.size() * 8, hipMemcpyDeviceToHost));
            RFX_HIP(hipMemcpy(hen.data(), seg_end ? seg_end : sb + 1, hen.size() * 8, hipMemcpyDeviceToHost));
            uint64_t mn = ~0ULL, mx = 0;
            for (int64_t q = 0; q < nseg_h; q++) { const uint64_t n = hen[(size_t)q] - hb[(size_t)q]; mn = std::min(mn, n); mx = std::max(mx, n); }
            fprintf(stderr, "  parents: %lld of %llu .. %llu records\n", (long long)nseg_h, (unsigned long long)mn, (unsigned long long)mx);
        }
        fprintf(stderr, "  %lld of %lld children ran over; worst: child %lld sampled %u, region %llu, holds %llu\n", (long long)nover, (long long)nchild,
                (long long)wi, hsm[(size_t)wi], (unsigned long long)(hs[(size_t)wi + 1] - hs[(size_t)wi]), (unsigned long long)(he[(size_t)wi] - hs[(size_t)wi]));
    }
    if (h_flags[0] || h_flags[1]) return RFX_OK;
    *dst_out = dst; *nchild_out = nchild; *total_out = bound; *ok = true;     // (the leaves may lie anywhere in the buffer)
    return RFX_OK;
}

// every level but the last in the exact form, then the last in one sweep if it can be, in the exact form if not: -> the leaves
template <int MODE>
static int partition_to_leaves(rfx_ctx *ctx, const typename LevelElem<MODE>::T *recs, int64_t n_recs, int ws_slot_of_recs,
                               const std::vector<int> &bits, size_t first_level, int used, DevBuf *seg_cur, DevBuf *seg_next, int64_t nseg,
                               const uint64_t *seg_end_first, const typename LevelElem<MODE>::T **cur_out, int64_t *elem_count,
                               const uint64_t **leaf_off, const uint64_t **leaf_end, int64_t *nleaf, DevBuf &cstart, DevBuf &lend) {
    using RT = typename LevelElem<MODE>::T;
    *leaf_end = nullptr;
    if (bits.size() <= first_level) {                          // (no level left: the segments are the leaves)
        *cur_out = recs; *elem_count = n_recs; *leaf_off = (const uint64_t *)seg_cur->as<uint64_t>(); *nleaf = nseg;
        *leaf_end = seg_end_first;
        return RFX_OK;
    }
    const size_t last = bits.size() - 1;
    std::vector<int> head(bits.begin(), bits.begin() + (long)last);
    int slot = ws_slot_of_recs, used_h = used;
    const RT *cur_h = recs;
    int64_t nseg_h = nseg;
    DevBuf *sc = seg_cur, *sn = seg_next;
    const uint64_t *seg_end = seg_end_first;
    DevBuf seg_end_head;                                       // (ends of the head levels' segments when the last of them was a sweep)
    if (last > first_level) {
        RFX_TRY(partition_record_levels<MODE>(ctx, recs, n_recs, ws_slot_of_recs, head, first_level, used, &sc, &sn, &nseg_h, &cur_h, seg_end_first,
                                              &seg_end_head, &seg_end));
        for (size_t l = first_level; l < last; l++) { used_h += bits[l]; slot = slot == 0 ? 1 : 0; }
    }
    bool ok = false;
    const RT *dst = nullptr;
    int64_t nchild = 0;
    uint64_t total = 0;
    RFX_TRY(last_level_sweep<MODE>(ctx, cur_h, n_recs, slot, bits[last], used_h, (const uint64_t *)sc->as<uint64_t>(), seg_end, nseg_h, last, &dst, cstart,
                                   lend, &nchild, &total, &ok));
    if (ok) {
        *cur_out = dst; *elem_count = (int64_t)total; *leaf_off = (const uint64_t *)cstart.as<uint64_t>();
        *leaf_end = (const uint64_t *)lend.as<uint64_t>(); *nleaf = nchild;
        return RFX_OK;
    }
    const RT *cur = nullptr;
    RFX_TRY(partition_record_levels<MODE>(ctx, cur_h, n_recs, slot, bits, last, used_h, &sc, &sn, &nseg_h, &cur, seg_end));
    *cur_out = cur; *elem_count = n_recs; *leaf_off = (const uint64_t *)sc->as<uint64_t>(); *nleaf = nseg_h;
    return RFX_OK;
}

static int count_records_levels(rfx_ctx *ctx, const Rec *recs, int64_t n_recs, int ws_slot_of_recs,
                                const std::vector<int> &bits, size_t first_level, int used, DevBuf *seg_cur,
                                DevBuf *seg_next, int64_t nseg, int k, int min_cov, int max_cov, int twin,
                                uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n,
                                int64_t *out_distinct, bool pair_out = false, const uint64_t *seg_end_first = nullptr) {
    const Rec *cur = nullptr;
    const uint64_t *loff = nullptr, *lend = nullptr;
    int64_t ec = 0, nleaf = 0;
    DevBuf cstart, le;
    RFX_TRY(partition_to_leaves<0>(ctx, recs, n_recs, ws_slot_of_recs, bits, first_level, used, seg_cur, seg_next, nseg, seg_end_first, &cur, &ec,
                                   &loff, &lend, &nleaf, cstart, le));
    return finish_leaves<1>(ctx, cur, ec, loff, nleaf, k, min_cov, max_cov, twin, 2 * k, d_out_keys, d_out_counts, cap, out_n, out_distinct,
                            pair_out, lend);
}

// reads -> records -> count (the default for k = 28..31)
static int count_reads_superkmer(rfx_ctx *ctx, const ReadStore *reads, int min_cov, int max_cov, int twin,
                                 uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n,
                                 int64_t *out_distinct, bool pair_out) {
    ctx->timing.clear();
    ReadSrc rsrc = make_read_src(reads);
    const int64_t n = instances_of(reads, rsrc);
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n <= 0) return RFX_OK;
    std::vector<int> bits;
    plan_levels(n, true, bits, 16384.0);           // measured best for the record leaf with double hashing + pre-split (tools/bits_sweep.sh)
    // (20 bits and more put 1024 bins on level 1 and one workgroup per CU there, the rings fill the LDS.  Measured on one
    // GPU: 20 Gbp of the 4.64 Mbp genome 138.8 ms with 10 + 10 bits, 131.7 with 9 + 10 and leaves twice as large, 139
    // with three levels; 18.75 Gbp of a 400 Mbp genome -- 2.5e9 distinct k-mers, a human-scale share -- 238 ms with
    // 10 + 10, 254 with 9 + 10: the leaves of a deep data set want to be small, so the plan stays as it is)
    Level lv{};
    lv.bits = bits[0];
    StageArena stage_arena(ctx, ((size_t)64 << 20) + (size_t)n / 8);
    DevBuf segA, segB;
    RFX_HIP(segA.alloc(((size_t)(1 << lv.bits) + 1) * 8, ctx->stream));
    Rec *recs = nullptr;
    int64_t R = 0;
    // level 1 in one sweep when the input is large enough for a sampled histogram to size the regions (and another
    // level follows: the leaves want gap-free buckets); RFX_SK_ONESWEEP=0 two passes always, =2 one sweep at any size
    const int os_mode = getenv("RFX_SK_ONESWEEP") ? atoi(getenv("RFX_SK_ONESWEEP")) : 1;
    bool swept = false;
    DevBuf segE;
    if (os_mode && bits.size() >= 2 && (os_mode == 2 || rsrc.n_threads >= ((int64_t)1 << 22))) {
        RFX_HIP(segE.alloc((size_t)(1 << lv.bits) * 8, ctx->stream));
        RFX_TRY(records_onesweep(ctx, rsrc, lv, 0, segA.as<uint64_t>(), segE.as<uint64_t>(), &recs, &R, &swept, "hist1", "part1"));
    }
    if (!swept) RFX_TRY(records_from_reads(ctx, rsrc, lv, true, 0, nullptr, 0, segA.as<uint64_t>(), &recs, &R, "hist1", "part1"));
    return count_records_levels(ctx, recs, R, 0, bits, 1, lv.bits, &segA, &segB, (int64_t)1 << lv.bits, reads->k,
                                min_cov, max_cov, twin, d_out_keys, d_out_counts, cap, out_n, out_distinct, pair_out,
                                swept ? (const uint64_t *)segE.as<uint64_t>() : nullptr);
}

int count_filter(rfx_ctx *ctx, const ReadStore *reads, const uint64_t *d_kmers, int64_t n,
                 int min_cov, int max_cov, int twin, void *ws, int64_t ws_bytes,
                 uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                 int64_t *out_n, int64_t *out_distinct, bool pair_out) {
    (void)ws; (void)ws_bytes;
    if (reads && superkmer_enabled(reads->k))
        return count_reads_superkmer(ctx, reads, min_cov, max_cov, twin, d_out_keys, d_out_counts, cap, out_n,
                                     out_distinct, pair_out);
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    ctx->timing.clear();
    ReadSrc rsrc{};
    int k_bits = 64;
    const bool from_reads = reads != nullptr;
    if (from_reads) {
        rsrc = make_read_src(reads);
        n = instances_of(reads, rsrc);
        k_bits = 2 * reads->k;
    }
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n <= 0) return RFX_OK;

    std::vector<int> bits;
    plan_levels(n, from_reads, bits);
    RFX_TRY(set_scatter_attrs(ctx));

    DevBuf bufA, bufB, segA, segB, co_buf;
    uint64_t seg_init[2] = {0, (uint64_t)n};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));     // seg_init lives on the stack
    int64_t nseg = 1;
    DevBuf *seg_cur = &segA, *seg_next = &segB;
    DevBuf *out_buf = &bufA, *in_buf = &bufB;
    const uint64_t *cur_arr = d_kmers;
    bool cur_from_reads = from_reads;
    int used_bits = 0;
    for (size_t l = 0; l < bits.size(); l++) {
        Level lv{};
        lv.bits = bits[l];
        lv.n_owners = 0;
        lv.parent_shift = 64 - used_bits;
        used_bits += bits[l];
        lv.shift = 64 - used_bits;
        if (lv.shift >= 64) lv.shift = 63;            // bits == 0 on the first level: digit mask is 0
        const int64_t nchild = nseg << lv.bits;
        RFX_HIP(seg_next->alloc((size_t)(nchild + 1) * 8, ctx->stream));
        if (!out_buf->p) {
            // the two big instance buffers live in the context (grow-only), not in the pool
            void *wsp = ctx->ws_get(out_buf == &bufA ? 0 : 1, (size_t)n * 8);
            if (!wsp) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
            out_buf->p = wsp; out_buf->borrowed = true;
        }
        const char *hn = l == 0 ? "hist1" : l == 1 ? "hist2" : "hist3";
        const char *pn = l == 0 ? "part1" : l == 1 ? "part2" : "part3";
        if (cur_from_reads) {
            // level 1 straight from the packed reads: per-workgroup histogram rows, one scan,
            // then a scatter with private cursors (same grid, same tile assignment, no atomics)
            const int nb = 1 << lv.bits;
            const unsigned G = reads_grid(ctx, rsrc, nb <= 512 ? 2 : 1);
            DevBuf bh, scanned;
            RFX_HIP(bh.alloc((size_t)nb * G * 8, ctx->stream));
            RFX_HIP(scanned.alloc(((size_t)nb * G + 1) * 8, ctx->stream));
            {
                ScopedTimer t(ctx, hn);
                hipLaunchKernelGGL(k_reads_hist, dim3(G), dim3(PT), 0, ctx->stream, rsrc, lv, bh.as<uint64_t>());
                RFX_HIP(hipGetLastError());
            }
            RFX_TRY(exclusive_scan_u64(ctx, bh.as<uint64_t>(), scanned.as<uint64_t>(), (int64_t)nb * G));
            hipLaunchKernelGGL(k_bin_offsets, dim3((unsigned)ceil_div(nb + 1, 256)), dim3(256), 0, ctx->stream,
                               (const uint64_t *)scanned.as<uint64_t>(), nb, (int64_t)G, seg_next->as<uint64_t>());
            RFX_HIP(hipGetLastError());
            {
                ScopedTimer t(ctx, pn);
                hipLaunchKernelGGL(k_reads_scatter, dim3(G), dim3(PT), scatter_lds_bytes(nb), ctx->stream, rsrc,
                                   lv, (const uint64_t *)scanned.as<uint64_t>(), out_buf->as<uint64_t>());
                RFX_HIP(hipGetLastError());
            }
        } else {
            // virtual workgroups of tpb tiles inside every parent; table rows -> scan -> private cursors
            const int nb = 1 << lv.bits;
            const int64_t total_tiles = ceil_div(n, PTILE);
            int tpb = (int)std::min<int64_t>(32, std::max<int64_t>(1, total_tiles / ((int64_t)ctx->num_cu * 8)));
            if (const char *e = getenv("RFX_TPB")) tpb = std::max(1, atoi(e));
            const int64_t v_bound = ceil_div(n, (int64_t)tpb * PTILE) + nseg;
            DevBuf nvb, vb_start, table, scanned;
            RFX_HIP(nvb.alloc((size_t)nseg * 8, ctx->stream));
            RFX_HIP(vb_start.alloc((size_t)(nseg + 1) * 8, ctx->stream));
            RFX_HIP(table.alloc((size_t)nb * v_bound * 4, ctx->stream));
            RFX_HIP(scanned.alloc(((size_t)nb * v_bound + 1) * 8, ctx->stream));
            RFX_HIP(hipMemsetAsync(table.p, 0, (size_t)nb * v_bound * 4, ctx->stream));
            hipLaunchKernelGGL(k_vb_per_seg, dim3((unsigned)ceil_div(nseg, 256)), dim3(256), 0, ctx->stream,
                               (const uint64_t *)seg_cur->as<uint64_t>(), (const uint64_t *)nullptr, nseg, tpb, nvb.as<uint64_t>());
            RFX_HIP(hipGetLastError());
            RFX_TRY(exclusive_scan_u64(ctx, nvb.as<uint64_t>(), vb_start.as<uint64_t>(), nseg));
            VbMap vm{seg_cur->as<uint64_t>(), vb_start.as<uint64_t>(), nseg, tpb};
            {
                ScopedTimer t(ctx, hn);
                hipLaunchKernelGGL(k_vb_hist, dim3((unsigned)v_bound), dim3(PT), 0, ctx->stream, cur_arr, vm, lv,
                                   table.as<uint32_t>());
                RFX_HIP(hipGetLastError());
            }
            RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), scanned.as<uint64_t>(), (int64_t)nb * v_bound));
            hipLaunchKernelGGL(k_child_offsets, dim3((unsigned)ceil_div(nchild + 1, 256)), dim3(256), 0, ctx->stream,
                               (const uint64_t *)scanned.as<uint64_t>(), vm, nb, (uint64_t)n, seg_next->as<uint64_t>());
            RFX_HIP(hipGetLastError());
            {
                ScopedTimer t(ctx, pn);
                hipLaunchKernelGGL(k_vb_scatter, dim3((unsigned)v_bound), dim3(PT), scatter_lds_bytes(nb), ctx->stream,
                                   cur_arr, vm, lv, (const uint64_t *)scanned.as<uint64_t>(), out_buf->as<uint64_t>());
                RFX_HIP(hipGetLastError());
            }
        }
        cur_arr = out_buf->as<uint64_t>();
        cur_from_reads = false;
        std::swap(out_buf, in_buf);
        std::swap(seg_cur, seg_next);
        nseg = nchild;
    }

    return finish_leaves<0>(ctx, cur_arr, n, (const uint64_t *)seg_cur->as<uint64_t>(), nseg, from_reads ? reads->k : 31,
                            min_cov, max_cov, twin, from_reads ? k_bits : 64, d_out_keys, d_out_counts, cap, out_n,
                            out_distinct, pair_out);
}

int bucket_by_owner(rfx_ctx *ctx, const ReadStore *reads, int n_owners, uint64_t *d_out, int64_t cap,
                    int64_t *d_owner_off, int64_t *h_owner_off) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (n_owners < 1 || n_owners > 64) return RFX_E_ARG;
    ReadSrc rsrc = make_read_src(reads);
    const int64_t n = instances_of(reads, rsrc);
    if (n > cap) return RFX_E_CAP;
    RFX_TRY(set_scatter_attrs(ctx));
    Level lv{};
    lv.bits = 6; lv.shift = 0; lv.parent_shift = 64; lv.n_owners = n_owners;
    const unsigned G = reads_grid(ctx, rsrc, 2);
    DevBuf bh, scanned;
    RFX_HIP(bh.alloc((size_t)n_owners * G * 8, ctx->stream));
    RFX_HIP(scanned.alloc(((size_t)n_owners * G + 1) * 8, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_reads_hist, dim3(G), dim3(PT), 0, ctx->stream, rsrc, lv, bh.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u64(ctx, bh.as<uint64_t>(), scanned.as<uint64_t>(), (int64_t)n_owners * G));
        hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)scanned.as<uint64_t>(),
                           n_owners, (int64_t)G, reinterpret_cast<uint64_t *>(d_owner_off));
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_reads_scatter, dim3(G), dim3(PT), scatter_lds_bytes(n_owners), ctx->stream, rsrc, lv,
                           (const uint64_t *)scanned.as<uint64_t>(), d_out);
        RFX_HIP(hipGetLastError());
    } else {
        RFX_HIP(hipMemsetAsync(d_owner_off, 0, (size_t)(n_owners + 1) * 8, ctx->stream));
    }
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    return RFX_OK;
}

// multi-GPU, record form: bucket this rank's reads by the OWNER of each run's minimiser
// (owner = mulhi(hash(minimiser), n_owners)) -- the exchange then ships ~2.6 B per instance
int bucket_records_by_owner(rfx_ctx *ctx, const ReadStore *reads, int n_owners, void *d_out, int64_t cap_records,
                            int64_t *d_owner_off, int64_t *h_owner_off, int64_t *out_n_records) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (n_owners < 1 || n_owners > 64 || !superkmer_enabled(reads->k)) return RFX_E_ARG;
    ReadSrc rsrc = make_read_src(reads);
    if (out_n_records) *out_n_records = 0;
    if (rsrc.nk <= 0 || reads->n_reads <= 0) {
        RFX_HIP(hipMemsetAsync(d_owner_off, 0, (size_t)(n_owners + 1) * 8, ctx->stream));
        if (h_owner_off) memset(h_owner_off, 0, (size_t)(n_owners + 1) * 8);
        return RFX_OK;
    }
    Level lv{};
    lv.n_owners = n_owners;
    Rec *recs = nullptr;
    int64_t R = 0;
    int st = records_from_reads(ctx, rsrc, lv, false, 0, (Rec *)d_out, cap_records,
                                reinterpret_cast<uint64_t *>(d_owner_off), &recs, &R, "hist1", "part1");
    if (out_n_records) *out_n_records = R;
    if (st != RFX_OK) return st;
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    return RFX_OK;
}

// One owner's bucket lies in S consecutive bins of the sweep, bin i gap-free in [seg_begin, seg_end) with the regions' slack
// between the bins: the records that lie beyond the bucket's own length are moved into the gaps before it, so that the
// bucket is ONE run [begin, begin + total) -- one message per peer, as with the two-pass form.  Sources lie at or
// beyond the cut, destinations before it.  grid (owners, chunks).
template <class RT>
__global__ __launch_bounds__(1024) void k_close_gaps(RT *__restrict__ recs, const uint64_t *__restrict__ seg_begin,
                                                     const uint64_t *__restrict__ seg_end, int S, uint64_t *__restrict__ out_begin,
                                                     uint64_t *__restrict__ out_end, int *__restrict__ fault) {
    __shared__ uint64_t slo[512], dlo[512], spre[513], dpre[513];       // (S <= 512: one owner and all the sweep's bins)
    const int o = blockIdx.x;
    const uint64_t *sb = seg_begin + (size_t)o * S, *se = seg_end + (size_t)o * S;
    if (threadIdx.x == 0) {
        uint64_t R = 0;
        for (int i = 0; i < S; i++) R += se[i] - sb[i];
        const uint64_t cut = sb[0] + R;
        spre[0] = dpre[0] = 0;
        for (int i = 0; i < S; i++) {
            const uint64_t lo = sb[i] > cut ? sb[i] : cut;
            slo[i] = lo;
            spre[i + 1] = spre[i] + (se[i] > lo ? se[i] - lo : 0);
            uint64_t g0 = se[i], g1 = i + 1 < S ? sb[i + 1] : se[i];
            if (g1 > cut) g1 = cut;
            dlo[i] = g0;
            dpre[i + 1] = dpre[i] + (g1 > g0 ? g1 - g0 : 0);
        }
        if (spre[S] != dpre[S]) *fault = 1;                      // (never: both count the bucket's records beyond the cut)
        if (blockIdx.y == 0) { out_begin[o] = sb[0]; out_end[o] = cut; }
    }
    __syncthreads();
    const uint64_t M = spre[S] < dpre[S] ? spre[S] : dpre[S];
    for (uint64_t k = (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; k < M; k += (uint64_t)gridDim.y * blockDim.x) {
        int i = 0, j = 0;                                               // the last range that starts at or before k
        for (int st = S >> 1; st > 0; st >>= 1) {
            if (spre[i + st] <= k) i += st;
            if (dpre[j + st] <= k) j += st;
        }
        recs[dlo[j] + (k - dpre[j])] = recs[slo[i] + (k - spre[i])];
    }
}

// The same by level 1's ONE SWEEP (round 3).  The sweep wants a bin to receive a handful of records per round (8-slot
// rings, extents of 64), and an owner bucket of a rank of 8 receives hundreds -- so every owner's bucket is cut into
// 2^sub_bits bins by the top bits of the record header (512 bins in all; the receiver does not care which bin a record
// came through), and k_close_gaps then makes every owner's bins one run: owner b = records [h_begin[b], h_end[b]) of
// d_out, with slack between the owners.  *done = false: not tried or void (small input, skew, a region that
// overflowed): the caller takes bucket_records_by_owner.  RFX_E_CAP: *out_n_records = the records d_out must hold
// (regions + slack).
template <bool WIDE>
static int owner_sweep(rfx_ctx *ctx, const ReadSrc &rsrc, int n_owners, void *d_out, int64_t cap_records, int64_t *h_begin,
                       int64_t *h_end, int64_t *out_n_records, bool *done) {
    using RT = std::conditional_t<WIDE, WRec, Rec>;
    *done = false;
    const int os_mode = getenv("RFX_SK_ONESWEEP") ? atoi(getenv("RFX_SK_ONESWEEP")) : 1;
    if (!os_mode || rsrc.nk <= 0 || rsrc.n_reads <= 0 || !(os_mode == 2 || rsrc.n_threads >= ((int64_t)1 << 22))) return RFX_OK;
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    int obits = 0;
    while ((1 << obits) < n_owners) obits++;
    Level lv{};
    lv.n_owners = n_owners;
    lv.sub_bits = 9 - obits;                              // >= 3
    lv.bits = 9;
    const int nb = 1 << lv.bits, S = 1 << lv.sub_bits;
    DevBuf segB, segE, ob, oe, flt;
    RFX_HIP(segB.alloc((size_t)(nb + 1) * 8, ctx->stream));
    RFX_HIP(segE.alloc((size_t)nb * 8, ctx->stream));
    RFX_HIP(ob.alloc((size_t)n_owners * 8, ctx->stream));
    RFX_HIP(oe.alloc((size_t)n_owners * 8, ctx->stream));
    RFX_HIP(flt.alloc(4, ctx->stream));
    RFX_HIP(hipMemsetAsync(flt.p, 0, 4, ctx->stream));
    RT *recs = nullptr;
    int64_t R = 0;
    bool swept = false;
    int st = records_onesweep<WIDE>(ctx, rsrc, lv, 0, segB.as<uint64_t>(), segE.as<uint64_t>(), &recs, &R, &swept, "hist1", "part1",
                                    (RT *)d_out, cap_records);
    if (out_n_records) *out_n_records = R;
    if (st != RFX_OK || !swept) return st;
    {
        ScopedTimer t(ctx, "part1");
        hipLaunchKernelGGL(k_close_gaps<RT>, dim3((unsigned)n_owners, 16), dim3(1024), 0, ctx->stream, recs, (const uint64_t *)segB.as<uint64_t>(),
                           (const uint64_t *)segE.as<uint64_t>(), S, ob.as<uint64_t>(), oe.as<uint64_t>(), flt.as<int>());
        RFX_HIP(hipGetLastError());
    }
    int h_fault = 0;
    static_assert(sizeof(int64_t) == sizeof(uint64_t), "offsets");
    RFX_HIP(hipMemcpyAsync(h_begin, ob.p, (size_t)n_owners * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(h_end, oe.p, (size_t)n_owners * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(&h_fault, flt.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (h_fault) { ctx->last_error = "owner sweep: gaps and tail records do not balance"; return RFX_E_STATE; }
    *done = true;
    return RFX_OK;
}

int bucket_records_by_owner_sweep(rfx_ctx *ctx, const ReadStore *reads, int n_owners, void *d_out, int64_t cap_records,
                                  int64_t *h_begin, int64_t *h_end, int64_t *out_n_records, bool *done) {
    *done = false;
    if (n_owners < 1 || n_owners > 64 || !superkmer_enabled(reads->k)) return RFX_E_ARG;
    return owner_sweep<false>(ctx, make_read_src(reads), n_owners, d_out, cap_records, h_begin, h_end, out_n_records, done);
}

// count + filter of records that arrived from the exchange (any order): all radix levels run
// on the record headers, then the record leaves
int count_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k,
                  int min_cov, int max_cov, int twin, uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                  int64_t *out_n, int64_t *out_distinct) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (!superkmer_enabled(k)) return RFX_E_ARG;
    ctx->timing.clear();
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n_records <= 0) return RFX_OK;
    const int64_t n_inst = n_instances_hint > 0 ? n_instances_hint : n_records * 6;
    std::vector<int> bits;
    plan_levels(n_inst, false, bits, 16384.0);
    DevBuf segA, segB;
    uint64_t seg_init[2] = {0, (uint64_t)n_records};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    // slot -1: the caller's buffer; the first level writes workspace slot 0
    return count_records_levels(ctx, (const Rec *)d_records, n_records, 1, bits, 0, 0, &segA, &segB, 1, k, min_cov,
                                max_cov, twin, d_out_keys, d_out_counts, cap, out_n, out_distinct);
}

// ---- multi-GPU, combine form (the map-side combine of reduceByKey, P/ReflexivMain.java:155): every rank
// counts its own reads first (count_filter with pair_out: all distinct k-mers as 16-byte {k-mer, count}
// pairs), the PAIRS are bucketed by owner = mulhi(kmer_hash(k-mer), n_owners) and cross the exchange,
// and the owner sums what arrives.  At high coverage this ships ~1.4 B per instance instead of ~2.6.
int bucket_pairs_by_owner(rfx_ctx *ctx, const void *d_pairs, int64_t n, int n_owners, void *d_out,
                          int64_t *d_owner_off, int64_t *h_owner_off) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (n_owners < 1 || n_owners > 64 || n < 0) return RFX_E_ARG;
    if (n == 0) {
        RFX_HIP(hipMemsetAsync(d_owner_off, 0, (size_t)(n_owners + 1) * 8, ctx->stream));
        if (h_owner_off) memset(h_owner_off, 0, (size_t)(n_owners + 1) * 8);
        return RFX_OK;
    }
    Level lv{};
    lv.n_owners = n_owners;
    const int nb = n_owners;
    DevBuf seg, nvb, vb_start, table, scanned;
    uint64_t seg_init[2] = {0, (uint64_t)n};
    RFX_HIP(seg.alloc(16, ctx->stream));
    RFX_HIP(hipMemcpyAsync(seg.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    const int64_t total_tiles = ceil_div(n, PTILE);
    const int tpb = (int)std::min<int64_t>(32, std::max<int64_t>(1, total_tiles / ((int64_t)ctx->num_cu * 8)));
    const int64_t v_bound = ceil_div(n, (int64_t)tpb * PTILE) + 1;
    RFX_HIP(nvb.alloc(8, ctx->stream));
    RFX_HIP(vb_start.alloc(16, ctx->stream));
    RFX_HIP(table.alloc((size_t)nb * v_bound * 4, ctx->stream));
    RFX_HIP(scanned.alloc(((size_t)nb * v_bound + 1) * 8, ctx->stream));
    RFX_HIP(hipMemsetAsync(table.p, 0, (size_t)nb * v_bound * 4, ctx->stream));
    hipLaunchKernelGGL(k_vb_per_seg, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)seg.as<uint64_t>(), (const uint64_t *)nullptr, (int64_t)1, tpb,
                       nvb.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u64(ctx, nvb.as<uint64_t>(), vb_start.as<uint64_t>(), 1));
    VbMap vm{seg.as<uint64_t>(), vb_start.as<uint64_t>(), 1, tpb};
    const Rec *src = (const Rec *)d_pairs;
    {
        ScopedTimer t(ctx, "pair_hist");
        hipLaunchKernelGGL(k_rec_hist<2>, dim3((unsigned)v_bound), dim3(PT), 0, ctx->stream, src, vm, lv, 0, table.as<uint32_t>());
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), scanned.as<uint64_t>(), (int64_t)nb * v_bound));
    hipLaunchKernelGGL(k_child_offsets, dim3((unsigned)ceil_div(nb + 1, 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t *)scanned.as<uint64_t>(), vm, nb, (uint64_t)n, reinterpret_cast<uint64_t *>(d_owner_off));
    RFX_HIP(hipGetLastError());
    // holes (count 0) are not counted: the end of the last bucket is the histogram's total, not n
    RFX_HIP(hipMemcpyAsync(d_owner_off + n_owners, scanned.as<uint64_t>() + (size_t)nb * v_bound, 8, hipMemcpyDeviceToDevice,
                           ctx->stream));
    {
        ScopedTimer t(ctx, "pair_part");
        if (nb >= 16) {
            const size_t lds = (size_t)nb * (16 * sizeof(Rec) + 16);
            RFX_HIP(hipFuncSetAttribute((const void *)k_rec_scatter_wc<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_rec_scatter_wc<16, 2>), dim3((unsigned)v_bound), dim3(WCT), lds, ctx->stream, src, vm, lv, 0,
                               (const uint64_t *)scanned.as<uint64_t>(), (Rec *)d_out);
        } else {
            hipLaunchKernelGGL(k_rec_scatter<2>, dim3((unsigned)v_bound), dim3(PT), 0, ctx->stream, src, vm, lv, 0,
                               (const uint64_t *)scanned.as<uint64_t>(), (Rec *)d_out);
        }
        RFX_HIP(hipGetLastError());
    }
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        ScopedTimer::collect(ctx);
    }
    return RFX_OK;
}

// the owner's half: sum the partial counts of every k-mer that arrived, filter, ascending order
int merge_pairs(rfx_ctx *ctx, const void *d_pairs, int64_t n, int k, int min_cov, int max_cov, int twin,
                uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    ctx->timing.clear();
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n <= 0) return RFX_OK;
    std::vector<int> bits;
    plan_levels(n, false, bits, 2048.0);           // pairs are mostly distinct keys: ~half-full tables
    DevBuf segA, segB;
    uint64_t seg_init[2] = {0, (uint64_t)n};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    DevBuf *seg_cur = &segA, *seg_next = &segB;
    int64_t nseg = 1;
    const Rec *cur = nullptr;
    RFX_TRY(partition_record_levels<2>(ctx, (const Rec *)d_pairs, n, 1, bits, 0, 0, &seg_cur, &seg_next, &nseg, &cur));
    return finish_leaves<2>(ctx, cur, n, (const uint64_t *)seg_cur->as<uint64_t>(), nseg, k, min_cov, max_cov, twin, 2 * k,
                            d_out_keys, d_out_counts, cap, out_n, out_distinct);
}

template <bool RECS = false>
static int finish_wide2(rfx_ctx *ctx, const std::conditional_t<RECS, WRec, Rec> *cur, const uint64_t *d_leaf_off, int64_t nseg,
                        int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                        int64_t *out_distinct, int k = 63, const uint64_t *d_leaf_end = nullptr) {
    if (!d_leaf_end) d_leaf_end = d_leaf_off + 1;
    DevBuf co_buf;
    RFX_HIP(co_buf.alloc(sizeof(CountOut), ctx->stream));
    RFX_HIP(hipMemsetAsync(co_buf.p, 0, sizeof(CountOut), ctx->stream));
    {
        ScopedTimer t(ctx, "leaf");
        const int wleaf_per_cu = getenv("RFX_WLEAF_PER_CU") ? std::max(1, atoi(getenv("RFX_WLEAF_PER_CU"))) : 1;
        const int64_t grid = std::min<int64_t>(nseg, (int64_t)ctx->num_cu * wleaf_per_cu);
        hipLaunchKernelGGL(k_leaf_count_wide<RECS>, dim3((unsigned)grid), dim3(WLT), 0, ctx->stream, cur, d_leaf_off, d_leaf_end, nseg, k,
                           min_cov, max_cov, d_out_keys, d_out_counts, (unsigned long long)cap, co_buf.as<CountOut>(),
                           (uint32_t)(getenv("RFX_WIDE_PRESPLIT") ? atoi(getenv("RFX_WIDE_PRESPLIT")) : 2600) |
                               (getenv("RFX_WIDE_DBG") ? (uint32_t)atoi(getenv("RFX_WIDE_DBG")) << 30 : 0u) |
                               (getenv("RFX_WIDE_LINEAR") ? 0x20000000u : 0u) | (getenv("RFX_WIDE_NOAGG") ? 0x10000000u : 0u) |
                               (getenv("RFX_WIDE_STATS") ? 0x08000000u : 0u));
        RFX_HIP(hipGetLastError());
    }
    CountOut co{};
    RFX_HIP(hipMemcpyAsync(&co, co_buf.p, sizeof co, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (getenv("RFX_TRACE"))
        fprintf(stderr, "wide leaves: %lld buckets, %llu table passes, %llu overflowed\n", (long long)nseg, co.n_passes, co.n_overflow);
    if (getenv("RFX_WIDE_STATS"))
        fprintf(stderr, "wide record table: %llu records counted in it, %llu expanded on the spot, %llu slots swept (%.1f per leaf)\n",
                co.r_placed, co.r_direct, co.r_slots, (double)co.r_slots / (double)std::max<int64_t>(nseg, 1));
    ctx->timing["stat_leaves"].launches += nseg; ctx->timing["stat_passes"].launches += (int64_t)co.n_passes;
    ctx->timing["stat_overflows"].launches += (int64_t)co.n_overflow;
    if (out_n) *out_n = (int64_t)co.n_out;
    if (out_distinct) *out_distinct = (int64_t)co.n_distinct;
    if (co.n_failed) { ctx->last_error = "leaf split depth exhausted"; return RFX_E_LIMIT; }
    if ((int64_t)co.n_out > cap) return RFX_E_CAP;
    return RFX_OK;
}

// k = 33..63: n two-word canonical k-mers (16-byte elements {word0, word1}) -> distinct keys with
// counts, unordered.  Same bucket structure as the k <= 31 record path: hash digits, exact
// histograms, write-combining scatters, LDS-table leaves.
int count_wide2(rfx_ctx *ctx, const void *d_elems, int64_t n, int min_cov, int max_cov, uint64_t *d_out_keys,
                int64_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n <= 0) return RFX_OK;
    std::vector<int> bits;
    plan_levels(n, false, bits, 8192.0);             // a 4096-slot table per leaf; overflowing leaves split
    if (bits.empty()) bits.push_back(0);
    DevBuf segA, segB;
    uint64_t seg_init[2] = {0, (uint64_t)n};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    DevBuf *seg_cur = &segA, *seg_next = &segB;
    int64_t nseg = 1;
    const Rec *cur = nullptr;
    RFX_TRY(partition_record_levels<1>(ctx, (const Rec *)d_elems, n, 1, bits, 0, 0, &seg_cur, &seg_next, &nseg, &cur));
    return finish_wide2(ctx, cur, (const uint64_t *)seg_cur->as<uint64_t>(), nseg, min_cov, max_cov, d_out_keys, d_out_counts,
                        cap, out_n, out_distinct);
}

// level 1 of the k = 33..63 path: packed uniform reads -> two-word elements grouped by `lv`
// (radix digit or owner) in d_dst, group offsets in d_seg_off[nb + 1]
static int wide_level1(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                       const Level &lv, Rec *d_dst, uint64_t *d_seg_off) {
    const int nb = lv.n_owners > 0 ? lv.n_owners : (1 << lv.bits);
    WideSrc ws{d_words, n_reads, nk, ceil_div(nk, W2SEG), 0, wpr, k, fc};
    ws.total = n_reads * ws.segs;
    const unsigned G = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(ws.total, W2T), (int64_t)ctx->num_cu));
    DevBuf bh, scanned;
    RFX_HIP(bh.alloc((size_t)nb * G * 8, ctx->stream));
    RFX_HIP(scanned.alloc(((size_t)nb * G + 1) * 8, ctx->stream));
    {
        ScopedTimer t(ctx, "hist1");
        hipLaunchKernelGGL(k_w2_hist, dim3(G), dim3(W2T), 0, ctx->stream, ws, lv, bh.as<uint64_t>());
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u64(ctx, bh.as<uint64_t>(), scanned.as<uint64_t>(), (int64_t)nb * G));
    hipLaunchKernelGGL(k_bin_offsets, dim3((unsigned)ceil_div(nb + 1, 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t *)scanned.as<uint64_t>(), nb, (int64_t)G, d_seg_off);
    RFX_HIP(hipGetLastError());
    {
        ScopedTimer t(ctx, "part1");
        const size_t lds = (size_t)nb * (W2B * sizeof(Rec) + 16);
        RFX_HIP(hipFuncSetAttribute((const void *)k_w2_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_w2_scatter, dim3(G), dim3(W2T), lds, ctx->stream, ws, lv, (const uint64_t *)scanned.as<uint64_t>(),
                           d_dst);
        RFX_HIP(hipGetLastError());
    }
    return RFX_OK;
}

// multi-GPU support: the two-word k-mers of packed uniform reads, grouped by owning rank
int bucket_wide_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                         int n_owners, void *d_out, int64_t cap_elems, int64_t *d_owner_off, int64_t *h_owner_off) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (n_owners < 1 || n_owners > 64) return RFX_E_ARG;
    const int64_t n = nk * n_reads;
    if (n > cap_elems) return RFX_E_CAP;
    if (n <= 0) {
        RFX_HIP(hipMemsetAsync(d_owner_off, 0, (size_t)(n_owners + 1) * 8, ctx->stream));
        if (h_owner_off) memset(h_owner_off, 0, (size_t)(n_owners + 1) * 8);
        return RFX_OK;
    }
    ctx->timing.clear();
    Level lv{};
    lv.n_owners = n_owners;
    RFX_TRY(wide_level1(ctx, d_words, n_reads, wpr, nk, k, fc, lv, (Rec *)d_out, reinterpret_cast<uint64_t *>(d_owner_off)));
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    ScopedTimer::collect(ctx);
    return RFX_OK;
}

// k = 33..63 through super-k-mer records (the default): the k = 31 / 30 front end on the CENTRAL window of
// every k-mer (a thread's 16 windows, runs that share the central minimiser), 32-byte records that carry
// the run's k + windows - 1 bases, record levels on the header, leaves that expand the records.  ~5 B per
// instance through the levels instead of 16.
// (measured, 5 Gbp at k = 63: levels 63 -> 27 ms, leaves 23 -> 44 ms, 76 ms per step against 90 for the
// element path; RFX_WIDE_RECORDS=0 selects the element path.  The leaves pay for the minimiser buckets' skew:
// DESIGN.md section 5)
static bool wide_records_enabled(int k) {
    const char *e = getenv("RFX_WIDE_RECORDS");
    return !(e && atoi(e) == 0) && k >= 33 && k <= 63;
}

static ReadSrc wide_read_src(const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc) {
    ReadSrc s{};
    const int c = (k & 1) ? 31 : 30;               // central window; (k - c) / 2 bases of the k-mer on either side
    s.words = d_words; s.n_reads = n_reads; s.wpr = wpr;
    s.wfl = (k - c) / 2;
    s.fc = fc + s.wfl; s.k = c;
    s.nk = (int)nk;
    s.segs = s.nk > 0 ? (s.nk + PK - 1) / PK : 1;
    s.n_threads = s.n_reads * s.segs;
    return s;
}

static void plan_wide_record_levels(int64_t n, std::vector<int> &bits) {
    plan_levels(n, true, bits, 6144.0);          // measured best of 3072 .. 49152 (tools/w63_sweep.sh)
    int B = 0;
    for (int b : bits) B += b;
    if (getenv("RFX_LEVEL_BITS")) return;
    // 32-byte records: the level-1 rings (8 slots) fill the LDS at 512 bins; later levels take up to 10 bits
    bits.clear();
    if (B <= 9) { bits.push_back(B); return; }
    bits.push_back(9);
    for (int rest = B - 9; rest > 0; rest -= MAX_BITS) bits.push_back(std::min(rest, MAX_BITS));
}

static int count_wide2_reads_records(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                                     int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap,
                                     int64_t *out_n, int64_t *out_distinct) {
    const int64_t n = nk * n_reads;
    ReadSrc rsrc = wide_read_src(d_words, n_reads, wpr, nk, k, fc);
    std::vector<int> bits;
    plan_wide_record_levels(n, bits);
    Level lv{};
    lv.bits = bits[0];
    DevBuf segA, segB;
    RFX_HIP(segA.alloc(((size_t)(1 << lv.bits) + 1) * 8, ctx->stream));
    WRec *recs = nullptr;
    int64_t R = 0;
    // level 1 in one sweep, as on the k <= 31 path (count_reads_superkmer)
    const int os_mode = getenv("RFX_SK_ONESWEEP") ? atoi(getenv("RFX_SK_ONESWEEP")) : 1;
    bool swept = false;
    DevBuf segE;
    if (os_mode && bits.size() >= 2 && (os_mode == 2 || rsrc.n_threads >= ((int64_t)1 << 22))) {
        RFX_HIP(segE.alloc((size_t)(1 << lv.bits) * 8, ctx->stream));
        RFX_TRY(records_onesweep<true>(ctx, rsrc, lv, 0, segA.as<uint64_t>(), segE.as<uint64_t>(), &recs, &R, &swept, "hist1", "part1"));
    }
    if (!swept) RFX_TRY(records_from_reads<true>(ctx, rsrc, lv, true, 0, nullptr, 0, segA.as<uint64_t>(), &recs, &R, "hist1", "part1"));
    const WRec *cur = nullptr;
    const uint64_t *loff = nullptr, *lend = nullptr;
    int64_t ec = 0, nleaf = 0;
    DevBuf cstart, le;
    RFX_TRY(partition_to_leaves<3>(ctx, recs, R, 0, bits, 1, lv.bits, &segA, &segB, (int64_t)1 << lv.bits,
                                   swept ? (const uint64_t *)segE.as<uint64_t>() : nullptr, &cur, &ec, &loff, &lend, &nleaf, cstart, le));
    return finish_wide2<true>(ctx, cur, loff, nleaf, min_cov, max_cov, d_out_keys, d_out_counts, cap, out_n, out_distinct, k, lend);
}

// multi-GPU, k = 33..63: the 32-byte records grouped by the owner of their minimiser (~5 B per instance
// across the exchange instead of 16), and the count of records that arrived
int bucket_wide_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                                 int n_owners, void *d_out, int64_t cap_records, int64_t *d_owner_off, int64_t *h_owner_off,
                                 int64_t *out_n_records) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (n_owners < 1 || n_owners > 64 || k < 33 || k > 63) return RFX_E_ARG;
    if (out_n_records) *out_n_records = 0;
    if (nk <= 0 || n_reads <= 0) {
        RFX_HIP(hipMemsetAsync(d_owner_off, 0, (size_t)(n_owners + 1) * 8, ctx->stream));
        if (h_owner_off) memset(h_owner_off, 0, (size_t)(n_owners + 1) * 8);
        return RFX_OK;
    }
    ReadSrc rsrc = wide_read_src(d_words, n_reads, wpr, nk, k, fc);
    Level lv{};
    lv.n_owners = n_owners;
    WRec *recs = nullptr;
    int64_t R = 0;
    const int st = records_from_reads<true>(ctx, rsrc, lv, false, 0, (WRec *)d_out, cap_records,
                                            reinterpret_cast<uint64_t *>(d_owner_off), &recs, &R, "hist1", "part1");
    if (out_n_records) *out_n_records = R;
    if (st != RFX_OK) return st;
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    return RFX_OK;
}

// the same for the 32-byte records of k = 33..63 (bucket_wide_records_by_owner)
int bucket_wide_records_by_owner_sweep(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                                       int n_owners, void *d_out, int64_t cap_records, int64_t *h_begin, int64_t *h_end,
                                       int64_t *out_n_records, bool *done) {
    *done = false;
    if (n_owners < 1 || n_owners > 64 || k < 33 || k > 63) return RFX_E_ARG;
    if (nk <= 0 || n_reads <= 0) return RFX_OK;
    return owner_sweep<true>(ctx, wide_read_src(d_words, n_reads, wpr, nk, k, fc), n_owners, d_out, cap_records, h_begin, h_end,
                             out_n_records, done);
}


int count_wide_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k, int min_cov,
                       int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                       int64_t *out_distinct) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (k < 33 || k > 63) return RFX_E_ARG;
    if (n_records <= 0) return RFX_OK;
    std::vector<int> bits;
    plan_wide_record_levels(n_instances_hint > 0 ? n_instances_hint : n_records * 6, bits);
    DevBuf segA, segB;
    uint64_t seg_init[2] = {0, (uint64_t)n_records};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    const WRec *cur = nullptr;
    const uint64_t *loff = nullptr, *lend = nullptr;
    int64_t ec = 0, nleaf = 0;
    DevBuf cstart, le;
    RFX_TRY(partition_to_leaves<3>(ctx, (const WRec *)d_records, n_records, 1, bits, 0, 0, &segA, &segB, 1, nullptr, &cur, &ec, &loff, &lend, &nleaf,
                                   cstart, le));
    return finish_wide2<true>(ctx, cur, loff, nleaf, min_cov, max_cov, d_out_keys, d_out_counts, cap, out_n, out_distinct, k, lend);
}

// k = 33..63 from packed uniform reads: level 1 straight from the reads, then count_wide2's levels/leaves
int count_wide2_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int wpr, int64_t nk, int k, int fc,
                      int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap, int64_t *out_n,
                      int64_t *out_distinct) {
    StageArena stage_arena(ctx, (size_t)256 << 20);       // temporaries of this call (see StageArena)
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    const int64_t n = nk * n_reads;
    if (n <= 0) return RFX_OK;
    if (wide_records_enabled(k))
        return count_wide2_reads_records(ctx, d_words, n_reads, wpr, nk, k, fc, min_cov, max_cov, d_out_keys, d_out_counts, cap,
                                         out_n, out_distinct);
    std::vector<int> bits;
    plan_levels(n, true, bits, 8192.0);
    if (bits[0] > 9) {                               // the level-1 rings hold 512 bins: move the excess down
        const int extra = bits[0] - 9;
        bits[0] = 9;
        if (bits.size() == 1) bits.push_back(extra); else bits[1] += extra;
        if (bits[1] > MAX_BITS) { bits.push_back(bits[1] - MAX_BITS); bits[1] = MAX_BITS; }
    }
    Level lv{};
    lv.bits = bits[0];
    const int nb = 1 << lv.bits;
    DevBuf segA, segB;
    RFX_HIP(segA.alloc(((size_t)nb + 1) * 8, ctx->stream));
    Rec *dst = (Rec *)ctx->ws_get(0, (size_t)n * sizeof(Rec));
    if (!dst) { ctx->last_error = "workspace allocation failed"; return RFX_E_HIP; }
    RFX_TRY(wide_level1(ctx, d_words, n_reads, wpr, nk, k, fc, lv, dst, segA.as<uint64_t>()));
    DevBuf *seg_cur = &segA, *seg_next = &segB;
    int64_t nseg = nb;
    const Rec *cur = nullptr;
    RFX_TRY(partition_record_levels<1>(ctx, dst, n, 0, bits, 1, lv.bits, &seg_cur, &seg_next, &nseg, &cur));
    return finish_wide2(ctx, cur, (const uint64_t *)seg_cur->as<uint64_t>(), nseg, min_cov, max_cov, d_out_keys, d_out_counts,
                        cap, out_n, out_distinct);
}

int synth_genome(rfx_ctx *ctx, uint64_t seed, int64_t genome_len, uint64_t *d_genome) {
    int64_t nw = (genome_len + 31) / 32;
    if (nw <= 0) return RFX_E_ARG;
    hipLaunchKernelGGL(k_synth_genome, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, ctx->stream,
                       splitmix64(seed ^ TAG_GENOME), nw, d_genome);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

int synth_reads(rfx_ctx *ctx, uint64_t seed, const uint64_t *d_genome, int64_t genome_len,
                int64_t first_read, int64_t n_reads, int read_len, uint32_t err, int words_per_read,
                uint64_t *d_words) {
    if (read_len > genome_len || words_per_read * 32 < read_len) return RFX_E_ARG;
    int64_t total = n_reads * words_per_read;
    if (total <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_synth_reads, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream,
                       splitmix64(seed ^ TAG_PAIRS), splitmix64(seed ^ TAG_ERRORS), d_genome, genome_len,
                       first_read, n_reads, read_len, err, words_per_read, d_words);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

}  // namespace rfx
