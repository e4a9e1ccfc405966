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

// -------------------------------------------------------- instance sources

struct Src {
    const uint64_t *kmers;     // explicit instances (reduceByKey input) or nullptr
    const uint64_t *words;     // packed uniform reads
    int wpr, nk, fc, k;        // words per read, k-mers per read, front clip
};

// second, independent hash for the local radix levels (the first one, mix64(key), picks
// the owning GPU in the multi-GPU path, so its top bits are constant inside a shard)
__device__ __forceinline__ uint64_t local_hash(uint64_t key) { return mix64(mix64(key) ^ 0x5bd1e9955bd1e995ULL); }

// ----------------------------------------------------------- radix levels

constexpr int PT = 512;               // threads per workgroup
constexpr int PK = 16;                // instances per thread
constexpr int PTILE = PT * PK;        // 8192 instances per tile
constexpr int MAX_BITS = 10;

struct Level {
    int bits;                 // digit width of this level
    int shift;                // digit = (h >> shift) & (2^bits - 1); child id = h >> shift
    int parent_shift;         // parent id = h >> parent_shift (64 -> id 0)
};

// tile -> (segment, first slot, count).  seg_off[nseg+1] element offsets of the parent
// buckets, tile_start[nseg+1] exclusive scan of tiles per segment.
__device__ __forceinline__ bool locate_tile(const uint64_t *__restrict__ seg_off,
                                            const uint64_t *__restrict__ tile_start, int64_t nseg,
                                            int64_t tile, int64_t *begin, int *count) {
    if (tile >= (int64_t)tile_start[nseg]) return false;
    int64_t lo = 0, hi = nseg;             // last seg with tile_start[seg] <= tile
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    int64_t b = (int64_t)seg_off[lo] + (tile - (int64_t)tile_start[lo]) * PTILE;
    int64_t e = (int64_t)seg_off[lo + 1];
    *begin = b;
    *count = (int)((e - b) < PTILE ? (e - b) : PTILE);
    return true;
}

__global__ void k_tiles_per_seg(const uint64_t *__restrict__ seg_off, int64_t nseg,
                                uint64_t *__restrict__ tiles) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg) tiles[s] = (seg_off[s + 1] - seg_off[s] + PTILE - 1) / PTILE;
}

// Loads this thread's PK instances of the tile (slot = begin + i*PT + tid) into key[].
template <bool FROM_READS>
__device__ __forceinline__ void load_tile(const Src &src, int64_t begin, int count, uint64_t (&key)[PK]) {
    if (FROM_READS) {
        int64_t slot = begin + threadIdx.x;
        int64_t r = slot / src.nk;
        int p = (int)(slot - r * src.nk);
#pragma unroll
        for (int i = 0; i < PK; i++) {
            int idx = i * PT + threadIdx.x;
            if (idx < count) key[i] = canonical(kmer_at(src.words + r * src.wpr, src.fc + p, src.k), src.k);
            else key[i] = 0;
            p += PT;
            int q = p / src.nk;
            r += q; p -= q * src.nk;
        }
    } else {
#pragma unroll
        for (int i = 0; i < PK; i++) {
            int idx = i * PT + threadIdx.x;
            key[i] = idx < count ? src.kmers[begin + idx] : 0;
        }
    }
}

template <bool FROM_READS>
__global__ __launch_bounds__(PT) void k_level_hist(Src src, const uint64_t *__restrict__ seg_off,
                                                   const uint64_t *__restrict__ tile_start, int64_t nseg,
                                                   Level lv, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[1 << MAX_BITS];
    int64_t begin; int count;
    if (!locate_tile(seg_off, tile_start, nseg, blockIdx.x, &begin, &count)) return;
    const int nb = 1 << lv.bits;
    for (int i = threadIdx.x; i < nb; i += PT) h[i] = 0;
    __syncthreads();
    uint64_t key[PK];
    load_tile<FROM_READS>(src, begin, count, key);
    uint64_t parent = 0;
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) {
            uint64_t hh = local_hash(key[i]);
            atomicAdd(&h[(hh >> lv.shift) & (nb - 1)], 1u);
            if (i == 0) parent = lv.parent_shift >= 64 ? 0 : (hh >> lv.parent_shift);
        }
    }
    __shared__ uint64_t s_parent;
    if (threadIdx.x == 0) s_parent = parent;       // thread 0 always holds a valid instance
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += PT) {
        uint32_t c = h[i];
        if (c) atomicAdd(&hist[(s_parent << lv.bits) | (uint64_t)i], (unsigned long long)c);
    }
}

// LDS: sorted tile (PTILE keys) + per-digit count / local offset / reserved global base
template <bool FROM_READS>
__global__ __launch_bounds__(PT) void k_level_scatter(Src src, const uint64_t *__restrict__ seg_off,
                                                      const uint64_t *__restrict__ tile_start, int64_t nseg,
                                                      Level lv, unsigned long long *__restrict__ cursor,
                                                      uint64_t *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem);                      // PTILE
    uint64_t *gbase = skey + PTILE;                                           // 2^bits
    uint32_t *cnt = reinterpret_cast<uint32_t *>(gbase + (1 << MAX_BITS));    // 2^bits
    uint32_t *loff = cnt;      // aliases cnt: every count is read before the scan's barrier
    __shared__ uint32_t wsum[PT / 64];
    __shared__ uint64_t s_parent;

    int64_t begin; int count;
    if (!locate_tile(seg_off, tile_start, nseg, blockIdx.x, &begin, &count)) return;
    const int nb = 1 << lv.bits;
    for (int i = threadIdx.x; i < nb; i += PT) cnt[i] = 0;
    __syncthreads();

    uint64_t key[PK];
    uint32_t rank[PK];
    load_tile<FROM_READS>(src, begin, count, key);
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) {
            uint64_t hh = local_hash(key[i]);
            unsigned d = (unsigned)((hh >> lv.shift) & (nb - 1));
            rank[i] = atomicAdd(&cnt[d], 1u) | (d << 16);       // rank < 8192 fits 16 bits
            if (i == 0 && threadIdx.x == 0) s_parent = lv.parent_shift >= 64 ? 0 : (hh >> lv.parent_shift);
        }
    }
    __syncthreads();
    // local exclusive offsets (nb <= 1024 = 2 per thread at most) and global reservation
    {
        uint32_t c0 = 0, c1 = 0;
        int d0 = 2 * threadIdx.x, d1 = d0 + 1;
        if (d0 < nb) c0 = cnt[d0];
        if (d1 < nb) c1 = cnt[d1];
        uint32_t ex = block_exclusive_scan(c0 + c1, wsum, nullptr);
        if (d0 < nb) {
            loff[d0] = ex;
            gbase[d0] = c0 ? atomicAdd(&cursor[(s_parent << lv.bits) | (uint64_t)d0], (unsigned long long)c0) : 0;
        }
        if (d1 < nb) {
            loff[d1] = ex + c0;
            gbase[d1] = c1 ? atomicAdd(&cursor[(s_parent << lv.bits) | (uint64_t)d1], (unsigned long long)c1) : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) {
            unsigned d = rank[i] >> 16;
            skey[loff[d] + (rank[i] & 0xFFFFu)] = key[i];
        }
    }
    __syncthreads();
    // copy out: consecutive lanes write consecutive elements of a digit's run
    for (int i = threadIdx.x; i < count; i += PT) {
        uint64_t kk = skey[i];
        unsigned d = (unsigned)((local_hash(kk) >> lv.shift) & (nb - 1));
        out[gbase[d] + (uint64_t)(i - loff[d])] = kk;
    }
}

// ------------------------------------------------------------------- leaves

constexpr int LT = 256;                 // threads per leaf workgroup
constexpr int LCAP = 4096;              // hash slots
constexpr int LFULL = (LCAP * 3) / 4;   // give up on a sub-pass beyond this many distinct keys
constexpr uint64_t EMPTY = ~0ULL;
constexpr int LSTACK = 48;

struct CountOut {
    unsigned long long n_out;        // survivors appended (may exceed cap)
    unsigned long long n_distinct;   // distinct k-mers seen
    unsigned long long n_failed;     // leaves that ran out of split depth (must stay 0)
};

__global__ __launch_bounds__(LT) void k_leaf_count(const uint64_t *__restrict__ keys,
                                                   const uint64_t *__restrict__ leaf_off, int64_t nleaf,
                                                   int min_cov, int max_cov, int apply_filter,
                                                   uint64_t *__restrict__ out_keys, int32_t *__restrict__ out_counts,
                                                   unsigned long long cap, CountOut *__restrict__ co) {
    __shared__ unsigned long long tkey[LCAP];
    __shared__ uint32_t tcnt[LCAP];
    __shared__ uint32_t stackS[LSTACK], stacks[LSTACK];
    __shared__ int sp;
    __shared__ uint32_t n_dist, overflow, n_emit, emit_pos;
    __shared__ unsigned long long g_emit;

    for (int64_t leaf = blockIdx.x; leaf < nleaf; leaf += gridDim.x) {
        const uint64_t begin = leaf_off[leaf], end = leaf_off[leaf + 1];
        if (begin == end) continue;                // uniform per block
        if (threadIdx.x == 0) { sp = 1; stackS[0] = 1; stacks[0] = 0; }
        __syncthreads();
        while (true) {
            __syncthreads();
            if (sp == 0) break;
            const uint32_t S = stackS[sp - 1], s = stacks[sp - 1];
            __syncthreads();
            if (threadIdx.x == 0) { sp--; n_dist = 0; overflow = 0; n_emit = 0; emit_pos = 0; }
            for (int i = threadIdx.x; i < LCAP; i += LT) { tkey[i] = EMPTY; tcnt[i] = 0; }
            __syncthreads();
            for (uint64_t i = begin + threadIdx.x; i < end; i += LT) {
                const uint64_t key = keys[i];
                const uint64_t h = local_hash(key);
                if (S > 1 && ((uint32_t)(h >> 12) & (S - 1)) != s) continue;
                uint32_t slot = (uint32_t)h & (LCAP - 1);
                for (int probe = 0; probe < LCAP; probe++) {
                    unsigned long long prev = atomicCAS(&tkey[slot], EMPTY, (unsigned long long)key);
                    if (prev == EMPTY) {
                        if (atomicAdd(&n_dist, 1u) >= (uint32_t)LFULL) overflow = 1;
                        atomicAdd(&tcnt[slot], 1u);
                        break;
                    }
                    if (prev == key) { atomicAdd(&tcnt[slot], 1u); break; }
                    slot = (slot + 1) & (LCAP - 1);
                    if (probe == LCAP - 1) overflow = 1;
                }
                if (overflow) break;               // the sub-pass is abandoned anyway
            }
            __syncthreads();
            if (overflow) {
                if (threadIdx.x == 0) {
                    if (sp + 2 <= LSTACK && S < (1u << 20)) {
                        stackS[sp] = 2 * S; stacks[sp] = s + S; sp++;
                        stackS[sp] = 2 * S; stacks[sp] = s;     sp++;
                    } else {
                        atomicAdd(&co->n_failed, 1ULL);
                    }
                }
                continue;
            }
            // emit survivors of this sub-pass
            for (int i = threadIdx.x; i < LCAP; i += LT) {
                if (tkey[i] != EMPTY) {
                    int32_t c = (int32_t)tcnt[i];
                    if (!apply_filter || (c >= min_cov && c <= max_cov)) atomicAdd(&n_emit, 1u);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                g_emit = n_emit ? atomicAdd(&co->n_out, (unsigned long long)n_emit) : 0;
                atomicAdd(&co->n_distinct, (unsigned long long)n_dist);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < LCAP; i += LT) {
                if (tkey[i] != EMPTY) {
                    int32_t c = (int32_t)tcnt[i];
                    if (!apply_filter || (c >= min_cov && c <= max_cov)) {
                        unsigned long long pos = g_emit + atomicAdd(&emit_pos, 1u);
                        if (pos < cap) { out_keys[pos] = tkey[i]; out_counts[pos] = c; }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------ owner buckets (multi-GPU)

__global__ __launch_bounds__(PT) void k_owner_hist(Src src, int64_t n, int n_owners,
                                                   unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    int64_t begin = (int64_t)blockIdx.x * PTILE;
    int count = (int)((n - begin) < PTILE ? (n - begin) : PTILE);
    uint64_t key[PK];
    load_tile<true>(src, begin, count, key);
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) atomicAdd(&h[(unsigned)__umul64hi(mix64(key[i]), (uint64_t)n_owners)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_owners && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ __launch_bounds__(PT) void k_owner_scatter(Src src, int64_t n, int n_owners,
                                                      unsigned long long *__restrict__ cursor,
                                                      uint64_t *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem);
    __shared__ uint32_t cnt[64], loff[64];
    __shared__ uint64_t gbase[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    int64_t begin = (int64_t)blockIdx.x * PTILE;
    int count = (int)((n - begin) < PTILE ? (n - begin) : PTILE);
    uint64_t key[PK];
    uint32_t rank[PK];
    load_tile<true>(src, begin, count, key);
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) {
            unsigned d = (unsigned)__umul64hi(mix64(key[i]), (uint64_t)n_owners);
            rank[i] = atomicAdd(&cnt[d], 1u) | (d << 16);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int d = 0; d < n_owners; d++) {
            loff[d] = run; run += cnt[d];
            gbase[d] = cnt[d] ? atomicAdd(&cursor[d], (unsigned long long)cnt[d]) : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PK; i++) {
        int idx = i * PT + threadIdx.x;
        if (idx < count) { unsigned d = rank[i] >> 16; skey[loff[d] + (rank[i] & 0xFFFFu)] = key[i]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += PT) {
        uint64_t kk = skey[i];
        unsigned d = (unsigned)__umul64hi(mix64(kk), (uint64_t)n_owners);
        out[gbase[d] + (uint64_t)(i - loff[d])] = kk;
    }
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

size_t scatter_lds_bytes() { return (size_t)PTILE * 8 + (size_t)(1 << MAX_BITS) * (8 + 4); }

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

int64_t count_workspace_bytes(int64_t n_kmers) { return 2 * n_kmers * 8 + (int64_t)(64 << 20); }

static void plan_levels(int64_t n, bool from_reads, std::vector<int> &bits) {
    bits.clear();
    const double target = 8192.0;
    int B = 0;
    if ((double)n > target) B = (int)std::ceil(std::log2((double)n / target));
    if (B > 3 * MAX_BITS) B = 3 * MAX_BITS;
    int L = (B + MAX_BITS - 1) / MAX_BITS;
    for (int l = 0; l < L; l++) bits.push_back(B / L + (l < B % L ? 1 : 0));
    if (bits.empty() && from_reads) bits.push_back(0);   // materialise the instances once
}

int count_filter(rfx_ctx *ctx, const ReadStore *reads, const uint64_t *d_kmers, int64_t n,
                 int min_cov, int max_cov, int twin, void *ws, int64_t ws_bytes,
                 uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                 int64_t *out_n, int64_t *out_distinct) {
    (void)ws; (void)ws_bytes;
    ctx->timing.clear();
    Src src{};
    int k_bits = 64;
    const bool from_reads = reads != nullptr;
    if (from_reads) {
        src.words = reads->words; src.wpr = reads->words_per_read; src.fc = reads->front_clip; src.k = reads->k;
        src.nk = (int)kmers_per_read(reads->read_len, reads->k, reads->front_clip, reads->end_clip);
        n = (int64_t)src.nk * reads->n_reads;
        k_bits = 2 * reads->k;
    } else {
        src.kmers = d_kmers;
    }
    if (out_n) *out_n = 0;
    if (out_distinct) *out_distinct = 0;
    if (n <= 0) return RFX_OK;

    std::vector<int> bits;
    plan_levels(n, from_reads, bits);

    static bool attr_set = false;
    if (!attr_set) {
        RFX_HIP(hipFuncSetAttribute((const void *)k_level_scatter<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)scatter_lds_bytes()));
        RFX_HIP(hipFuncSetAttribute((const void *)k_level_scatter<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)scatter_lds_bytes()));
        attr_set = true;
    }

    DevBuf bufA, bufB, segA, segB, tiles, tile_start, hist, cursor, co_buf;
    uint64_t seg_init[2] = {0, (uint64_t)n};
    RFX_HIP(segA.alloc(2 * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(segA.p, seg_init, 16, hipMemcpyHostToDevice, ctx->stream));
    RFX_HIP(hipStreamSynchronize(ctx->stream));     // seg_init lives on the stack
    int64_t nseg = 1;
    DevBuf *seg_cur = &segA, *seg_next = &segB;
    DevBuf *out_buf = &bufA, *in_buf = &bufB;
    const uint64_t *cur_arr = d_kmers;
    bool cur_from_reads = from_reads;
    int used_bits = 0;
    for (size_t l = 0; l < bits.size(); l++) {
        Level lv;
        lv.bits = bits[l];
        lv.parent_shift = 64 - used_bits;
        used_bits += bits[l];
        lv.shift = 64 - used_bits;
        if (lv.shift >= 64) lv.shift = 63;            // bits == 0 on the first level: digit mask is 0
        const int64_t nchild = nseg << lv.bits;
        const int64_t max_tiles = ceil_div(n, PTILE) + nseg;
        RFX_HIP(tiles.alloc((size_t)nseg * 8, ctx->stream));
        RFX_HIP(tile_start.alloc((size_t)(nseg + 1) * 8, ctx->stream));
        hipLaunchKernelGGL(k_tiles_per_seg, dim3((unsigned)ceil_div(nseg, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)seg_cur->as<uint64_t>(), nseg, tiles.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u64(ctx, tiles.as<uint64_t>(), tile_start.as<uint64_t>(), nseg));
        RFX_HIP(hist.alloc((size_t)nchild * 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(hist.p, 0, (size_t)nchild * 8, ctx->stream));
        RFX_HIP(seg_next->alloc((size_t)(nchild + 1) * 8, ctx->stream));
        Src s2 = src;
        if (!cur_from_reads) { s2.kmers = cur_arr; s2.words = nullptr; }
        {
            ScopedTimer t(ctx, l == 0 ? "hist1" : l == 1 ? "hist2" : "hist3");
            if (cur_from_reads)
                hipLaunchKernelGGL(k_level_hist<true>, dim3((unsigned)max_tiles), dim3(PT), 0, ctx->stream, s2,
                                   (const uint64_t *)seg_cur->as<uint64_t>(), (const uint64_t *)tile_start.as<uint64_t>(),
                                   nseg, lv, hist.as<unsigned long long>());
            else
                hipLaunchKernelGGL(k_level_hist<false>, dim3((unsigned)max_tiles), dim3(PT), 0, ctx->stream, s2,
                                   (const uint64_t *)seg_cur->as<uint64_t>(), (const uint64_t *)tile_start.as<uint64_t>(),
                                   nseg, lv, hist.as<unsigned long long>());
            RFX_HIP(hipGetLastError());
        }
        RFX_TRY(exclusive_scan_u64(ctx, hist.as<uint64_t>(), seg_next->as<uint64_t>(), nchild));
        RFX_HIP(cursor.alloc((size_t)nchild * 8, ctx->stream));
        RFX_HIP(hipMemcpyAsync(cursor.p, seg_next->p, (size_t)nchild * 8, hipMemcpyDeviceToDevice, ctx->stream));
        RFX_HIP(out_buf->alloc((size_t)n * 8, ctx->stream));
        {
            ScopedTimer t(ctx, l == 0 ? "part1" : l == 1 ? "part2" : "part3");
            if (cur_from_reads)
                hipLaunchKernelGGL(k_level_scatter<true>, dim3((unsigned)max_tiles), dim3(PT), scatter_lds_bytes(),
                                   ctx->stream, s2, (const uint64_t *)seg_cur->as<uint64_t>(),
                                   (const uint64_t *)tile_start.as<uint64_t>(), nseg, lv,
                                   cursor.as<unsigned long long>(), out_buf->as<uint64_t>());
            else
                hipLaunchKernelGGL(k_level_scatter<false>, dim3((unsigned)max_tiles), dim3(PT), scatter_lds_bytes(),
                                   ctx->stream, s2, (const uint64_t *)seg_cur->as<uint64_t>(),
                                   (const uint64_t *)tile_start.as<uint64_t>(), nseg, lv,
                                   cursor.as<unsigned long long>(), out_buf->as<uint64_t>());
            RFX_HIP(hipGetLastError());
        }
        cur_arr = out_buf->as<uint64_t>();
        cur_from_reads = false;
        std::swap(out_buf, in_buf);
        if (l + 1 < bits.size()) out_buf->release();   // the buffer two levels back is dead
        std::swap(seg_cur, seg_next);
        nseg = nchild;
    }

    RFX_HIP(co_buf.alloc(sizeof(CountOut), ctx->stream));
    RFX_HIP(hipMemsetAsync(co_buf.p, 0, sizeof(CountOut), ctx->stream));
    const int apply = !(twin == RFX_TWIN_RDD && min_cov <= 1);    // P/ReflexivMain.java:160
    {
        ScopedTimer t(ctx, "leaf");
        int64_t grid = std::min<int64_t>(nseg, (int64_t)1 << 22);
        hipLaunchKernelGGL(k_leaf_count, dim3((unsigned)grid), dim3(LT), 0, ctx->stream, cur_arr,
                           (const uint64_t *)seg_cur->as<uint64_t>(), nseg, min_cov, max_cov, apply, d_out_keys,
                           d_out_counts, (unsigned long long)cap, co_buf.as<CountOut>());
        RFX_HIP(hipGetLastError());
    }
    CountOut co{};
    RFX_HIP(hipMemcpyAsync(&co, co_buf.p, sizeof co, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipStreamSynchronize(ctx->stream));
    if (out_n) *out_n = (int64_t)co.n_out;
    if (out_distinct) *out_distinct = (int64_t)co.n_distinct;
    if (co.n_failed) { ctx->last_error = "leaf split depth exhausted"; ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
    if ((int64_t)co.n_out > cap) { ScopedTimer::collect(ctx); return RFX_E_CAP; }
    if ((int64_t)co.n_out > (int64_t)0xFFFFFFFFLL) { ScopedTimer::collect(ctx); return RFX_E_LIMIT; }
    // ascending k-mer order (order contract B.0)
    {
        DevBuf tk, tv;
        RFX_HIP(tk.alloc((size_t)co.n_out * 8, ctx->stream));
        RFX_HIP(tv.alloc((size_t)co.n_out * 4, ctx->stream));
        ScopedTimer t(ctx, "sort");
        RFX_TRY(sort_pairs(ctx, d_out_keys, reinterpret_cast<uint32_t *>(d_out_counts), (int64_t)co.n_out,
                           from_reads ? k_bits : 64, tk.as<uint64_t>(), tv.as<uint32_t>()));
        t.stop();
        RFX_HIP(hipStreamSynchronize(ctx->stream));
    }
    ScopedTimer::collect(ctx);
    return RFX_OK;
}

int bucket_by_owner(rfx_ctx *ctx, const ReadStore *reads, int n_owners, uint64_t *d_out, int64_t cap,
                    int64_t *d_owner_off, int64_t *h_owner_off) {
    if (n_owners < 1 || n_owners > 64) return RFX_E_ARG;
    Src src{};
    src.words = reads->words; src.wpr = reads->words_per_read; src.fc = reads->front_clip; src.k = reads->k;
    src.nk = (int)kmers_per_read(reads->read_len, reads->k, reads->front_clip, reads->end_clip);
    const int64_t n = (int64_t)src.nk * reads->n_reads;
    if (n > cap) return RFX_E_CAP;
    static bool attr_set = false;
    if (!attr_set) {
        RFX_HIP(hipFuncSetAttribute((const void *)k_owner_scatter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    PTILE * 8));
        attr_set = true;
    }
    DevBuf hist, cursor;
    RFX_HIP(hist.alloc(64 * 8, ctx->stream));
    RFX_HIP(cursor.alloc(64 * 8, ctx->stream));
    RFX_HIP(hipMemsetAsync(hist.p, 0, 64 * 8, ctx->stream));
    const int64_t tiles = ceil_div(n, PTILE);
    if (tiles > 0) {
        hipLaunchKernelGGL(k_owner_hist, dim3((unsigned)tiles), dim3(PT), 0, ctx->stream, src, n, n_owners,
                           hist.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u64(ctx, hist.as<uint64_t>(), reinterpret_cast<uint64_t *>(d_owner_off), n_owners));
    RFX_HIP(hipMemcpyAsync(cursor.p, d_owner_off, (size_t)n_owners * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (tiles > 0) {
        hipLaunchKernelGGL(k_owner_scatter, dim3((unsigned)tiles), dim3(PT), PTILE * 8, ctx->stream, src, n, n_owners,
                           cursor.as<unsigned long long>(), d_out);
        RFX_HIP(hipGetLastError());
    }
    if (h_owner_off) {
        RFX_HIP(hipMemcpyAsync(h_owner_off, d_owner_off, (size_t)(n_owners + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RFX_OK;
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
