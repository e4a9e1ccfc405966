// rfx_sort.hip -- stable LSD radix sort of (uint64 key, uint32 value) pairs in HBM.
//
// This is the MI355X stand-in for Spark's sortByKey()/sort("k-1") shuffle
// (P/ReflexivMain.java:179,191,211,235,247,286): a global STABLE sort by the (k-1)-mer key;
// stability is what carries the order contract's "ties keep arrival order".
//
// 8-bit digits.  Per pass: per-tile digit histogram -> exclusive scan of the
// [digit][tile] table -> scatter with a stable in-tile rank.  The in-tile rank uses
// wave64 ballots to find, for every lane, the lanes of the same wave that hold the same
// digit (match-any), so no key ever goes through LDS; LDS only holds the per-wave digit
// counters.  Small inputs (<= one tile) take a single-workgroup path.
#include "rfx_internal.h"
#include "rfx_device.h"

namespace {

constexpr int ST = 256;             // threads per workgroup (4 waves)
constexpr int SI = 8;               // keys per thread
constexpr int STILE = ST * SI;      // 2048 keys per tile
constexpr int SW = ST / 64;         // waves per workgroup

// Second MSD level (large inputs): tiles must not straddle the buckets of the first level.  Bucket b of the 256
// has nt_b = ceil(size_b / STILE) "virtual tiles"; vtile v = tstart[b] + t covers [bstart[b] + t*STILE, ...) inside
// the bucket, and its histogram row sits at table[tstart[b]*D + d*nt_b + t]: scanned flat, that layout is the order
// (bucket, digit, tile), i.e. the output position of every (tile, digit) group.
struct SegMap {
    const uint32_t *bstart;     // [NP + 1] first position of every parent bucket (bstart[NP] = n)
    const uint32_t *tstart;     // [NP + 1] first virtual tile of every bucket (tstart[NP] = number of virtual tiles)
    int D;                      // digits of this level (power of two <= 256)
    int NP = 256;               // parent buckets: 256 after one level, 65536 after two (third level, n > 2^26)
};
struct SegPos { int64_t begin, end, row; int64_t stride; bool ok; };
__device__ __forceinline__ SegPos seg_pos(const SegMap &m, uint32_t v) {
    SegPos p; p.ok = v < m.tstart[m.NP];
    if (!p.ok) { p.begin = p.end = p.row = 0; p.stride = 1; return p; }
    int lo = 0, hi = m.NP;                             // last b with tstart[b] <= v
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (m.tstart[mid] <= v) lo = mid; else hi = mid; }
    const uint32_t t = v - m.tstart[lo];
    p.begin = (int64_t)m.bstart[lo] + (int64_t)t * STILE;
    p.end = (int64_t)m.bstart[lo + 1];
    if (p.end > p.begin + STILE) p.end = p.begin + STILE;
    p.stride = (int64_t)(m.tstart[lo + 1] - m.tstart[lo]);
    p.row = (int64_t)m.tstart[lo] * m.D + t;           // + d * stride
    return p;
}

template <bool SEG>
__global__ __launch_bounds__(ST) void k_hist(const uint64_t *__restrict__ keys, int64_t n, int shift,
                                             uint32_t *__restrict__ table, int64_t ntiles, SegMap sm) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * STILE, end = n, row = blockIdx.x, stride = ntiles;
    unsigned mask = 255u;
    if (SEG) {
        const SegPos p = seg_pos(sm, blockIdx.x);
        if (!p.ok) return;                              // (uniform)
        base = p.begin; end = p.end; row = p.row; stride = p.stride; mask = (unsigned)sm.D - 1u;
    }
#pragma unroll
    for (int i = 0; i < SI; i++) {
        int64_t idx = base + (int64_t)i * ST + threadIdx.x;
        if (idx < end) atomicAdd(&h[(unsigned)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (!SEG || (int)threadIdx.x < sm.D) table[(int64_t)threadIdx.x * stride + row] = h[threadIdx.x];
}

template <bool SEG>
__global__ __launch_bounds__(ST) void k_scatter(const uint64_t *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, int64_t n, int shift,
                                                const uint64_t *__restrict__ offs, int64_t ntiles,
                                                uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals,
                                                const uint32_t *__restrict__ table, SegMap sm) {
    __shared__ volatile uint32_t wcnt[SW][256];
    __shared__ uint64_t wbase[SW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t tbase = (int64_t)blockIdx.x * STILE, tend = n, row = blockIdx.x, stride = ntiles;
    unsigned mask = 255u;
    if (SEG) {
        const SegPos p = seg_pos(sm, blockIdx.x);
        if (!p.ok) return;                              // (uniform)
        tbase = p.begin; tend = p.end; row = p.row; stride = p.stride; mask = (unsigned)sm.D - 1u;
    }
    for (int i = threadIdx.x; i < SW * 256; i += ST) ((volatile uint32_t *)wcnt)[i] = 0;
    __syncthreads();

    // wave `wave` owns the contiguous chunk [wave*512, wave*512+512) of the tile, read in
    // 8 rounds of 64 consecutive keys: arrival order = (round, lane)
    const int64_t cbase = tbase + (int64_t)wave * (64 * SI);
    uint64_t key[SI];
    uint32_t val[SI], rank[SI];
    const uint64_t lt = (1ULL << lane) - 1;
#pragma unroll
    for (int r = 0; r < SI; r++) {
        int64_t idx = cbase + r * 64 + lane;
        bool ok = idx < tend;
        key[r] = ok ? keys[idx] : 0;
        val[r] = ok ? vals[idx] : 0;
        unsigned d = (unsigned)(key[r] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t before = 0;
        if (ok) {
            before = wcnt[wave][d];
            rank[r] = before + (uint32_t)__popcll(peers & lt);
        }
        // every peer has read the counter (one wave-wide LDS read) before the leader bumps it
        if (ok && (peers & lt) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {
        const int d = threadIdx.x;                 // ST == 256 digits
        uint64_t run;
        if (SEG) {
            run = d < sm.D ? offs[(int64_t)d * stride + row] : 0;
        } else if (offs) {
            run = offs[(int64_t)d * ntiles + blockIdx.x];
        } else {
            // few tiles: every workgroup scans the [digit][tile] histogram table itself -- two launches
            // per digit pass instead of five (the launch chain is what a 30 K-record sort costs)
            uint32_t before = 0, tot = 0;
#pragma unroll 8
            for (int64_t t = 0; t < ntiles; t++) {
                const uint32_t c = table[(int64_t)d * ntiles + t];
                tot += c;
                if (t < blockIdx.x) before += c;
            }
            __shared__ uint32_t wsum_inline[SW];
            run = (uint64_t)rfxd::block_exclusive_scan(tot, wsum_inline, nullptr) + before;
        }
#pragma unroll
        for (int w = 0; w < SW; w++) { wbase[w][d] = run; run += wcnt[w][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SI; r++) {
        int64_t idx = cbase + r * 64 + lane;
        if (idx < tend) {
            unsigned d = (unsigned)(key[r] >> shift) & mask;
            uint64_t dst = wbase[wave][d] + rank[r];
            okeys[dst] = key[r];
            ovals[dst] = val[r];
        }
    }
}

// level-1 bucket starts from the scanned [digit][tile] table, and the virtual tiles of the second level
__global__ __launch_bounds__(256) void k_l2_setup(const uint64_t *__restrict__ offs1, int64_t ntiles, int64_t n,
                                                  uint32_t *__restrict__ bstart, uint32_t *__restrict__ tstart) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t bs[257];
    bs[threadIdx.x] = (uint32_t)offs1[(int64_t)threadIdx.x * ntiles];
    if (threadIdx.x == 0) bs[256] = (uint32_t)n;
    __syncthreads();
    const uint32_t size = bs[threadIdx.x + 1] - bs[threadIdx.x];
    const uint32_t nt = (size + STILE - 1) / STILE;
    uint32_t tot = 0;
    const uint32_t ex = rfxd::block_exclusive_scan(nt, wsum, &tot);
    bstart[threadIdx.x] = bs[threadIdx.x];
    tstart[threadIdx.x] = ex;
    if (threadIdx.x == 0) { bstart[256] = (uint32_t)n; tstart[256] = tot; }
}

// virtual tiles of a level whose parents are the NP buckets bounded by `bounds` (one workgroup; NP up to 65536)
__global__ __launch_bounds__(1024) void k_seg_setup(const uint32_t *__restrict__ bounds, int NP, uint32_t *__restrict__ tstart) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < NP; base += 1024) {
        const int b = base + threadIdx.x;
        const uint32_t nt = b < NP ? (bounds[b + 1] - bounds[b] + STILE - 1) / STILE : 0u;
        uint32_t tot = 0;
        const uint32_t ex = rfxd::block_exclusive_scan(nt, wsum, &tot);
        if (b < NP) tstart[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) tstart[NP] = carry;
}

// starts of the 256*D final buckets (bucket b, digit d) from the scanned level-2 table, and the largest bucket
__global__ void k_l3_bounds(const uint64_t *__restrict__ offs2, SegMap sm, int64_t n, uint32_t *__restrict__ b3,
                            uint32_t *__restrict__ maxc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int NB = sm.NP * sm.D;
    if (i > NB) return;
    auto start_of = [&](int q) -> uint32_t {
        if (q >= NB) return (uint32_t)n;
        const int b = q / sm.D, d = q - b * sm.D;
        const int64_t stride = (int64_t)(sm.tstart[b + 1] - sm.tstart[b]);
        if (stride == 0) return sm.bstart[b];           // empty level-1 bucket
        return (uint32_t)offs2[(int64_t)sm.tstart[b] * sm.D + (int64_t)d * stride];
    };
    const uint32_t s0 = start_of(i);
    b3[i] = s0;
    if (i < NB) atomicMax(maxc, start_of(i + 1) - s0);
}

// One workgroup sorts up to STILE pairs entirely on chip (all passes), stable: (ikeys, ivals)[0..n) ->
// (okeys, ovals)[0..n), in place when they are the same arrays.
// `bits` = low key bits that are still unsorted (the bits above them are equal over the whole tile or do not exist).
// First try: ONE stable pass on the top 8 of those bits, then thread d finishes sub-bucket d (a couple of pairs when the
// keys are spread) by a stable insertion sort on the whole key -- 4 barriers instead of 4 per 8 bits.  A sub-bucket
// longer than MSD_INS pairs (keys that share long prefixes: repeats) sends the tile down the LSD passes instead.
constexpr int MSD_INS = 24;
__device__ __forceinline__ void sort_tile_lds(const uint64_t *keys, const uint32_t *vals, int n, int bits,
                                              uint64_t *okeys, uint32_t *ovals) {
    __shared__ uint64_t sk[2][STILE];
    __shared__ uint32_t sv[2][STILE];
    __shared__ volatile uint32_t wcnt[SW][256];
    __shared__ uint32_t wbase[SW][256];
    __shared__ uint32_t dstart[257];
    __shared__ int too_long;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ULL << lane) - 1;
    for (int i = threadIdx.x; i < n; i += ST) { sk[0][i] = keys[i]; sv[0][i] = vals[i]; }
    if (threadIdx.x == 0) too_long = 0;
    int cur = 0;
    // the waves take equal contiguous chunks of the n pairs (a multiple of 64 each), so a bucket of a few hundred
    // pairs costs a few rounds per pass, not the eight of a full tile; arrival order = (wave, round, lane)
    const int per_wave = ((n + SW * 64 - 1) / (SW * 64)) * 64;
    const int rounds = per_wave >> 6;
    const int passes = (bits + 7) / 8;
    const int msd_shift = bits > 8 ? bits - 8 : 0;
    for (int p = -1; p < passes; p++) {
        // p = -1: the MSD attempt (reads sk[0], writes sk[1]); p >= 0: the LSD passes, restarted from sk[0]
        const int shift = p < 0 ? msd_shift : 8 * p;
        const unsigned dmask = (p < 0 && bits < 8) ? ((1u << bits) - 1u) : 255u;
        if (p == 0) cur = 0;
        for (int i = threadIdx.x; i < SW * 256; i += ST) ((volatile uint32_t *)wcnt)[i] = 0;
        __syncthreads();
        uint32_t rank[SI];
        const int cbase = wave * per_wave;
#pragma unroll
        for (int r = 0; r < SI; r++) {
            if (r < rounds) {                           // (uniform)
                int idx = cbase + r * 64 + lane;
                bool ok = idx < n;
                unsigned d = ok ? (unsigned)(sk[cur][idx] >> shift) & dmask : 0u;
                uint64_t peers = __ballot(ok);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    uint64_t m = __ballot((d >> b) & 1u);
                    peers &= ((d >> b) & 1u) ? m : ~m;
                }
                uint32_t before = 0;
                if (ok) { before = wcnt[wave][d]; rank[r] = before + (uint32_t)__popcll(peers & lt); }
                if (ok && (peers & lt) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
            }
        }
        __syncthreads();
        // exclusive scan over (digit, wave): thread d sums its digit, then a block scan over digits
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) tot += wcnt[w][threadIdx.x];
        __shared__ uint32_t wsum[SW];
        uint32_t dbase = rfxd::block_exclusive_scan(tot, wsum, nullptr);
        if (p < 0) { dstart[threadIdx.x] = dbase; if (threadIdx.x == 255) dstart[256] = dbase + tot; }
#pragma unroll
        for (int w = 0; w < SW; w++) { wbase[w][threadIdx.x] = dbase; dbase += wcnt[w][threadIdx.x]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SI; r++) {
            if (r < rounds) {
                int idx = cbase + r * 64 + lane;
                if (idx < n) {
                    uint64_t kk = sk[cur][idx];
                    unsigned d = (unsigned)(kk >> shift) & dmask;
                    uint32_t dst = wbase[wave][d] + rank[r];
                    sk[cur ^ 1][dst] = kk;
                    sv[cur ^ 1][dst] = sv[cur][idx];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
        if (p < 0) {
            // sub-bucket d = [dstart[d], dstart[d+1]) of sk[1]; finish it, or give up
            const uint32_t b = dstart[threadIdx.x], e = dstart[threadIdx.x + 1];
            if (e - b > (uint32_t)MSD_INS) too_long = 1;
            __syncthreads();
            if (!too_long) {
                // (ordered by the low `bits` bits only, as the LSD passes are: bits above key_bits do not take part)
                const uint64_t low = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
                for (uint32_t i = b + 1; i < e; i++) {
                    const uint64_t x = sk[1][i];
                    const uint32_t xv = sv[1][i];
                    uint32_t j = i;
                    while (j > b && (sk[1][j - 1] & low) > (x & low)) { sk[1][j] = sk[1][j - 1]; sv[1][j] = sv[1][j - 1]; j--; }
                    sk[1][j] = x; sv[1][j] = xv;
                }
                __syncthreads();
                break;                                  // (uniform) sorted in sk[1]: cur == 1
            }
        }
    }
    for (int i = threadIdx.x; i < n; i += ST) { okeys[i] = sk[cur][i]; ovals[i] = sv[cur][i]; }
}

__global__ __launch_bounds__(ST) void k_sort_small(uint64_t *keys, uint32_t *vals, int n, int bits) {
    sort_tile_lds(keys, vals, n, bits, keys, vals);
}

// ---- mid-size inputs (a few tiles .. ~400 K pairs): ONE global pass on the TOP digit, then every digit's
// bucket (<= one tile when the keys are spread) is finished on chip -- 4 launches instead of 16.
// bstart[d] = first position of top digit d (bstart[256] = n); *maxc = the largest bucket
__global__ __launch_bounds__(256) void k_bucket_bounds(const uint32_t *__restrict__ table, int64_t ntiles,
                                                       uint32_t *__restrict__ bstart, uint32_t *__restrict__ maxc,
                                                       volatile uint64_t *mbox, uint64_t seq) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t mx;
    if (threadIdx.x == 0) mx = 0;
    uint32_t c = 0;
    for (int64_t t = 0; t < ntiles; t++) c += table[(int64_t)threadIdx.x * ntiles + t];
    uint32_t tot = 0;
    const uint32_t ex = rfxd::block_exclusive_scan(c, wsum, &tot);
    bstart[threadIdx.x] = ex;
    if (threadIdx.x == 0) bstart[256] = tot;
    __syncthreads();
    atomicMax(&mx, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        *maxc = mx;
        if (mbox) { mbox[1] = mx; __threadfence_system(); mbox[0] = seq; }     // (rfx_ctx::mailbox: the host spins on `seq`)
    }
}

__global__ __launch_bounds__(ST) void k_sort_buckets(const uint64_t *__restrict__ ikeys, const uint32_t *__restrict__ ivals,
                                                     const uint32_t *__restrict__ bstart, int bits,
                                                     uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals) {
    const uint32_t b = bstart[blockIdx.x], e = bstart[blockIdx.x + 1];
    if (e <= b) return;                              // (uniform)
    sort_tile_lds(ikeys + b, ivals + b, (int)(e - b), bits, okeys + b, ovals + b);
}

}  // namespace

namespace rfx {

int sort_pairs(rfx_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int64_t n, int key_bits,
               uint64_t *d_tmp_keys, uint32_t *d_tmp_vals) {
    if (n <= 1) return RFX_OK;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    const int passes = (key_bits + 7) / 8;
    if (n <= STILE) {
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(ST), 0, ctx->stream, d_keys, d_vals, (int)n, key_bits);
        RFX_HIP(hipGetLastError());
        return RFX_OK;
    }
    const int64_t ntiles = ceil_div(n, STILE);
    DevBuf table, offs;
    RFX_HIP(table.alloc((size_t)ntiles * 256 * 4, ctx->stream));
    RFX_HIP(offs.alloc((size_t)(ntiles * 256 + 1) * 8, ctx->stream));
    uint64_t *sk = d_keys, *dk = d_tmp_keys;
    uint32_t *sv = d_vals, *dv = d_tmp_vals;
    const bool inline_scan = ntiles <= 64;           // <= 128 K pairs
    static const bool msd_off = getenv("RFX_SORT_MSD") && atoi(getenv("RFX_SORT_MSD")) == 0;
    const SegMap nomap{nullptr, nullptr, 256};
    if (n > ((int64_t)1 << 26) && n < ((int64_t)1 << 32) && key_bits > 24 && !msd_off) {
        // Beyond 2^26 pairs (the 4.5e8 survivors of a human-scale share): THREE stable MSD levels -- 8 bits, 8 bits, then as
        // many as bring a bucket to ~1000 pairs -- and the final buckets on chip: 85 -> ~25 ms where the eight LSD passes
        // ran at 1.3 TB/s.  The second level's 65536 buckets are the parents of the third (SegMap::NP).  Skewed keys (a
        // final bucket larger than a tile) take the LSD passes below from the regrouped state, as on the two-level path.
        const int shift1 = key_bits - 8, shift2 = key_bits - 16;
        int b3 = 1;
        while (b3 < 8 && (n >> (16 + b3)) > 1000) b3++;
        if (b3 > shift2) b3 = shift2;
        const int shift3 = shift2 - b3, D3 = 1 << b3;
        const int NP2 = 65536;
        const int64_t vmax2 = ntiles + 256, vmax3 = ntiles + NP2;
        DevBuf table2, offs2, maps1, bounds2, tstart2, table3, offs3, bounds3;
        RFX_HIP(table2.alloc((size_t)vmax2 * 256 * 4, ctx->stream));
        RFX_HIP(offs2.alloc((size_t)(vmax2 * 256 + 1) * 8, ctx->stream));
        RFX_HIP(maps1.alloc(2 * 257 * 4, ctx->stream));
        RFX_HIP(bounds2.alloc((size_t)(NP2 + 2) * 4, ctx->stream));
        RFX_HIP(tstart2.alloc((size_t)(NP2 + 1) * 4, ctx->stream));
        RFX_HIP(table3.alloc((size_t)vmax3 * D3 * 4, ctx->stream));
        RFX_HIP(offs3.alloc((size_t)(vmax3 * D3 + 1) * 8, ctx->stream));
        RFX_HIP(bounds3.alloc((size_t)((size_t)NP2 * D3 + 2) * 4, ctx->stream));
        uint32_t *d_b1 = maps1.as<uint32_t>(), *d_t1 = maps1.as<uint32_t>() + 257;
        uint32_t *d_max2 = bounds2.as<uint32_t>() + NP2 + 1, *d_max3 = bounds3.as<uint32_t>() + (size_t)NP2 * D3 + 1;
        // level 1: sk -> dk
        hipLaunchKernelGGL(k_hist<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift1, table.as<uint32_t>(), ntiles, nomap);
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
        hipLaunchKernelGGL(k_scatter<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift1,
                           (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv, (const uint32_t *)table.as<uint32_t>(), nomap);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l2_setup, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)offs.as<uint64_t>(), ntiles, n, d_b1, d_t1);
        RFX_HIP(hipGetLastError());
        // level 2: dk -> sk, 256 digits inside every level-1 bucket
        const SegMap sm2{d_b1, d_t1, 256, 256};
        RFX_HIP(hipMemsetAsync(table2.p, 0, (size_t)vmax2 * 256 * 4, ctx->stream));
        RFX_HIP(hipMemsetAsync(d_max2, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_hist<true>, dim3((unsigned)vmax2), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, n, shift2, table2.as<uint32_t>(), vmax2, sm2);
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table2.as<uint32_t>(), offs2.as<uint64_t>(), vmax2 * 256));
        hipLaunchKernelGGL(k_scatter<true>, dim3((unsigned)vmax2), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, (const uint32_t *)dv, n, shift2,
                           (const uint64_t *)offs2.as<uint64_t>(), vmax2, sk, sv, (const uint32_t *)table2.as<uint32_t>(), sm2);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l3_bounds, dim3((unsigned)ceil_div(NP2 + 1, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)offs2.as<uint64_t>(), sm2, n, bounds2.as<uint32_t>(), d_max2);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_seg_setup, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)bounds2.as<uint32_t>(), NP2, tstart2.as<uint32_t>());
        RFX_HIP(hipGetLastError());
        // level 3: sk -> dk, D3 digits inside every level-2 bucket
        const SegMap sm3{bounds2.as<uint32_t>(), tstart2.as<uint32_t>(), D3, NP2};
        RFX_HIP(hipMemsetAsync(table3.p, 0, (size_t)vmax3 * D3 * 4, ctx->stream));
        RFX_HIP(hipMemsetAsync(d_max3, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_hist<true>, dim3((unsigned)vmax3), dim3(ST), 0, ctx->stream, (const uint64_t *)sk, n, shift3, table3.as<uint32_t>(), vmax3, sm3);
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table3.as<uint32_t>(), offs3.as<uint64_t>(), vmax3 * D3));
        hipLaunchKernelGGL(k_scatter<true>, dim3((unsigned)vmax3), dim3(ST), 0, ctx->stream, (const uint64_t *)sk, (const uint32_t *)sv, n, shift3,
                           (const uint64_t *)offs3.as<uint64_t>(), vmax3, dk, dv, (const uint32_t *)table3.as<uint32_t>(), sm3);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l3_bounds, dim3((unsigned)ceil_div((int64_t)NP2 * D3 + 1, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)offs3.as<uint64_t>(), sm3, n, bounds3.as<uint32_t>(), d_max3);
        RFX_HIP(hipGetLastError());
        uint32_t maxc = 0;
        RFX_TRY(small_readback(ctx, &maxc, d_max3, 4));
        if (maxc <= (uint32_t)STILE) {
            hipLaunchKernelGGL(k_sort_buckets, dim3((unsigned)((int64_t)NP2 * D3)), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, (const uint32_t *)dv,
                               (const uint32_t *)bounds3.as<uint32_t>(), shift3, d_keys, d_vals);
            RFX_HIP(hipGetLastError());
            return RFX_OK;
        }
        // skewed: (dk, dv) hold a stable regrouping of the input; the LSD passes finish it from there
        std::swap(sk, dk);
        std::swap(sv, dv);
    } else if (ntiles > 192 && key_bits > 16 && n <= ((int64_t)1 << 26) && !msd_off) {
        // Large inputs: TWO stable MSD levels (8 bits, then as many as bring a bucket to ~1000 pairs), then every
        // final bucket is finished on chip -- ~90 B of HBM traffic per pair instead of 32 B x (key_bits / 8) LSD passes.
        // A bucket larger than a tile (skewed keys) sends the whole thing down the LSD passes below instead, which
        // give the same result from the partly grouped state (every level is stable).
        const int shift1 = key_bits - 8;
        int b2 = 1;
        while (b2 < 8 && (n >> (8 + b2)) > 1000) b2++;
        if (b2 > shift1) b2 = shift1;
        const int shift2 = shift1 - b2, D2 = 1 << b2;
        const int64_t vmax = ntiles + 256;               // upper bound of the virtual tiles
        DevBuf table2, offs2, maps, b3;
        RFX_HIP(table2.alloc((size_t)vmax * D2 * 4, ctx->stream));
        RFX_HIP(offs2.alloc((size_t)(vmax * D2 + 1) * 8, ctx->stream));
        RFX_HIP(maps.alloc(2 * 257 * 4, ctx->stream));
        RFX_HIP(b3.alloc((size_t)(256 * D2 + 2) * 4, ctx->stream));
        uint32_t *d_bstart = maps.as<uint32_t>(), *d_tstart = maps.as<uint32_t>() + 257;
        uint32_t *d_maxc = b3.as<uint32_t>() + 256 * D2 + 1;
        hipLaunchKernelGGL(k_hist<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift1, table.as<uint32_t>(), ntiles, nomap);
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
        hipLaunchKernelGGL(k_scatter<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift1,
                           (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv, (const uint32_t *)table.as<uint32_t>(), nomap);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l2_setup, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)offs.as<uint64_t>(), ntiles, n, d_bstart, d_tstart);
        RFX_HIP(hipGetLastError());
        const SegMap sm{d_bstart, d_tstart, D2};
        RFX_HIP(hipMemsetAsync(table2.p, 0, (size_t)vmax * D2 * 4, ctx->stream));
        RFX_HIP(hipMemsetAsync(d_maxc, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_hist<true>, dim3((unsigned)vmax), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, n, shift2, table2.as<uint32_t>(), vmax, sm);
        RFX_HIP(hipGetLastError());
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, table2.as<uint32_t>(), offs2.as<uint64_t>(), vmax * D2));
        hipLaunchKernelGGL(k_scatter<true>, dim3((unsigned)vmax), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, (const uint32_t *)dv, n, shift2,
                           (const uint64_t *)offs2.as<uint64_t>(), vmax, sk, sv, (const uint32_t *)table2.as<uint32_t>(), sm);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_l3_bounds, dim3((unsigned)ceil_div(256 * D2 + 1, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)offs2.as<uint64_t>(), sm, n, b3.as<uint32_t>(), d_maxc);
        RFX_HIP(hipGetLastError());
        uint32_t maxc = 0;
        RFX_TRY(small_readback(ctx, &maxc, d_maxc, 4));
        if (maxc <= (uint32_t)STILE) {
            hipLaunchKernelGGL(k_sort_buckets, dim3((unsigned)(256 * D2)), dim3(ST), 0, ctx->stream, (const uint64_t *)sk, (const uint32_t *)sv,
                               (const uint32_t *)b3.as<uint32_t>(), shift2, d_keys, d_vals);
            RFX_HIP(hipGetLastError());
            return RFX_OK;
        }
        // skewed: (sk, sv) = (d_keys, d_vals) hold a stable regrouping of the input; the LSD passes finish it
    }
    if (ntiles <= 192 && key_bits > 8 && !msd_off) {
        // top digit first; the 4-byte readback decides (a bucket larger than a tile -- skewed keys -- takes
        // the LSD passes below instead)
        const int shift = key_bits - 8;
        DevBuf bounds;
        RFX_HIP(bounds.alloc(258 * 4, ctx->stream));
        hipLaunchKernelGGL(k_hist<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift, table.as<uint32_t>(), ntiles, nomap);
        RFX_HIP(hipGetLastError());
        const uint64_t seq = mailbox_next(ctx);
        hipLaunchKernelGGL(k_bucket_bounds, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t *)table.as<uint32_t>(), ntiles,
                           bounds.as<uint32_t>(), bounds.as<uint32_t>() + 257, seq ? ctx->mailbox : (volatile uint64_t *)nullptr, seq);
        RFX_HIP(hipGetLastError());
        uint32_t maxc = 0;
        if (seq) { uint64_t v = 0; RFX_TRY(mailbox_wait(ctx, seq, &v, 1)); maxc = (uint32_t)v; }
        else {
            RFX_HIP(hipMemcpyAsync(&maxc, bounds.as<uint32_t>() + 257, 4, hipMemcpyDeviceToHost, ctx->stream));
            RFX_TRY(sync_checked(ctx));
        }
        if (maxc <= (uint32_t)STILE) {
            if (!inline_scan) RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
            hipLaunchKernelGGL(k_scatter<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift,
                               inline_scan ? (const uint64_t *)nullptr : (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv,
                               (const uint32_t *)table.as<uint32_t>(), nomap);
            RFX_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_sort_buckets, dim3(256), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, (const uint32_t *)dv,
                               (const uint32_t *)bounds.as<uint32_t>(), shift, d_keys, d_vals);
            RFX_HIP(hipGetLastError());
            return RFX_OK;
        }
    }
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_hist<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift,
                           table.as<uint32_t>(), ntiles, nomap);
        RFX_HIP(hipGetLastError());
        if (!inline_scan) RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
        hipLaunchKernelGGL(k_scatter<false>, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift,
                           inline_scan ? (const uint64_t *)nullptr : (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv,
                           (const uint32_t *)table.as<uint32_t>(), nomap);
        RFX_HIP(hipGetLastError());
        uint64_t *tk = sk; sk = dk; dk = tk;
        uint32_t *tv = sv; sv = dv; dv = tv;
    }
    if (sk != d_keys) {
        RFX_HIP(hipMemcpyAsync(d_keys, sk, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d_vals, sv, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return RFX_OK;
}

}  // namespace rfx
