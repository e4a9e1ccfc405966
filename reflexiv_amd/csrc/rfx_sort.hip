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

__global__ __launch_bounds__(ST) void k_hist(const uint64_t *__restrict__ keys, int64_t n, int shift,
                                             uint32_t *__restrict__ table, int64_t ntiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * STILE;
#pragma unroll
    for (int i = 0; i < SI; i++) {
        int64_t idx = base + (int64_t)i * ST + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255], 1u);
    }
    __syncthreads();
    table[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(ST) void k_scatter(const uint64_t *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, int64_t n, int shift,
                                                const uint64_t *__restrict__ offs, int64_t ntiles,
                                                uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals,
                                                const uint32_t *__restrict__ table) {
    __shared__ volatile uint32_t wcnt[SW][256];
    __shared__ uint64_t wbase[SW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < SW * 256; i += ST) ((volatile uint32_t *)wcnt)[i] = 0;
    __syncthreads();

    // wave `wave` owns the contiguous chunk [wave*512, wave*512+512) of the tile, read in
    // 8 rounds of 64 consecutive keys: arrival order = (round, lane)
    const int64_t cbase = (int64_t)blockIdx.x * STILE + (int64_t)wave * (64 * SI);
    uint64_t key[SI];
    uint32_t val[SI], rank[SI];
    const uint64_t lt = (1ULL << lane) - 1;
#pragma unroll
    for (int r = 0; r < SI; r++) {
        int64_t idx = cbase + r * 64 + lane;
        bool ok = idx < n;
        key[r] = ok ? keys[idx] : 0;
        val[r] = ok ? vals[idx] : 0;
        unsigned d = (unsigned)(key[r] >> shift) & 255u;
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
        if (offs) {
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
        if (idx < n) {
            unsigned d = (unsigned)(key[r] >> shift) & 255u;
            uint64_t dst = wbase[wave][d] + rank[r];
            okeys[dst] = key[r];
            ovals[dst] = val[r];
        }
    }
}

// One workgroup sorts up to STILE pairs entirely on chip (all passes), stable: (ikeys, ivals)[0..n) ->
// (okeys, ovals)[0..n), in place when they are the same arrays.
__device__ __forceinline__ void sort_tile_lds(const uint64_t *keys, const uint32_t *vals, int n, int passes,
                                              uint64_t *okeys, uint32_t *ovals) {
    __shared__ uint64_t sk[2][STILE];
    __shared__ uint32_t sv[2][STILE];
    __shared__ volatile uint32_t wcnt[SW][256];
    __shared__ uint32_t wbase[SW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ULL << lane) - 1;
    for (int i = threadIdx.x; i < n; i += ST) { sk[0][i] = keys[i]; sv[0][i] = vals[i]; }
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        for (int i = threadIdx.x; i < SW * 256; i += ST) ((volatile uint32_t *)wcnt)[i] = 0;
        __syncthreads();
        uint32_t rank[SI];
        const int cbase = wave * (64 * SI);
#pragma unroll
        for (int r = 0; r < SI; r++) {
            int idx = cbase + r * 64 + lane;
            bool ok = idx < n;
            unsigned d = ok ? (unsigned)(sk[cur][idx] >> shift) & 255u : 0u;
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
        __syncthreads();
        // exclusive scan over (digit, wave): thread d sums its digit, then a block scan over digits
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < SW; w++) tot += wcnt[w][threadIdx.x];
        __shared__ uint32_t wsum[SW];
        uint32_t dbase = rfxd::block_exclusive_scan(tot, wsum, nullptr);
#pragma unroll
        for (int w = 0; w < SW; w++) { wbase[w][threadIdx.x] = dbase; dbase += wcnt[w][threadIdx.x]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SI; r++) {
            int idx = cbase + r * 64 + lane;
            if (idx < n) {
                uint64_t kk = sk[cur][idx];
                unsigned d = (unsigned)(kk >> shift) & 255u;
                uint32_t dst = wbase[wave][d] + rank[r];
                sk[cur ^ 1][dst] = kk;
                sv[cur ^ 1][dst] = sv[cur][idx];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = threadIdx.x; i < n; i += ST) { okeys[i] = sk[cur][i]; ovals[i] = sv[cur][i]; }
}

__global__ __launch_bounds__(ST) void k_sort_small(uint64_t *keys, uint32_t *vals, int n, int passes) {
    sort_tile_lds(keys, vals, n, passes, keys, vals);
}

// ---- mid-size inputs (a few tiles .. ~400 K pairs): ONE global pass on the TOP digit, then every digit's
// bucket (<= one tile when the keys are spread) is finished on chip -- 4 launches instead of 16.
// bstart[d] = first position of top digit d (bstart[256] = n); *maxc = the largest bucket
__global__ __launch_bounds__(256) void k_bucket_bounds(const uint32_t *__restrict__ table, int64_t ntiles,
                                                       uint32_t *__restrict__ bstart, uint32_t *__restrict__ maxc) {
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
    if (threadIdx.x == 0) *maxc = mx;
}

__global__ __launch_bounds__(ST) void k_sort_buckets(const uint64_t *__restrict__ ikeys, const uint32_t *__restrict__ ivals,
                                                     const uint32_t *__restrict__ bstart, int passes,
                                                     uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals) {
    const uint32_t b = bstart[blockIdx.x], e = bstart[blockIdx.x + 1];
    if (e <= b) return;                              // (uniform)
    sort_tile_lds(ikeys + b, ivals + b, (int)(e - b), passes, okeys + b, ovals + b);
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
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(ST), 0, ctx->stream, d_keys, d_vals, (int)n, passes);
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
    if (ntiles <= 192 && key_bits > 8 && !msd_off) {
        // top digit first; the 4-byte readback decides (a bucket larger than a tile -- skewed keys -- takes
        // the LSD passes below instead)
        const int shift = key_bits - 8;
        DevBuf bounds;
        RFX_HIP(bounds.alloc(258 * 4, ctx->stream));
        hipLaunchKernelGGL(k_hist, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift, table.as<uint32_t>(), ntiles);
        RFX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_bucket_bounds, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t *)table.as<uint32_t>(), ntiles,
                           bounds.as<uint32_t>(), bounds.as<uint32_t>() + 257);
        RFX_HIP(hipGetLastError());
        uint32_t maxc = 0;
        RFX_HIP(hipMemcpyAsync(&maxc, bounds.as<uint32_t>() + 257, 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipStreamSynchronize(ctx->stream));
        if (maxc <= (uint32_t)STILE) {
            if (!inline_scan) RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
            hipLaunchKernelGGL(k_scatter, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift,
                               inline_scan ? (const uint64_t *)nullptr : (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv,
                               (const uint32_t *)table.as<uint32_t>());
            RFX_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_sort_buckets, dim3(256), dim3(ST), 0, ctx->stream, (const uint64_t *)dk, (const uint32_t *)dv,
                               (const uint32_t *)bounds.as<uint32_t>(), (shift + 7) / 8, d_keys, d_vals);
            RFX_HIP(hipGetLastError());
            return RFX_OK;
        }
    }
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_hist, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, n, shift,
                           table.as<uint32_t>(), ntiles);
        RFX_HIP(hipGetLastError());
        if (!inline_scan) RFX_TRY(exclusive_scan_u32_to_u64(ctx, table.as<uint32_t>(), offs.as<uint64_t>(), ntiles * 256));
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)ntiles), dim3(ST), 0, ctx->stream, sk, sv, n, shift,
                           inline_scan ? (const uint64_t *)nullptr : (const uint64_t *)offs.as<uint64_t>(), ntiles, dk, dv,
                           (const uint32_t *)table.as<uint32_t>());
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
