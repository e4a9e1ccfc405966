// rfx_scan.hip -- device exclusive prefix sums (uint64 results), used for bucket offsets,
// emission indices and extension-word offsets.  out has n+1 entries; out[n] = total.
#include "rfx_internal.h"
#include "rfx_device.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *lds, uint64_t *total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        uint64_t s = lds[i];
        if (i < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// per-tile sums
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_sums(const T *__restrict__ in, int64_t n,
                                                            uint64_t *__restrict__ sums) {
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + (int64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += (uint64_t)in[idx];
    }
    uint64_t tot;
    block_excl_scan_u64(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// scan each tile, adding the tile's scanned offset; the thread owning element n-1 also
// writes out[n].  offs == nullptr means a single tile with offset 0.
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_scan(const T *__restrict__ in, int64_t n,
                                                            const uint64_t *__restrict__ offs,
                                                            uint64_t *__restrict__ out) {
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + i;
        v[i] = idx < n ? (uint64_t)in[idx] : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(s, lds, &tot) + (offs ? offs[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + i;
        if (idx < n) out[idx] = ex;
        ex += v[i];
        if (idx == n - 1) out[n] = ex;
    }
}

__global__ void k_zero_total(uint64_t *out) { out[0] = 0; }

template <class T>
int scan_impl(rfx_ctx *ctx, const T *d_in, uint64_t *d_out, int64_t n) {
    if (n <= 0) {
        hipLaunchKernelGGL(k_zero_total, dim3(1), dim3(1), 0, ctx->stream, d_out);
        RFX_HIP(hipGetLastError());
        return RFX_OK;
    }
    int64_t nt = ceil_div(n, SCAN_TILE);
    if (nt == 1) {
        hipLaunchKernelGGL(k_tile_scan<T>, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n,
                           (const uint64_t *)nullptr, d_out);
        RFX_HIP(hipGetLastError());
        return RFX_OK;
    }
    DevBuf sums, offs;
    RFX_HIP(sums.alloc((size_t)nt * 8, ctx->stream));
    RFX_HIP(offs.alloc((size_t)(nt + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_tile_sums<T>, dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n,
                       sums.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(scan_impl<uint64_t>(ctx, sums.as<uint64_t>(), offs.as<uint64_t>(), nt));
    hipLaunchKernelGGL(k_tile_scan<T>, dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n,
                       (const uint64_t *)offs.as<uint64_t>(), d_out);
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

}  // namespace

namespace rfx {

int exclusive_scan_u64(rfx_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int64_t n) {
    return scan_impl<uint64_t>(ctx, d_in, d_out, n);
}
int exclusive_scan_u32_to_u64(rfx_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, int64_t n) {
    return scan_impl<uint32_t>(ctx, d_in, d_out, n);
}

}  // namespace rfx
