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

// ---- single-pass scan with decoupled look-back: ONE launch whatever n is.  Tile t (taken in ticket order, so every
// predecessor is running or done) publishes its aggregate, walks back over its predecessors' descriptors until it meets
// an inclusive prefix, publishes its own inclusive prefix and writes its outputs.  Descriptor = epoch (20 bits) | status
// (2 bits: 1 aggregate, 2 inclusive prefix) | value (42 bits); DUAL scans two arrays with two descriptors per tile.
// (16 items a thread: with 8 the 4,500 tiles of a 9 M-element scan were a chain of look-backs, 23 ns a tile = 105 us for 110 MB;
// counts -> contigs 15.7 -> 15.6 ms at k = 31, 21.6 -> 21.1 at k = 63; 32 items: no better)
constexpr int LB_THREADS = 256, LB_ITEMS = 16, LB_TILE = LB_THREADS * LB_ITEMS;
constexpr uint64_t LB_VMASK = (1ULL << 42) - 1;
constexpr uint32_t LB_SPIN_LIMIT = 1u << 24;      // seconds of polling
__device__ __forceinline__ uint64_t lb_pack(uint32_t epoch, uint32_t status, uint64_t v) {
    return ((uint64_t)epoch << 44) | ((uint64_t)status << 42) | (v & LB_VMASK);
}

template <bool DUAL>
__global__ __launch_bounds__(LB_THREADS) void k_scan_lookback(const uint32_t *__restrict__ in_a, const uint32_t *__restrict__ in_b,
                                                             int64_t n, uint64_t *__restrict__ out_a, uint64_t *__restrict__ out_b,
                                                             uint64_t *desc, unsigned long long *ticket,
                                                             unsigned long long ticket_base, uint32_t epoch, int *fault) {
    __shared__ uint64_t lds[LB_THREADS / 64];
    __shared__ unsigned long long my_tile;
    __shared__ uint64_t excl[2];
    if (threadIdx.x == 0) my_tile = atomicAdd(ticket, 1ULL) - ticket_base;
    __syncthreads();
    const int64_t tile = (int64_t)my_tile;
    const int64_t base = tile * LB_TILE + (int64_t)threadIdx.x * LB_ITEMS;
    uint32_t va[LB_ITEMS], vb[LB_ITEMS];
    uint64_t sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < LB_ITEMS; i++) {
        const int64_t idx = base + i;
        va[i] = idx < n ? in_a[idx] : 0u; sa += va[i];
        if (DUAL) { vb[i] = idx < n ? in_b[idx] : 0u; sb += vb[i]; }
    }
    uint64_t ta, tb = 0;
    const uint64_t ea = block_excl_scan_u64(sa, lds, &ta);
    uint64_t eb = 0;
    if (DUAL) eb = block_excl_scan_u64(sb, lds, &tb);
    constexpr int ND = DUAL ? 2 : 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < ND) {
        // one wave per scanned array looks back over 64 predecessors at a time.  A descriptor carries its value in the
        // same 64-bit word as its status, so relaxed atomics are enough: nothing else is published through it.
        const int which = wave;
        const uint64_t agg = which ? tb : ta;
        uint64_t *d = desc + (size_t)tile * ND + which;
        uint64_t prefix = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(d, lb_pack(epoch, 1u, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t p0 = tile - 1;
            for (;;) {
                const int64_t p = p0 - lane;
                uint64_t x = lb_pack(epoch, 2u, 0);             // before tile 0: an inclusive prefix of nothing
                if (p >= 0) {
                    uint32_t spins = 0;
                    for (;;) {
                        x = __hip_atomic_load(desc + (size_t)p * ND + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(x >> 44) == epoch && ((x >> 42) & 3u) != 0u) break;
                        if (++spins >= LB_SPIN_LIMIT) {         // never expected: leave instead of hanging, the host reports it
                            __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            x = lb_pack(epoch, 2u, 0);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const unsigned long long pm = __ballot(((x >> 42) & 3u) == 2u);
                const int first = pm ? __ffsll((long long)pm) - 1 : 64;   // nearest predecessor holding a prefix
                uint64_t v = lane <= first ? (x & LB_VMASK) : 0;
#pragma unroll
                for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
                prefix += v;
                if (pm) break;
                p0 -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(d, lb_pack(epoch, 2u, prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            excl[which] = prefix;
        }
    }
    __syncthreads();
    uint64_t xa = ea + excl[0], xb = DUAL ? eb + excl[1] : 0;
#pragma unroll
    for (int i = 0; i < LB_ITEMS; i++) {
        const int64_t idx = base + i;
        if (idx < n) { out_a[idx] = xa; if (DUAL) out_b[idx] = xb; }
        xa += va[i]; if (DUAL) xb += vb[i];
        if (idx == n - 1) { out_a[n] = xa; if (DUAL) out_b[n] = xb; }
    }
}

__global__ void k_mailbox_post(const uint8_t *__restrict__ src, int nbytes, volatile uint64_t *mbox, uint64_t seq) {
    volatile uint8_t *dst = (volatile uint8_t *)(mbox + 1);
    if ((int)threadIdx.x < nbytes) dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); mbox[0] = seq; }
}

int scan_lookback(rfx_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t *oa, uint64_t *ob, int64_t n) {
    const int64_t nt = ceil_div(n, LB_TILE);
    const size_t need = (size_t)nt * 2;
    if (ctx->scan_fault && *ctx->scan_fault) { ctx->last_error = "scan: look-back gave up waiting for a predecessor tile"; return RFX_E_HIP; }
    if (ctx->scan_desc_cap < need || !ctx->scan_ticket) {
        RFX_TRY(sync_checked(ctx));
        if (ctx->scan_desc) (void)hipFree(ctx->scan_desc);
        const size_t cap = need + need / 4 + 4096;
        RFX_HIP(hipMalloc((void **)&ctx->scan_desc, cap * 8));
        RFX_HIP(hipMemsetAsync(ctx->scan_desc, 0, cap * 8, ctx->stream));   // stream-ordered: the context's stream does not
                                                                              // wait for the null stream
        ctx->scan_desc_cap = cap;
        ctx->scan_epoch = 0;
        if (!ctx->scan_ticket) {
            RFX_HIP(hipHostMalloc((void **)&ctx->scan_fault, sizeof(int), hipHostMallocMapped));
            *ctx->scan_fault = 0;
            RFX_HIP(hipMalloc((void **)&ctx->scan_ticket, 8));
            RFX_HIP(hipMemsetAsync(ctx->scan_ticket, 0, 8, ctx->stream));
            ctx->scan_tickets_issued = 0;
        }
    }
    if (++ctx->scan_epoch >= (1u << 20)) {             // epochs wrapped: forget every old descriptor
        RFX_HIP(hipMemsetAsync(ctx->scan_desc, 0, ctx->scan_desc_cap * 8, ctx->stream));
        ctx->scan_epoch = 1;
    }
    const unsigned long long tb = ctx->scan_tickets_issued;
    if (b)
        hipLaunchKernelGGL(k_scan_lookback<true>, dim3((unsigned)nt), dim3(LB_THREADS), 0, ctx->stream, a, b, n, oa, ob, ctx->scan_desc,
                           ctx->scan_ticket, tb, ctx->scan_epoch, ctx->scan_fault);
    else
        hipLaunchKernelGGL(k_scan_lookback<false>, dim3((unsigned)nt), dim3(LB_THREADS), 0, ctx->stream, a, (const uint32_t *)nullptr, n, oa,
                           (uint64_t *)nullptr, ctx->scan_desc, ctx->scan_ticket, tb, ctx->scan_epoch, ctx->scan_fault);
    RFX_HIP(hipGetLastError());
    ctx->scan_tickets_issued += (unsigned long long)nt;        // only once the launch is known to be queued: the device counter moves with it
    return RFX_OK;
}

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

int small_readback(rfx_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes) {
    const uint64_t seq = nbytes <= 56 ? mailbox_next(ctx) : 0;
    if (!seq) {
        RFX_HIP(hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
        return sync_checked(ctx);
    }
    hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(64), 0, ctx->stream, (const uint8_t *)d_src, (int)nbytes, ctx->mailbox, seq);
    RFX_HIP(hipGetLastError());
    uint64_t v[7] = {0, 0, 0, 0, 0, 0, 0};
    RFX_TRY(mailbox_wait(ctx, seq, v, (int)((nbytes + 7) / 8)));
    memcpy(h_dst, v, nbytes);
    return RFX_OK;
}

namespace rfx {

int exclusive_scan_u64(rfx_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int64_t n) {
    return scan_impl<uint64_t>(ctx, d_in, d_out, n);
}
int exclusive_scan_u32_to_u64(rfx_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, int64_t n) {
    static const bool off = getenv("RFX_SCAN_LOOKBACK") && atoi(getenv("RFX_SCAN_LOOKBACK")) == 0;
    if (n > SCAN_TILE && !off) return scan_lookback(ctx, d_in, nullptr, d_out, nullptr, n);
    return scan_impl<uint32_t>(ctx, d_in, d_out, n);
}
int exclusive_scan2_u32_to_u64(rfx_ctx *ctx, const uint32_t *d_in_a, const uint32_t *d_in_b, uint64_t *d_out_a,
                               uint64_t *d_out_b, int64_t n) {
    static const bool off = getenv("RFX_SCAN_LOOKBACK") && atoi(getenv("RFX_SCAN_LOOKBACK")) == 0;
    if (n > 0 && !off) return scan_lookback(ctx, d_in_a, d_in_b, d_out_a, d_out_b, n);
    RFX_TRY(scan_impl<uint32_t>(ctx, d_in_a, d_out_a, n));
    return scan_impl<uint32_t>(ctx, d_in_b, d_out_b, n);
}

}  // namespace rfx
