// rfx_shard.hip -- the extend stage on several GPUs, behind the C ABI: the range shuffle of `sortByKey`.
//
// The reference sorts the whole record set by key before each fork filter and before each extend pass
// (P/ReflexivMain.java:179,191,211,235,247,286; DS P/ReflexivDSMain.java:232,244,261,272,276,322;
// k > 31: P/ReflexivDSMain64.java:504,516,533,545,549,563): a range-partitioned shuffle of whole records.  Here:
//
//   sort_exchange   local stable sort -> W-1 rank splitters picked ON THE DEVICE from ONE all-gather of a regular sample
//                   of every rank's sorted keys (equal keys never straddle ranks: the cut is an upper bound) -> one count
//                   matrix (all-gather; the only host wait of the shuffle) -> one all-to-all(v) of whole records (fixed
//                   fields packed (KW + 2) words a record, extension words as they lie) -> local stable sort of what
//                   arrived, source rank after source rank, which IS the global stable order because ranks hold
//                   consecutive key ranges and the global arrival order is rank after rank.
//   partition plan  the order contract's P logical partitions (start p = floor(p N / P) moved forward past equal keys,
//                   DESIGN.md section 2) are cut out of the GLOBAL sorted sequence, whatever the number of ranks: a rank
//                   computes the starts that fall into its own range from N and its global offset alone.  A partition
//                   may therefore begin on one rank and go on on the next (any P with any W: P = 4 on 8 GPUs, P = 8 on 3).
//   carry           what such a partition brings from the earlier ranks is the PARITY of a count -- records (random
//                   reflection: orientation 2,1,2,1 in arrival order) or emissions (extend pass: the toggling
//                   randomReflexivMarker) -- SURVEY.md 2.4 C7: one all-gather of four numbers per rank, consumed on the
//                   device between the pass's scan and its emission; no host wait.
//   count(), trace  one all-reduce per pass (it also carries every rank's status: a rank that fails says so at the next
//                   collective instead of leaving its peers inside one).
//
// While the set is large it stays sharded; once it has shrunk to `gather_below` records (or before the k > 31 extras of
// P/ReflexivDSMain64.java:584-619, which want the whole set) it is gathered on rank 0, rank after rank, and the one-GPU
// driver (rfx_api.hip assemble_impl) takes the loop up from the same variables.  Same P => same contigs as one GPU.
#include "rfx_comm.h"
#include "rfx_device.h"

using namespace rfxd;
using namespace rfx;

namespace {

constexpr int SH_MAXW = 64;                 // ranks (rfx_comm_init's limit)

inline unsigned grid_for(int64_t n, int block = 256) { return (unsigned)ceil_div(n > 0 ? n : 1, block); }

#define RFX_KW_SWITCH(kw, ...)                                                \
    switch (kw) {                                                             \
    case 1: { constexpr int KW = 1; __VA_ARGS__; } break;                     \
    case 2: { constexpr int KW = 2; __VA_ARGS__; } break;                     \
    case 3: { constexpr int KW = 3; __VA_ARGS__; } break;                     \
    case 4: { constexpr int KW = 4; __VA_ARGS__; } break;                     \
    default: return RFX_E_ARG;                                                \
    }

// the order of the stable sort (rfx_graph.hip sort_records): word after word, unsigned
template <int KW> __device__ __forceinline__ bool key_le(const uint64_t *a, const uint64_t *b) {
#pragma unroll
    for (int i = 0; i < KW; i++) {
        if (a[i] < b[i]) return true;
        if (a[i] > b[i]) return false;
    }
    return true;
}

// sample i of S = the key at position (2 i + 1) n / (2 S) of the locally sorted keys; row = S keys, then n
template <int KW>
__global__ void k_sh_sample(const KeyW<KW> *__restrict__ skey, int64_t n, int S, uint64_t *__restrict__ row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == S) { row[(size_t)S * KW] = (uint64_t)n; return; }
    if (i > S) return;
    KeyW<KW> v;
#pragma unroll
    for (int w = 0; w < KW; w++) v.w[w] = ~0ULL;
    if (n > 0) {
        int64_t p = (int64_t)(((unsigned __int128)(2 * (uint64_t)i + 1) * (uint64_t)n) / (2 * (uint64_t)S));
        if (p >= n) p = n - 1;
        v = skey[p];
    }
#pragma unroll
    for (int w = 0; w < KW; w++) row[(size_t)i * KW + w] = v.w[w];
}

// Splitter t (t = 0 .. W-2) = the smallest sample key whose weighted rank -- a sample of rank r stands for n_r / S records:
// weight n_r -- reaches (t + 1) / W of the whole, S * N.  The weighted rank rises strictly with the key over the samples of
// non-empty ranks (a sample counts itself), so "smallest key" is "smallest weighted rank that reaches the target": one
// 64-bit atomicMin per (candidate, target) in LDS.  One workgroup; every rank runs it on the same gathered rows and gets
// the same keys.  Rank t then owns the keys in (split[t-1], split[t]].
template <int KW>
__global__ void __launch_bounds__(1024) k_sh_splitters(const uint64_t *__restrict__ rows, int W, int S, uint64_t *__restrict__ split) {
    const int RW = S * KW + 1;
    __shared__ unsigned long long best[SH_MAXW];
    __shared__ unsigned long long nr[SH_MAXW];
    __shared__ unsigned long long total;
    const int tid = threadIdx.x;
    if (tid < W) { best[tid] = ~0ULL; nr[tid] = rows[(size_t)tid * RW + (size_t)S * KW]; }
    __syncthreads();
    if (tid == 0) { unsigned long long t = 0; for (int r = 0; r < W; r++) t += nr[r]; total = t; }
    __syncthreads();
    const unsigned long long N = total;
    if (N > 0) {
        for (int c = tid; c < W * S; c += blockDim.x) {
            const int rc = c / S;
            if (nr[rc] == 0) continue;
            const uint64_t *kc = rows + (size_t)rc * RW + (size_t)(c - rc * S) * KW;
            unsigned long long cw = 0;
            for (int r = 0; r < W; r++) {
                if (nr[r] == 0) continue;
                const uint64_t *row = rows + (size_t)r * RW;
                int lo = 0, hi = S;                       // samples of r with key <= kc: [0, lo)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (key_le<KW>(row + (size_t)mid * KW, kc)) lo = mid + 1; else hi = mid;
                }
                cw += nr[r] * (unsigned long long)lo;
            }
            // targets this candidate reaches: (t + 1) * S * N / W <= cw
            const unsigned long long packed = (cw << 16) | (unsigned long long)c;      // (cw <= S * N < 2^48, c < 2^16)
            for (int t = 0; t < W - 1; t++) {
                const unsigned __int128 target = ((unsigned __int128)(t + 1) * (unsigned long long)S * N + (unsigned)W - 1) / (unsigned)W;
                if ((unsigned __int128)cw >= target) atomicMin(&best[t], packed); else break;
            }
        }
    }
    __syncthreads();
    if (tid < W - 1) {
        const unsigned long long b = best[tid];
        for (int w = 0; w < KW; w++) {
            uint64_t v = ~0ULL;
            if (b != ~0ULL) {
                const int c = (int)(b & 0xFFFF), rc = c / S;
                v = rows[(size_t)rc * RW + (size_t)(c - rc * S) * KW + w];
            }
            split[(size_t)tid * KW + w] = v;
        }
    }
}

// cut[t] = records of the locally sorted set that go to ranks < t (upper bound of splitter t - 1), and the matrix row:
// [W] records to each rank, [W] extension words to each rank.  aux = {cut[W + 1], wcut[W + 1]}.  One workgroup.
template <int KW>
__global__ void k_sh_cuts(const KeyW<KW> *__restrict__ skey, int64_t n, const uint64_t *__restrict__ split, int W,
                          const int64_t *__restrict__ ext_off, int64_t *__restrict__ row, int64_t *__restrict__ aux) {
    __shared__ long long cut[SH_MAXW + 1];
    const int t = threadIdx.x;
    if (t <= W) {
        int64_t c;
        if (t == 0) c = 0;
        else if (t == W) c = n;
        else {
            const uint64_t *s = split + (size_t)(t - 1) * KW;
            int64_t lo = 0, hi = n;                        // keys <= s: [0, lo)
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (key_le<KW>(skey[mid].w, s)) lo = mid + 1; else hi = mid;
            }
            c = lo;
        }
        cut[t] = c;
    }
    __syncthreads();
    if (t <= W) { aux[t] = cut[t]; aux[W + 1 + t] = n > 0 ? ext_off[cut[t]] : 0; }
    if (t < W) {
        row[t] = cut[t + 1] - cut[t];
        row[W + t] = n > 0 ? ext_off[cut[t + 1]] - ext_off[cut[t]] : 0;
    }
}

// fixed fields of a record as (KW + 2) words: key, marker | left << 32, right | extension words << 32
template <int KW>
__global__ void k_sh_pack(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker, const int32_t *__restrict__ left,
                          const int32_t *__restrict__ right, const int64_t *__restrict__ ext_off, int64_t n,
                          uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t *o = out + (size_t)i * (KW + 2);
    const KeyW<KW> kk = key[i];
#pragma unroll
    for (int w = 0; w < KW; w++) o[w] = kk.w[w];
    o[KW] = (uint64_t)(uint32_t)marker[i] | ((uint64_t)(uint32_t)left[i] << 32);
    o[KW + 1] = (uint64_t)(uint32_t)right[i] | ((uint64_t)(uint32_t)(ext_off[i + 1] - ext_off[i]) << 32);
}
template <int KW>
__global__ void k_sh_unpack(const uint64_t *__restrict__ in, int64_t n, KeyW<KW> *__restrict__ key, int32_t *__restrict__ marker,
                            int32_t *__restrict__ left, int32_t *__restrict__ right, uint32_t *__restrict__ nw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *o = in + (size_t)i * (KW + 2);
    KeyW<KW> kk;
#pragma unroll
    for (int w = 0; w < KW; w++) kk.w[w] = o[w];
    key[i] = kk;
    marker[i] = (int32_t)(uint32_t)o[KW]; left[i] = (int32_t)(uint32_t)(o[KW] >> 32);
    right[i] = (int32_t)(uint32_t)o[KW + 1]; nw[i] = (uint32_t)(o[KW + 1] >> 32);
}

// The order contract's partition starts (rfx_graph.hip k_partition_starts) on a rank's share [off, off + n) of the
// global sorted sequence of N records: start p = floor(p N / P), moved forward past equal keys -- which never leave the
// rank, so the move is local; a start before this share is 0 here, one behind it n.
template <int KW>
__global__ void k_sh_part_starts(const KeyW<KW> *__restrict__ skey, int64_t n, int64_t N, int64_t off, int P, int64_t *__restrict__ start) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > P) return;
    if (p == P) { start[P] = n; return; }
    const int64_t sg = (int64_t)(((unsigned __int128)(uint64_t)p * (uint64_t)N) / (uint64_t)P);
    int64_t s = sg - off;
    if (s <= 0) s = 0;
    else if (s >= n) s = n;
    else while (s < n && key_eq(skey[s], skey[s - 1])) s++;
    start[p] = s;
}

__device__ __forceinline__ int sh_part_of(const int64_t *__restrict__ ps, int P, int64_t i) {
    int lo = 0, hi = P;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ps[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
// row = {records here, the partition of the first, the partition of the last, count inside that last partition here}
__global__ void k_sh_trailing(const int64_t *__restrict__ ps, int P, const uint64_t *__restrict__ cum, int64_t n, int64_t *__restrict__ row) {
    if (blockIdx.x || threadIdx.x) return;
    if (n <= 0) { row[0] = row[1] = row[2] = row[3] = 0; return; }
    const int pf = sh_part_of(ps, P, 0), pl = sh_part_of(ps, P, n - 1);
    const int64_t base = ps[pl];
    row[0] = n; row[1] = pf; row[2] = pl;
    row[3] = cum ? (int64_t)(cum[n] - cum[base]) : n - base;
}
// carry = {this rank's first partition, parity of what that partition counted on the ranks before}
__global__ void k_sh_carry(const int64_t *__restrict__ all, int W, int me, int32_t *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    if (all[(size_t)me * 4] == 0) { out[0] = -1; out[1] = 0; return; }
    const int64_t p = all[(size_t)me * 4 + 1];
    int par = 0;
    for (int r = me - 1; r >= 0; r--) {
        const int64_t *a = all + (size_t)r * 4;
        if (a[0] == 0) continue;                      // an empty rank is transparent
        if (a[2] != p) break;                         // its last record belongs to an earlier partition: p begins here
        par ^= (int)(a[3] & 1);
        if (a[1] != p) break;                         // p began inside that rank
    }
    out[0] = (int32_t)p; out[1] = par;
}

__global__ void k_sh_iota(int64_t *__restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) v[i] = i;
}

// ------------------------------------------------------------------------------------------------------------------ host
struct Shard : PartCarry {
    rfx_ctx *ctx;
    rfx_comm *c;
    int W, me;
    int st_local = RFX_OK;                // this rank's own failure, carried to the next collective (never returned between two)
    // device scratch of the communicator (grow-only): rows, samples, splitters, cuts, carry
    int64_t *d = nullptr, *h = nullptr;
    int S = 256;
    Arena arena[2];
    int turn = 0;
    // statistics of the call (RFX_TRACE)
    int64_t exchanged_records = 0, exchanged_words = 0;
    int n_exchanges = 0;

    // layout of the scratch (8-byte words)
    size_t o_row() const { return 0; }                                   // this rank's matrix row: 2 W + 4
    size_t o_all() const { return 256; }                                 // everybody's: W * (2 W + 4)  (<= 64 * 132)
    size_t o_aux() const { return o_all() + (size_t)SH_MAXW * (2 * SH_MAXW + 4); }     // cut, wcut: 2 (W + 1)
    size_t o_trail() const { return o_aux() + 2 * (SH_MAXW + 1) + 2; }   // 4
    size_t o_trall() const { return o_trail() + 4; }                     // 4 W
    size_t o_carry() const { return o_trall() + 4 * SH_MAXW; }           // int32[2] per slot, 4 slots
    size_t o_split() const { return o_carry() + 4; }                     // (W - 1) * KW
    size_t o_samp() const { return o_split() + (size_t)SH_MAXW * MAX_KEY_WORDS; }       // this rank's samples: S * KW + 1
    size_t o_sall() const { return o_samp() + (size_t)256 * MAX_KEY_WORDS + 8; }         // everybody's
    size_t words() const { return o_sall() + (size_t)SH_MAXW * (256 * MAX_KEY_WORDS + 8); }
    int carry_slot = 0;

    int init(size_t arena_bytes) {
        W = c->world; me = c->rank;
        S = W <= 16 ? 256 : 4096 / W;                                   // W * S candidates <= 4096
        if (!c->d_sh) {
            RFX_HIP(hipMalloc((void **)&c->d_sh, words() * 8));
            RFX_HIP(hipHostMalloc((void **)&c->h_sh, words() * 8, hipHostMallocDefault));
        }
        d = c->d_sh; h = c->h_sh;
        // the two arenas take at most a third of the free HBM each (what a pass does not find in them comes from the
        // stream-ordered allocator: slower, never wrong) -- the capacity-stress configuration must start, not fail here
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t held = ctx->ws[2].bytes + ctx->ws[3].bytes;
                arena_bytes = std::min(arena_bytes, (free_b + held) / 3);
            }
        }
        for (int i = 0; i < 2; i++) {
            arena[i].base = (char *)ctx->ws_get(2 + i, arena_bytes);
            if (!arena[i].base) { ctx->last_error = "rfx_dev_sharded_assemble: workspace allocation failed"; return RFX_E_HIP; }
            arena[i].cap = arena_bytes;
        }
        return RFX_OK;
    }
    void next_arena() { Arena *a = &arena[turn++ & 1]; a->off = 0; tl_arena = a; }

    // ---- collective: RFX_OK only if every rank says so (a tiny all-reduce; used where no matrix travels anyway)
    int agree(const char *what) {
        int64_t bad[1] = {st_local != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, bad, 1, 1));
        if (st_local != RFX_OK) return st_local;
        if (bad[0]) { ctx->last_error = std::string("rfx_dev_sharded_assemble: a peer failed (") + what + "); see its rfx_last_error()"; return RFX_E_STATE; }
        return RFX_OK;
    }

    // ---- PartCarry: the parity this rank's first partition brings from the ranks before (device -> device, no host wait)
    int compute(rfx_ctx *, const int64_t *d_ps, int P, const uint64_t *d_cum, int64_t n, const int32_t **d_carry) override {
        int32_t *out = (int32_t *)(d + o_carry()) + 2 * (carry_slot++ & 3);
        hipLaunchKernelGGL(k_sh_trailing, dim3(1), dim3(64), 0, ctx->stream, d_ps, P, d_cum, n, d + o_trail());
        RFX_HIP(hipGetLastError());
        RFX_NCCL(nccl().AllGather(d + o_trail(), d + o_trall(), 4, ncclInt64, c->comm, ctx->stream));
        hipLaunchKernelGGL(k_sh_carry, dim3(1), dim3(64), 0, ctx->stream, (const int64_t *)(d + o_trall()), W, me, out);
        RFX_HIP(hipGetLastError());
        *d_carry = out;
        return RFX_OK;
    }

    // the send buffer for n records of kw key words; called BEFORE the count matrix goes round, so that a failure travels in it
    void room_to_send(int64_t n, int kw) {
        if (st_local != RFX_OK) return;
        const int sg = comm_grow(ctx, &c->send, &c->send_bytes, (size_t)std::max<int64_t>(1, n * (kw + 2)) * 8, ctx->stream, c->xs);
        if (sg != RFX_OK) st_local = sg;
    }

    // ---- the all-to-all(v) of whole records.  `src`: records contiguous by destination; d_row (device, at o_row()): [W]
    // records and [W] extension words for each destination; cut / wcut (device, at o_aux()): their starts.  -> `dst`: what
    // arrived, source rank after source rank; *N = records on all ranks, *off = those on the ranks before this one.
    // Collective; a rank's own earlier failure (st_local) travels in the matrix and ends the call on every rank.
    int exchange(const DevRecords &src, int kw, DevRecords &dst, int64_t *N, int64_t *off) {
        const int RW = 2 * W + 4, fw = kw + 2;
        NcclApi &n = nccl();
        // row: [W] records, [W] words, status, what the receive buffer holds (8-byte words)
        h[o_row() + 2 * W] = st_local; h[o_row() + 2 * W + 1] = (int64_t)(c->recv_bytes / 8);     // (pinned: the copy is queued)
        RFX_HIP(hipMemcpyAsync(d + o_row() + 2 * W, h + o_row() + 2 * W, 16, hipMemcpyHostToDevice, ctx->stream));
        RFX_NCCL(n.AllGather(d + o_row(), d + o_all(), (size_t)RW, ncclInt64, c->comm, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h + o_all(), d + o_all(), (size_t)RW * W * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h + o_aux(), d + o_aux(), (size_t)(2 * W + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipStreamSynchronize(ctx->stream));                       // the one host wait of the shuffle
        note_foreign_hip_error(ctx, "rfx_dev_sharded_assemble (count matrix)");
        const int64_t *all = h + o_all(), *cut = h + o_aux(), *wcut = cut + W + 1;
        for (int r = 0; r < W; r++)
            if (all[(size_t)r * RW + 2 * W] != RFX_OK) {
                if (st_local != RFX_OK) return st_local;
                char buf[160];
                snprintf(buf, sizeof buf, "rfx_dev_sharded_assemble: rank %d failed (status %lld); see its rfx_last_error()", r, (long long)all[(size_t)r * RW + 2 * W]);
                ctx->last_error = buf;
                return RFX_E_STATE;
            }
        // layouts: the fixed parts of all sources, then the extension words of all sources, in the receive buffer
        std::vector<int64_t> soff(W), scnt(W), roff(W), rcnt(W), swoff(W), swcnt(W), rwoff(W), rwcnt(W);
        int64_t n_in = 0, w_in = 0, tot = 0, before = 0, mx = 0;
        bool any_short = false;
        for (int r = 0; r < W; r++) {
            int64_t nr = 0, wr = 0;
            for (int s = 0; s < W; s++) {
                const int64_t a = all[(size_t)s * RW + r], b = all[(size_t)s * RW + W + r];
                nr += a; wr += b;
                mx = std::max(mx, std::max(a * fw, b));
            }
            if (nr * fw + wr > all[(size_t)r * RW + 2 * W + 1]) any_short = true;
            if (r < me) before += nr;
            tot += nr;
            if (r == me) { n_in = nr; w_in = wr; }
        }
        int64_t pr = 0, pw = n_in * fw;
        for (int s = 0; s < W; s++) {
            const int64_t a = all[(size_t)s * RW + me], b = all[(size_t)s * RW + W + me];
            roff[s] = pr; rcnt[s] = a * fw; pr += a * fw;
            rwoff[s] = pw; rwcnt[s] = b; pw += b;
            soff[s] = cut[s] * fw; scnt[s] = (cut[s + 1] - cut[s]) * fw;
            swoff[s] = wcut[s]; swcnt[s] = wcut[s + 1] - wcut[s];
        }
        *N = tot; *off = before;
        if (any_short) {                                  // the short ranks grow; everybody learns how that went
            if ((size_t)(n_in * fw + w_in) * 8 > c->recv_bytes) {
                const int sg = comm_grow(ctx, &c->recv, &c->recv_bytes, (size_t)(n_in * fw + w_in + 1024) * 8, ctx->stream, c->xs);
                if (sg != RFX_OK) st_local = sg;
            }
            RFX_TRY(agree("receive buffer"));
        }
        const int64_t lim = (int64_t)(c->limit_bytes / 8);
        const int64_t rounds = std::max<int64_t>(1, (mx + lim - 1) / lim);
        // the fixed parts go through the send buffer (packed; room_to_send() made it large enough before the matrix went
        // round), the extension words leave from where they lie
        if (st_local == RFX_OK && src.n > 0) {
            RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_pack<KW>, dim3(grid_for(src.n)), dim3(256), 0, ctx->stream,
                                                 (const KeyW<KW> *)src.key.as<KeyW<KW>>(), (const int32_t *)src.marker.as<int32_t>(),
                                                 (const int32_t *)src.left.as<int32_t>(), (const int32_t *)src.right.as<int32_t>(),
                                                 (const int64_t *)src.ext_off.as<int64_t>(), src.n, (uint64_t *)c->send));
            if (hipGetLastError() != hipSuccess) { st_local = RFX_E_HIP; ctx->last_error = "rfx_dev_sharded_assemble: pack launch failed"; }
        }
        RFX_TRY(alltoallv_words(c, (const uint64_t *)c->send, soff.data(), scnt.data(), (uint64_t *)c->recv, roff.data(), rcnt.data(), rounds, 1, ctx->stream));
        RFX_TRY(alltoallv_words(c, src.ext.as<uint64_t>(), swoff.data(), swcnt.data(), (uint64_t *)c->recv, rwoff.data(), rwcnt.data(), rounds, 1, ctx->stream));
        n_exchanges++; exchanged_records += src.n; exchanged_words += src.words;
        // what arrived -> a record set (a failure from here on is this rank's own: carried to the next collective)
        auto unpack = [&]() -> int {
            RFX_TRY(dev_records_alloc(ctx, dst, n_in, w_in, kw));
            DevBuf nw;
            RFX_HIP(nw.alloc((size_t)std::max<int64_t>(1, n_in) * 4, ctx->stream));
            if (n_in > 0) {
                RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_unpack<KW>, dim3(grid_for(n_in)), dim3(256), 0, ctx->stream, (const uint64_t *)c->recv, n_in,
                                                     dst.key.as<KeyW<KW>>(), dst.marker.as<int32_t>(), dst.left.as<int32_t>(), dst.right.as<int32_t>(),
                                                     nw.as<uint32_t>()));
                RFX_HIP(hipGetLastError());
            }
            RFX_TRY(exclusive_scan_u32_to_u64(ctx, nw.as<uint32_t>(), (uint64_t *)dst.ext_off.as<int64_t>(), n_in));
            if (w_in > 0)
                RFX_HIP(hipMemcpyAsync(dst.ext.p, (const uint64_t *)c->recv + n_in * fw, (size_t)w_in * 8, hipMemcpyDeviceToDevice, ctx->stream));
            dst.n = n_in; dst.words = w_in;
            return RFX_OK;
        };
        const int su = unpack();
        if (su != RFX_OK) { st_local = su; dst.n = 0; dst.words = 0; }
        return RFX_OK;
    }

    // ---- sortByKey: `in` (this rank's records, any order) -> `out` (this rank's share of the global sorted sequence) and the
    // logical partition starts inside it.  k_eff: the k whose (k - 1)-mer the key is (the k-mers of the first shuffle:
    // k + 1).  P = 0: no partition plan (ps untouched).
    int sort_exchange(const DevRecords &in, int P, int k_eff, int key_bits, DevRecords &out, DevBuf &ps, int64_t *N_out) {
        const int kw = in.kw;
        DevRecords sorted, merged;
        DevBuf ps0;
        next_arena();
        if (st_local == RFX_OK) {
            const int s1 = sort_records(ctx, in, 1, key_bits, sorted, ps0, k_eff);
            if (s1 != RFX_OK) st_local = s1;
        }
        if (st_local != RFX_OK) { sorted.n = 0; sorted.words = 0; sorted.kw = kw; }
        room_to_send(sorted.n, kw);
        // sample -> splitters -> cuts -> this rank's row of the matrix, all on the device
        auto plan = [&]() -> int {
            if (st_local != RFX_OK) { RFX_HIP(hipMemsetAsync(d + o_row(), 0, (size_t)(2 * W) * 8, ctx->stream)); RFX_HIP(hipMemsetAsync(d + o_aux(), 0, (size_t)(2 * W + 2) * 8, ctx->stream)); }
            RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_sample<KW>, dim3(grid_for(S + 1)), dim3(256), 0, ctx->stream,
                                                 (const KeyW<KW> *)sorted.key.as<KeyW<KW>>(), sorted.n, S, (uint64_t *)(d + o_samp())));
            RFX_HIP(hipGetLastError());
            return RFX_OK;
        };
        if (st_local == RFX_OK) { const int sp = plan(); if (sp != RFX_OK) st_local = sp; }
        else (void)plan();
        RFX_NCCL(nccl().AllGather(d + o_samp(), d + o_sall(), (size_t)(S * kw + 1), ncclInt64, c->comm, ctx->stream));
        if (st_local == RFX_OK) {
            auto cuts = [&]() -> int {
                RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_splitters<KW>, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t *)(d + o_sall()), W, S,
                                                     (uint64_t *)(d + o_split())));
                RFX_HIP(hipGetLastError());
                RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_cuts<KW>, dim3(1), dim3(128), 0, ctx->stream, (const KeyW<KW> *)sorted.key.as<KeyW<KW>>(), sorted.n,
                                                     (const uint64_t *)(d + o_split()), W, (const int64_t *)sorted.ext_off.as<int64_t>(),
                                                     d + o_row(), d + o_aux()));
                RFX_HIP(hipGetLastError());
                return RFX_OK;
            };
            const int sc = cuts();
            if (sc != RFX_OK) st_local = sc;
        }
        next_arena();                                     // (`in` is dead: its arena takes what arrives)
        int64_t N = 0, off = 0;
        RFX_TRY(exchange(sorted, kw, merged, &N, &off));
        next_arena();                                     // (`sorted` is dead once the exchange, stream-ordered before us, has read it)
        if (st_local == RFX_OK) {
            auto fin = [&]() -> int {
                RFX_TRY(sort_records(ctx, merged, 1, key_bits, out, ps0, k_eff));
                if (P > 0) {
                    RFX_HIP(ps.alloc((size_t)(P + 1) * 8, ctx->stream));
                    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_sh_part_starts<KW>, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream,
                                                         (const KeyW<KW> *)out.key.as<KeyW<KW>>(), out.n, N, off, P, ps.as<int64_t>()));
                    RFX_HIP(hipGetLastError());
                }
                return RFX_OK;
            };
            const int sf = fin();
            if (sf != RFX_OK) st_local = sf;
        }
        if (st_local != RFX_OK) { out.n = 0; out.words = 0; out.kw = kw; }
        if (N_out) *N_out = N;
        return RFX_OK;
    }

    // ---- everything to rank 0, rank after rank (the global arrival order); plain allocations, outside the arenas
    int gather_to_root(const DevRecords &in, DevRecords &dst) {
        const int kw = in.kw;
        std::vector<int64_t> row(2 * W, 0), aux(2 * (W + 1), 0);
        room_to_send(st_local == RFX_OK ? in.n : 0, kw);
        if (st_local == RFX_OK) {
            row[0] = in.n; row[W] = in.words;
            for (int t = 1; t <= W; t++) { aux[t] = in.n; aux[W + 1 + t] = in.words; }
        }
        // (pinned staging of the communicator: the copies are queued, the host buffer must outlive them)
        for (int i = 0; i < 2 * W; i++) h[o_row() + i] = row[i];
        for (int i = 0; i < 2 * (W + 1); i++) h[o_aux() + i] = aux[i];
        RFX_HIP(hipMemcpyAsync(d + o_row(), h + o_row(), (size_t)(2 * W) * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d + o_aux(), h + o_aux(), (size_t)(2 * W + 2) * 8, hipMemcpyHostToDevice, ctx->stream));
        Arena *saved = tl_arena;
        tl_arena = nullptr;
        int64_t N = 0, off = 0;
        DevRecords empty;
        empty.kw = kw;
        const int st = exchange(st_local == RFX_OK ? in : empty, kw, dst, &N, &off);
        tl_arena = saved;
        return st;
    }
};

}  // namespace

extern "C" {

// The extend stage on several GPUs (sortByKey as a range shuffle over RCCL): this rank's shard of the filtered
// (k-mer, count) list -> the contig text on rank 0, identical to rfx_dev_assemble[_w] on one GPU for the same
// prm->partitions.  include/reflexiv_hip.h has the contract.
int rfx_dev_sharded_assemble(rfx_ctx *ctx, rfx_comm *c, const uint64_t *d_keys, const int32_t *d_counts, int64_t n, const rfx_params *prm,
                             int64_t gather_below, char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs, int64_t *trace,
                             int64_t trace_cap, int64_t *n_trace) try {
    if (!ctx || !c || c->ctx != ctx || !prm || !out_len || n < 0 || (n > 0 && (!d_keys || !d_counts))) return RFX_E_ARG;
    const int k = prm->k;
    const bool wide = k > 31;
    const int kw = sub_words(k), aw = asm_words(k);
    if (k < 3 || aw > MAX_KEY_WORDS) { ctx->last_error = "rfx_dev_sharded_assemble: k = 3..124"; return RFX_E_ARG; }
    RFX_HIP(hipSetDevice(ctx->device));
    comm_options(c);
    *out_len = 0;
    if (out_contigs) *out_contigs = 0;
    if (n_trace) *n_trace = 0;
    if (gather_below < 0) {
        const char *e = getenv("RFX_SHARD_GATHER_BELOW");
        gather_below = e ? atoll(e) : (int64_t)32 << 20;
    }
    const int twin = wide ? RFX_TWIN_DS : prm->twin;
    int P = prm->partitions > 0 ? prm->partitions : 1;
    const int key_bits = 2 * (k - 1);
    const bool verbose = getenv("RFX_TRACE") != nullptr;

    Shard sh;
    sh.ctx = ctx; sh.c = c;
    // what a rank holds at most: its share of the 2 n records of all ranks, with slack for the imbalance of sampled splitters
    int64_t nn[1] = {n};
    RFX_TRY(rfx_comm_all_reduce_i64(c, nn, 1, 0));
    const int64_t n_all = nn[0];
    {
        const int64_t share = (2 * n_all) / c->world + (2 * n_all) / (4 * c->world) + (1 << 16);
        RFX_TRY(sh.init((size_t)std::max<int64_t>(share, 2 * n) * (176 + 24 * (size_t)(std::max(kw, aw) - 1)) + ((size_t)64 << 20)));
    }
    struct ArenaGuard { ~ArenaGuard() { tl_arena = nullptr; } } arena_guard;
    int64_t nt = 0;
    DevRecords a, b;
    DevBuf ps, ops;

    // the driver's variables (rfx_api.hip assemble_impl)
    int passes_done = 0, iterations = 0, partitionNumber = P, scramble = 2;
    int64_t contigNumber = 0, n_glob = 0;
    const bool extras = wide && prm->extras != 0;

    // a local step whose failure is this rank's own: remembered, not returned (the peers are waiting at the next collective)
#define SH_LOCAL(call)                                                             \
    do {                                                                           \
        if (sh.st_local == RFX_OK) { const int s_ = (call); if (s_ != RFX_OK) sh.st_local = s_; } \
    } while (0)

    bool gathered = false;
    DevRecords fin;                                   // the set on rank 0 once gathered
    auto gather_now = [&](const DevRecords &r) -> int {
        RFX_TRY(sh.gather_to_root(r, fin));
        RFX_TRY(sh.agree("gather"));
        gathered = true;
        return RFX_OK;
    };

    // 0. the count stage's order contract: ascending canonical k-mer over the ranks -> range-shard the survivors first
    //    (as records: key = the k-mer, marker = its count, one dummy extension word)
    sh.next_arena();
    DevRecords kr;
    {
        auto make = [&]() -> int {
            RFX_TRY(dev_records_alloc(ctx, kr, n, n, aw));
            if (n > 0) {
                RFX_HIP(hipMemcpyAsync(kr.key.p, d_keys, (size_t)n * 8 * aw, hipMemcpyDeviceToDevice, ctx->stream));
                RFX_HIP(hipMemcpyAsync(kr.marker.p, d_counts, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                RFX_HIP(hipMemsetAsync(kr.left.p, 0, (size_t)n * 4, ctx->stream));
                RFX_HIP(hipMemsetAsync(kr.right.p, 0, (size_t)n * 4, ctx->stream));
                RFX_HIP(hipMemsetAsync(kr.ext.p, 0, (size_t)n * 8, ctx->stream));
            }
            hipLaunchKernelGGL(k_sh_iota, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream, kr.ext_off.as<int64_t>(), n);
            RFX_HIP(hipGetLastError());
            kr.n = n; kr.words = n;
            return RFX_OK;
        };
        SH_LOCAL(make());
        if (sh.st_local != RFX_OK) { kr.n = 0; kr.words = 0; kr.kw = aw; }
    }
    if (2 * n_all <= gather_below) {
        // few enough from the start (a bacterial genome: D' << N): the survivors themselves go to rank 0, are put in
        // ascending order there, and the one-GPU driver does everything
        DevRecords all_k;
        RFX_TRY(sh.gather_to_root(kr, all_k));
        RFX_TRY(sh.agree("gather of the survivors"));
        tl_arena = nullptr;
        int64_t len = 0, ncont = 0, ntr = 0;
        int st_asm = RFX_OK;
        if (sh.me == 0) {
            auto drive = [&]() -> int {
                DevRecords ks;
                DevBuf ps0;
                RFX_TRY(sort_records(ctx, all_k, 1, 2 * k, ks, ps0, k + 1));
                return assemble_impl(ctx, wide, ks.key.as<uint64_t>(), ks.marker.as<int32_t>(), ks.n, prm, out, cap, &len, &ncont, trace, trace_cap, &ntr, nullptr);
            };
            try { st_asm = drive(); } catch (...) { st_asm = rfx_api_exception(ctx, "rfx_dev_sharded_assemble"); }
        }
        int64_t res[3] = {st_asm == RFX_E_CAP ? 1 : 0, st_asm == RFX_E_CAP ? len : 0, (st_asm != RFX_OK && st_asm != RFX_E_CAP) ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, res, 3, 1));
        if (res[2]) {
            if (st_asm != RFX_OK) return st_asm;
            ctx->last_error = "rfx_dev_sharded_assemble: the driver failed on rank 0; see its rfx_last_error()";
            return RFX_E_STATE;
        }
        if (res[0]) { *out_len = res[1]; return RFX_E_CAP; }
        if (sh.me == 0) { *out_len = len; if (out_contigs) *out_contigs = ncont; if (n_trace) *n_trace = ntr; }
        return RFX_OK;
    }
    {
        DevRecords ks;
        int64_t Nk = 0;
        RFX_TRY(sh.sort_exchange(kr, 0, k + 1, 2 * k, ks, ps, &Nk));
        // KmerReverseComplement + ForwardSubKmerExtraction  :168-176 on this rank's range of the ascending list
        sh.next_arena();
        SH_LOCAL(rc_expand_subkmer(ctx, ks.key.as<uint64_t>(), ks.marker.as<int32_t>(), ks.n, k, a));
        if (sh.st_local != RFX_OK) { a.n = 0; a.words = 0; a.kw = kw; }
    }
    const double t0 = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }();
    // sortByKey + FilterForkSubKmer[WithErrorCorrection]  :179-186
    RFX_TRY(sh.sort_exchange(a, P, k, key_bits, b, ps, &n_glob));
    sh.next_arena();
    SH_LOCAL(fork_filter(ctx, false, b, ps.as<int64_t>(), P, k, prm->min_error_cov, twin, a, ops));
    // ReflectedSubKmerExtractionFromForward  :188-189
    sh.next_arena();
    SH_LOCAL(reflect_from_forward(ctx, a, k, b));
    if (sh.st_local != RFX_OK) { b.n = 0; b.words = 0; b.kw = kw; }
    // sortByKey + FilterForkReflectedSubKmer[WithErrorCorrection]  :191-198
    RFX_TRY(sh.sort_exchange(b, P, k, key_bits, a, ps, &n_glob));
    sh.next_arena();
    SH_LOCAL(fork_filter(ctx, true, a, ps.as<int64_t>(), P, k, prm->min_error_cov, twin, b, ops));
    // kmerRandomReflection on the filter's output partitions  :204-205 -- a partition that began on an earlier rank brings
    // the parity of its surviving records there
    sh.next_arena();
    {
        const int32_t *d_carry = nullptr;
        if (sh.st_local != RFX_OK) { b.n = 0; b.words = 0; b.kw = kw; RFX_HIP(ops.alloc((size_t)(P + 1) * 8, ctx->stream)); RFX_HIP(hipMemsetAsync(ops.p, 0, (size_t)(P + 1) * 8, ctx->stream)); }
        RFX_TRY(sh.compute(ctx, ops.as<int64_t>(), P, nullptr, b.n, &d_carry));
        SH_LOCAL(random_reflection(ctx, b, ops.as<int64_t>(), P, k, a, d_carry));
        if (sh.st_local != RFX_OK) { a.n = 0; a.words = 0; a.kw = kw; }
    }

    // one pass of the loop, sharded; -> the record count over all ranks (count() of the stop rule, the trace)
    auto one_pass = [&](int stage, int start_marker) -> int {
        RFX_TRY(sh.sort_exchange(a, P, k, key_bits, b, ps, nullptr));                 // sortByKey :211,:235,:247,:286
        if (sh.st_local != RFX_OK) {                                                  // (the carry's collective must still be met)
            const int32_t *dc = nullptr;
            DevBuf z;
            RFX_HIP(z.alloc((size_t)(P + 1) * 8, ctx->stream));
            RFX_HIP(hipMemsetAsync(z.p, 0, (size_t)(P + 1) * 8, ctx->stream));
            RFX_TRY(sh.compute(ctx, z.as<int64_t>(), P, nullptr, 0, &dc));
            a.n = 0; a.words = 0; a.kw = kw;
        } else {
            sh.next_arena();
            const int before = sh.carry_slot;
            const int se = extend_pass(ctx, b, ps.as<int64_t>(), P, k, twin, stage, a, ops, start_marker, &sh);
            if (se != RFX_OK) {
                sh.st_local = se; a.n = 0; a.words = 0; a.kw = kw;
                if (sh.carry_slot == before) {                                    // the pass failed before its carry: the peers are in that all-gather
                    const int32_t *dc = nullptr;
                    RFX_HIP(ops.alloc((size_t)(P + 1) * 8, ctx->stream));
                    RFX_HIP(hipMemsetAsync(ops.p, 0, (size_t)(P + 1) * 8, ctx->stream));
                    RFX_TRY(sh.compute(ctx, ops.as<int64_t>(), P, nullptr, 0, &dc));
                }
            }
        }
        int64_t v[3] = {a.n, a.words, sh.st_local != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, v, 3, 0));
        if (sh.st_local != RFX_OK) return sh.st_local;
        if (v[2]) { ctx->last_error = "rfx_dev_sharded_assemble: a peer failed in an extend pass; see its rfx_last_error()"; return RFX_E_STATE; }
        n_glob = v[0];
        if (trace && nt < trace_cap) trace[nt] = n_glob;
        nt++;
        passes_done++;
        if (verbose && sh.me == 0) fprintf(stderr, "sharded pass %d: %lld records on all ranks (%lld here)\n", passes_done, (long long)n_glob, (long long)a.n);
        return RFX_OK;
    };
    // records on all ranks before the first pass
    {
        int64_t v[2] = {a.n, sh.st_local != RFX_OK ? 1 : 0};
        RFX_TRY(rfx_comm_all_reduce_i64(c, v, 2, 0));
        if (sh.st_local != RFX_OK) return sh.st_local;
        if (v[1]) { ctx->last_error = "rfx_dev_sharded_assemble: a peer failed before the loop; see its rfx_last_error()"; return RFX_E_STATE; }
        n_glob = v[0];
    }
    // the five passes before the loop (:221-254), then the loop (:265-296; k > 31: 64 :582-661) -- sharded while the set is large
    while (!gathered) {
        if (n_glob <= gather_below) { RFX_TRY(gather_now(a)); break; }
        if (passes_done < 5) {
            if (passes_done >= 1) iterations++;
            RFX_TRY(one_pass(passes_done < 4 ? 0 : 1, 2));
            continue;
        }
        if (iterations > prm->max_iter) { RFX_TRY(gather_now(a)); break; }
        if (extras && iterations + 1 >= prm->min_iter + 3) { RFX_TRY(gather_now(a)); break; }     // 64 :584-619 wants the whole set
        // (from here on the one-GPU driver's loop, statement for statement, on global counts)
        iterations++;
        bool stop = false;
        if (wide) {
            if (iterations >= prm->min_iter + 3 && iterations % 3 == 0) {         // 64 :621-647
                if (contigNumber == n_glob) { if (scramble == 2) { scramble = 3; contigNumber = n_glob; } else stop = true; }
                else contigNumber = n_glob;
            }
        } else if (iterations >= prm->min_iter && iterations % 3 == 0) {          // :267-283
            if (contigNumber == n_glob) stop = true;
            else {
                contigNumber = n_glob;
                if (prm->coalesce && partitionNumber >= 16 && n_glob / partitionNumber <= 20) { partitionNumber = partitionNumber / 4 + 1; P = partitionNumber; }
            }
        }
        if (stop) { iterations = prm->max_iter + 1; RFX_TRY(gather_now(a)); break; }     // (the resumed driver runs no further pass)
        RFX_TRY(one_pass(2, wide && scramble == 3 ? 1 : 2));
    }
    if (verbose && sh.me == 0) {
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        fprintf(stderr, "sharded extend: %d passes sharded, %d exchanges (%lld records, %lld words left this rank), %.3f ms; gathered %lld records\n",
                passes_done, sh.n_exchanges, (long long)sh.exchanged_records, (long long)sh.exchanged_words, ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - t0,
                (long long)n_glob);
    }
    tl_arena = nullptr;

    // rank 0: the one-GPU driver takes the loop up; its outcome is every rank's (a short text buffer is RFX_E_CAP on EVERY
    // rank with the length needed, so that the callers' retries stay collective)
    int64_t len = 0, ncont = 0, ntr = nt;
    int st_asm = RFX_OK;
    if (sh.me == 0) {
        AsmResume rs{&fin, passes_done, iterations, contigNumber, scramble, P, partitionNumber, nt};
        try {
            st_asm = assemble_impl(ctx, wide, nullptr, nullptr, 0, prm, out, cap, &len, &ncont, trace, trace_cap, &ntr, &rs);
        } catch (...) { st_asm = rfx_api_exception(ctx, "rfx_dev_sharded_assemble"); }
    }
    int64_t res[3] = {st_asm == RFX_E_CAP ? 1 : 0, st_asm == RFX_E_CAP ? len : 0, (st_asm != RFX_OK && st_asm != RFX_E_CAP) ? 1 : 0};
    RFX_TRY(rfx_comm_all_reduce_i64(c, res, 3, 1));
    if (res[2]) {
        if (st_asm != RFX_OK) return st_asm;
        ctx->last_error = "rfx_dev_sharded_assemble: the driver failed on rank 0; see its rfx_last_error()";
        return RFX_E_STATE;
    }
    if (res[0]) { *out_len = res[1]; return RFX_E_CAP; }
    if (sh.me == 0) { *out_len = len; if (out_contigs) *out_contigs = ncont; if (n_trace) *n_trace = ntr; }
    return RFX_OK;
#undef SH_LOCAL
} RFX_API_CATCH(ctx)

}  // extern "C"
