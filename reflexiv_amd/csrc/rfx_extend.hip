// rfx_extend.hip -- one reflexible extend-and-merge pass (K10/K11/K12 of SURVEY.md 2.3).
//
// Replaces the bodies of
//   ExtendReflexivKmer.call                    P/ReflexivMain.java:2048-2362 (DS :3040-3329)
//   ExtendReflexivKmerToArrayFirstTime.call    P/ReflexivMain.java:1594-1974 (DS :2589-2971)
//   ExtendReflexivKmerToArrayLoop.call         P/ReflexivMain.java:792-1519  (DS :1776-2518)
// which share one algorithm and one word layout (a single long is a
// one-word array), and of their k > 31 twins DSExtendReflexivKmer / ...ToArrayFirstTime / ...ToArrayLoop
// (P/ReflexivDSMain64.java:9465-10070, 8733-9463, 7446-8731): the same pass on (k-1)-mer keys of KW
// words of 31 bases (templates on KW; sequence level).
//
// The reference scans a sorted partition sequentially with a one-record holder and a
// marker that toggles on every emission.  Both dependencies are local: the holder never
// survives a key change, so every equal-key run resolves on its own, and the toggling
// marker is the parity of the emission index inside the partition.  Hence, on the GPU:
//   1. resolve : the thread owning a run head walks its run (<= 2 records once the fork
//                filters ran) and writes one descriptor per emission -- flip(a) or
//                merge(R, F) with the merged left/right -- into the run's own index range;
//   2. scan    : prefix sums of emissions and of output words give every descriptor its
//                output slot, its partition-relative parity (= orientation) and its
//                extension offset;
//   3. emit    : keys and extension words are produced base-exactly from the source
//                records in the reference's packing (word 0 = f bases under a sentinel,
//                then 31 bases per word); long records are spread over a workgroup.
// Integer/byte work bound by HBM; no MFMA.
#include "rfx_internal.h"
#include "rfx_device.h"

using namespace rfxd;

namespace {

constexpr int32_t BLOCKED = INT32_MIN;
constexpr int EMIT_SHORT = 2;           // words the record's own thread emits; longer ones go word-parallel

struct Desc {          // one emission
    uint32_t a, b;     // flip: a = source; merge: a = reflected source R, b = forward source F
    int32_t left, right;
    uint32_t type;     // 0 none, 1 flip, 2 merge
    uint32_t len;      // extension length of the output in bases
};

__global__ void k_ext_len(const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext, int64_t n,
                          uint32_t *__restrict__ len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t b = ext_off[i], nw = ext_off[i + 1] - b;
    len[i] = (uint32_t)((nw - 1) * 31 + sentinel_len(ext[b]));           // :820-823
}

// B.5 on one equal-key run; descriptors go to desc[i .. i+emissions)
template <int KW>
__global__ void k_resolve(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                          const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                          const uint32_t *__restrict__ len, int64_t n, int twin, int stage,
                          Desc *__restrict__ desc, uint32_t *__restrict__ flag, uint32_t *__restrict__ onw,
                          int *__restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KeyW<KW> kk = key[i];
    if (i > 0 && key_eq(key[i - 1], kk)) return;               // not a run head
    int64_t o = i;                                             // next descriptor slot
    int64_t holder = i;                                        // :797-799
    int64_t s = i + 1;
#define PUT(TYPE, A, B, L, R, LEN) do { \
        Desc d_; d_.a = (uint32_t)(A); d_.b = (uint32_t)(B); d_.left = (L); d_.right = (R); \
        d_.type = (TYPE); d_.len = (uint32_t)(LEN); desc[o] = d_; flag[o] = 1u; \
        onw[o] = ((uint32_t)(LEN) + 30u) / 31u; \
        if (stage == 0 && (LEN) > 31) atomicOr(status, 1); \
        o++; } while (0)
    for (; s < n && key_eq(key[s], kk); s++) {
        if (holder < 0) { holder = s; continue; }                                   // :813-814
        if (marker[s] == marker[holder]) {                                          // :845-854
            PUT(1u, s, 0, left[s], right[s], len[s]);
            continue;
        }
        const int64_t F = marker[s] == 1 ? s : holder, R = marker[s] == 1 ? holder : s;
        const int32_t a = left[F], b = right[R];
        const int64_t lenF = len[F], lenR = len[R];
        int64_t d;
        if ((a < 0 && b < 0) || (a >= 0 && b >= 0)) d = -1;                         // :825-832
        else if (s == F) {                                                          // :833-840
            if (a >= 0 && a - lenR >= 0) d = a - lenR;
            else if (b >= 0 && b - lenF >= 0) d = b - lenF;
            else d = BLOCKED;
        } else {                                                                    // :868-875
            if (b >= 0 && b - lenF >= 0) d = b - lenF;
            else if (twin == RFX_TWIN_RDD) {                                        // :872-873
                int64_t hr = right[F];
                d = (a >= 0 && hr - lenR >= 0) ? hr - lenR : (int64_t)BLOCKED;
            } else d = (a >= 0 && a - lenR >= 0) ? a - lenR : (int64_t)BLOCKED;     // DS :1856-1857
        }
        if (d == BLOCKED) { PUT(1u, s, 0, left[s], right[s], len[s]); continue; }   // :841-843
        int32_t L, Rt;
        if (d < 0) { L = left[R]; Rt = right[F]; }                                  // :1214-1218
        else if (left[F] > 0) { L = (int32_t)d; Rt = right[F]; }                    // :1220-1226
        else { L = left[R]; Rt = (int32_t)d; }                                      // :1227-1233
        PUT(2u, R, F, L, Rt, lenR + lenF);
        holder = -1;
    }
    if (holder >= 0) PUT(1u, holder, 0, left[holder], right[holder], len[holder]);  // :886-893, :902
    for (; o < s; o++) { flag[o] = 0u; onw[o] = 0u; desc[o].type = 0u; }
#undef PUT
}

// view of a source record
template <int KW> struct SrcRec {
    KeyW<KW> key; const uint64_t *w; int f; int marker; int64_t len;
};
template <int KW>
__device__ __forceinline__ SrcRec<KW> load_src(uint32_t i, const KeyW<KW> *__restrict__ key,
                                               const int32_t *__restrict__ marker,
                                               const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                                               const uint32_t *__restrict__ len) {
    SrcRec<KW> r;
    r.key = key[i]; r.marker = marker[i]; r.w = ext + ext_off[i]; r.len = len[i];
    r.f = sentinel_len(r.w[0]);
    return r;
}
// base p of the full sequence of a record (marker 1: key||ext, marker 2: ext||key)
template <int KW>
__device__ __forceinline__ unsigned rec_base(const SrcRec<KW> &r, int sub, int64_t p) {
    if (r.marker == 1) return p < sub ? key_base_w<KW>(r.key, sub, (int)p) : ext_base(r.w, r.f, p - sub);
    return p < r.len ? ext_base(r.w, r.f, p) : key_base_w<KW>(r.key, sub, (int)(p - r.len));
}

template <int KW> struct OutSeq {         // S_out = flip: S_a ; merge: S_R || ext_F
    SrcRec<KW> a, b; int type; int64_t lenSa;
};
template <int KW>
__device__ __forceinline__ unsigned out_base(const OutSeq<KW> &s, int sub, int64_t p) {
    if (s.type == 1 || p < s.lenSa) return rec_base<KW>(s.a, sub, p);
    return ext_base(s.b.w, s.b.f, p - s.lenSa);
}

// words [w0, w1) of the output extension (bases q of ext_out = S_out[q + shift])
template <int KW>
__device__ __forceinline__ void emit_words(const OutSeq<KW> &s, int sub, int64_t shift, int64_t L, int64_t w0,
                                           int64_t w1, int64_t wstep, uint64_t *__restrict__ dst) {
    const int64_t nw = (L + 30) / 31;
    const int f = (int)(L - 31 * (nw - 1));
    for (int64_t w = w0; w < w1; w += wstep) {
        uint64_t x; int64_t q; int cnt;
        if (w == 0) { x = 1; q = 0; cnt = f; } else { x = 0; q = f + 31 * (w - 1); cnt = 31; }
        for (int j = 0; j < cnt; j++) x = (x << 2) | out_base<KW>(s, sub, shift + q + j);
        dst[w] = x;
    }
}

__device__ __forceinline__ int part_of(const int64_t *__restrict__ ps, int P, int64_t i) {
    int lo = 0, hi = P;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ps[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

template <int KW>
__global__ void k_emit(const Desc *__restrict__ desc, const uint32_t *__restrict__ flag,
                       const uint64_t *__restrict__ oidx, const uint64_t *__restrict__ owoff, int64_t n,
                       const int64_t *__restrict__ ps, int P, int sub, int start_marker,
                       const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                       const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                       const uint32_t *__restrict__ len,
                       KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker, int64_t *__restrict__ oext_off,
                       uint64_t *__restrict__ oext, int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) { oext_off[oidx[n]] = (int64_t)owoff[n]; return; }
    if (i > n || !flag[i]) return;
    const Desc d = desc[i];
    const int64_t j = (int64_t)oidx[i];
    const int p = part_of(ps, P, i);
    // randomReflexivMarker starts at 2 in every task (1 once param.scramble == 3 in the k > 31 array loop,
    // P/ReflexivDSMain64.java:7484-7486) and toggles on every emission (:770, :1058-1062, :1242, :1514):
    // orientation = parity of the emission index
    const int m = ((j - (int64_t)oidx[ps[p]]) & 1) ? 3 - start_marker : start_marker;
    OutSeq<KW> s;
    s.type = (int)d.type;
    s.a = load_src<KW>(d.a, key, marker, ext_off, ext, len);
    s.lenSa = s.a.len + sub;
    if (d.type == 2) s.b = load_src<KW>(d.b, key, marker, ext_off, ext, len); else s.b = s.a;
    const int64_t L = d.len;
    const int64_t wo = (int64_t)owoff[i];
    omarker[j] = m; oleft[j] = d.left; oright[j] = d.right; oext_off[j] = wo;
    if (KW == 1 && L <= 31) {
        // Single-word output (every record of the first passes): S_out has at most (k-1) + 31 <= 61
        // bases, so it is assembled as one 128-bit value with shifts and cut into key and extension --
        // the same bases the per-base walk below produces (seq_of / oriented in the oracle), ~40
        // instructions instead of ~60 base lookups.
        typedef unsigned __int128 u128;
        const u128 ea = (u128)(s.a.w[0] & low_mask((int)s.a.len));
        u128 S = s.a.marker == 1 ? (((u128)s.a.key.w[0] << (2 * (int)s.a.len)) | ea) : ((ea << (2 * sub)) | (u128)s.a.key.w[0]);
        if (d.type == 2) S = (S << (2 * (int)s.b.len)) | (u128)(s.b.w[0] & low_mask((int)s.b.len));
        uint64_t kk, eb;
        if (m == 1) { kk = (uint64_t)(S >> (2 * (int)L)); eb = (uint64_t)S & low_mask((int)L); }
        else { eb = (uint64_t)(S >> (2 * sub)); kk = (uint64_t)S & low_mask(sub); }
        okey[j].w[0] = kk;
        oext[wo] = (1ULL << (2 * (int)L)) | eb;
        return;
    }
    // key: first (m == 1) or last (m == 2) k-1 bases of S_out
    const int64_t kshift = m == 1 ? 0 : L;
    if (d.type == 1 && s.a.marker == m) okey[j] = s.a.key;
    else okey[j] = build_key<KW>(sub, [&](int t) { return out_base<KW>(s, sub, kshift + t); });
    const int64_t nw = (L + 30) / 31;
    if (nw > EMIT_SHORT) return;                     // k_emit_words: one thread per output word
    if (d.type == 1 && s.a.marker == m) {            // same orientation: the words are unchanged
        for (int64_t w = 0; w < nw; w++) oext[wo + w] = s.a.w[w];
        return;
    }
    emit_words<KW>(s, sub, m == 1 ? sub : 0, L, 0, nw, 1, oext + wo);
}

// Extensions longer than EMIT_SHORT words, one thread per OUTPUT WORD: thread t finds the emission
// that owns word t of the output array by a binary search in the word-offset scan (zero-length
// entries share their successor's offset, so the last entry with offset <= t is the owner) and
// writes that one word.  A 2.6 Mbp contig (84 K words) is 84 K threads; no per-record queues.
template <int KW>
__global__ void k_emit_words(const Desc *__restrict__ desc, const uint64_t *__restrict__ oidx,
                             const uint64_t *__restrict__ owoff, int64_t n, const int64_t *__restrict__ ps, int P, int sub,
                             int start_marker, const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                             const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                             const uint32_t *__restrict__ len, uint64_t *__restrict__ oext) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)owoff[n]) return;
    int64_t lo = 0, hi = n;                          // owoff[lo] <= t < owoff[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)owoff[mid] <= t) lo = mid; else hi = mid;
    }
    const int64_t i = lo;
    const Desc d = desc[i];
    const int64_t L = d.len, nw = (L + 30) / 31, wo = (int64_t)owoff[i], w = t - wo;
    if (nw <= EMIT_SHORT) return;                    // k_emit wrote it
    const int64_t j = (int64_t)oidx[i];
    const int p = part_of(ps, P, i);
    const int m = ((j - (int64_t)oidx[ps[p]]) & 1) ? 3 - start_marker : start_marker;
    OutSeq<KW> s;
    s.type = (int)d.type;
    s.a = load_src<KW>(d.a, key, marker, ext_off, ext, len);
    s.lenSa = s.a.len + sub;
    if (d.type == 2) s.b = load_src<KW>(d.b, key, marker, ext_off, ext, len); else s.b = s.a;
    if (d.type == 1 && s.a.marker == m) oext[wo + w] = s.a.w[w];
    else emit_words<KW>(s, sub, m == 1 ? sub : 0, L, w, w + 1, 1, oext + wo);
}

// output partition starts + the pass summary the host reads back in ONE copy:
// summary = {records out, words out, status}
__global__ void k_out_part_start(const int64_t *__restrict__ ps, int P, const uint64_t *__restrict__ oidx,
                                 int64_t *__restrict__ ops, const uint64_t *__restrict__ owoff, int64_t n,
                                 const int *__restrict__ status, uint64_t *__restrict__ summary) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= P) ops[p] = (int64_t)oidx[ps[p]];
    if (p == 0) { summary[0] = oidx[n]; summary[1] = owoff[n]; summary[2] = (uint64_t)(unsigned)status[0]; }
}

inline unsigned grid_for(int64_t n, int block = 256) { return (unsigned)ceil_div(n > 0 ? n : 1, block); }

}  // namespace

namespace rfx {

#define RFX_KW_SWITCH(kw, ...)                                                \
    switch (kw) {                                                             \
    case 1: { constexpr int KW = 1; __VA_ARGS__; } break;                     \
    case 2: { constexpr int KW = 2; __VA_ARGS__; } break;                     \
    case 3: { constexpr int KW = 3; __VA_ARGS__; } break;                     \
    case 4: { constexpr int KW = 4; __VA_ARGS__; } break;                     \
    default: return RFX_E_ARG;                                                \
    }

int extend_pass(rfx_ctx *ctx, const DevRecords &in, const int64_t *d_part_start, int P, int k, int twin,
                int stage, DevRecords &out, DevBuf &out_part_start, int start_marker) {
    const int64_t n = in.n;
    const int sub = k - 1;
    const int kw = in.kw;
    if (n > (int64_t)0xFFFFFFFFLL) return RFX_E_LIMIT;
    if (kw != sub_words(k) || (start_marker != 1 && start_marker != 2)) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, in.words, kw));
    RFX_HIP(out_part_start.alloc((size_t)(P + 1) * 8, ctx->stream));
    DevBuf len, desc, flag, onw, oidx, owoff, status;
    const int64_t a = n ? n : 1;
    RFX_HIP(len.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(desc.alloc((size_t)a * sizeof(Desc), ctx->stream));
    RFX_HIP(flag.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(onw.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(oidx.alloc((size_t)(n + 1) * 8, ctx->stream));
    RFX_HIP(owoff.alloc((size_t)(n + 1) * 8, ctx->stream));
    RFX_HIP(status.alloc(4, ctx->stream));
    RFX_HIP(hipMemsetAsync(status.p, 0, 4, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_ext_len, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                           (const int64_t *)in.ext_off.as<int64_t>(), (const uint64_t *)in.ext.as<uint64_t>(), n,
                           len.as<uint32_t>());
        RFX_HIP(hipGetLastError());
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_resolve<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                           (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                           (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(),
                           (const uint32_t *)len.as<uint32_t>(), n, twin, stage, desc.as<Desc>(),
                           flag.as<uint32_t>(), onw.as<uint32_t>(), status.as<int>()));
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, flag.as<uint32_t>(), oidx.as<uint64_t>(), n));
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, onw.as<uint32_t>(), owoff.as<uint64_t>(), n));
    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_emit<KW>, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream, (const Desc *)desc.as<Desc>(),
                       (const uint32_t *)flag.as<uint32_t>(), (const uint64_t *)oidx.as<uint64_t>(),
                       (const uint64_t *)owoff.as<uint64_t>(), n, d_part_start, P, sub, start_marker,
                       (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const int64_t *)in.ext_off.as<int64_t>(), (const uint64_t *)in.ext.as<uint64_t>(),
                       (const uint32_t *)len.as<uint32_t>(), out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(),
                       out.ext_off.as<int64_t>(), out.ext.as<uint64_t>(), out.left.as<int32_t>(),
                       out.right.as<int32_t>()));
    RFX_HIP(hipGetLastError());
    if (in.words > n) {          // some record has more than one word (output words <= input words)
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_emit_words<KW>, dim3(grid_for(in.words)), dim3(256), 0, ctx->stream, (const Desc *)desc.as<Desc>(),
                           (const uint64_t *)oidx.as<uint64_t>(), (const uint64_t *)owoff.as<uint64_t>(), n, d_part_start, P,
                           sub, start_marker, (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                           (const int64_t *)in.ext_off.as<int64_t>(), (const uint64_t *)in.ext.as<uint64_t>(),
                           (const uint32_t *)len.as<uint32_t>(), out.ext.as<uint64_t>()));
        RFX_HIP(hipGetLastError());
    }
    DevBuf summary;
    RFX_HIP(summary.alloc(24, ctx->stream));
    hipLaunchKernelGGL(k_out_part_start, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream, d_part_start, P,
                       (const uint64_t *)oidx.as<uint64_t>(), out_part_start.as<int64_t>(),
                       (const uint64_t *)owoff.as<uint64_t>(), n, (const int *)status.as<int>(), summary.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    uint64_t tot[3] = {0, 0, 0};
    RFX_HIP(hipMemcpyAsync(tot, summary.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipStreamSynchronize(ctx->stream));
    const int st = (int)tot[2];
    out.n = (int64_t)tot[0]; out.words = (int64_t)tot[1];
    if (st) { ctx->last_error = "extend pass: single-word stage produced an extension > 31 bases"; return RFX_E_STATE; }
    return RFX_OK;
}

}  // namespace rfx
