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
#include <utility>
#include <vector>

using namespace rfxd;

namespace {

constexpr int32_t BLOCKED = INT32_MIN;
constexpr int EMIT_SHORT = 2;           // words the record's own thread emits; longer ones go word-parallel

struct Desc {          // one emission
    uint32_t a, b;     // flip: a = source, b = the orientation to emit in (0: the one the emission index gives);
                       // merge: a = reflected source R, b = forward source F
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
__device__ __forceinline__ void resolve_at(int64_t i, const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                          const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                          const uint32_t *__restrict__ len, int64_t n, int twin, int stage,
                          Desc *__restrict__ desc, uint32_t *__restrict__ flag, uint32_t *__restrict__ onw,
                          int *__restrict__ status) {
    if (i >= n) return;
    const KeyW<KW> kk = key[i];
    if (i > 0 && key_eq(key[i - 1], kk)) return;               // not a run head
    int64_t o = i;                                             // next descriptor slot
    int64_t holder = i;                                        // :797-799
    int64_t s = i + 1;
#define PUT(TYPE, A, B, L, R, LEN) do { \
        Desc d_; d_.a = (uint32_t)(A); d_.b = (uint32_t)(B); d_.left = (L); d_.right = (R); \
        d_.type = (TYPE); d_.len = (uint32_t)(LEN); \
        /* single-word stage: a longer output is an error (RFX_E_STATE); it is cut to one word so that nothing is */ \
        /* written past the one word per emission that stage reserves */ \
        if (stage == 0 && (LEN) > 31) { atomicOr(status, 1); d_.len = 31u; } \
        desc[o] = d_; flag[o] = 1u; \
        onw[o] = (d_.len + 30u) / 31u; \
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

template <int KW>
__global__ void k_resolve(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                          const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                          const uint32_t *__restrict__ len, int64_t n, int twin, int stage,
                          Desc *__restrict__ desc, uint32_t *__restrict__ flag, uint32_t *__restrict__ onw,
                          int *__restrict__ status) {
    resolve_at<KW>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, key, marker, left, right, len, n, twin, stage, desc, flag,
                   onw, status);
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

// ---- the same sequences WORD-WISE (round 3): 1..31 consecutive bases of an extension, a key, a record, an output
// sequence as one right-aligned value -- a word or two loaded and shifted where the base-by-base walk above loads and
// shifts once per base (a mid-size pass spent 90 of its 220 us of kernel time in those walks).
// bases [q, q + cnt) of an extension in the reference's word layout (word 0: f bases under the sentinel; then 31 a word)
__device__ __forceinline__ uint64_t ext_fetch(const uint64_t *__restrict__ w, int f, int64_t q, int cnt) {
    const int64_t v = q + (31 - f);                    // word 0 read as 31 bases, its first 31 - f are padding
    const int64_t wi = v / 31;
    const int o = (int)(v - 31 * wi);
    const uint64_t hi = wi == 0 ? (w[0] & low_mask(f)) : (w[wi] & low_mask(31));        // (bits 62-63 of a later word may stray)
    if (o + cnt <= 31) return (hi >> (2 * (31 - o - cnt))) & low_mask(cnt);
    const uint64_t lo = w[wi + 1] & low_mask(31);
    const int n1 = 31 - o, n2 = cnt - n1;
    return ((hi & low_mask(n1)) << (2 * n2)) | (lo >> (2 * (31 - n2)));
}
// bases [q, q + cnt) of a key of `sub` bases (31 a word, the last word the rest, right-aligned)
template <int KW>
__device__ __forceinline__ uint64_t key_fetch(const KeyW<KW> &key, int sub, int q, int cnt) {
    if (KW == 1) return (key.w[0] >> (2 * (sub - q - cnt))) & low_mask(cnt);
    const int res = sub - 31 * (KW - 1);
    const int wi = q / 31, o = q - 31 * wi;
    auto word31 = [&](int i) __attribute__((always_inline)) -> uint64_t {      // word i as 31 left-aligned bases
        uint64_t x = key.w[0];
#pragma unroll
        for (int t = 1; t < KW; t++) if (i == t) x = key.w[t];
        return i == KW - 1 ? x << (2 * (31 - res)) : x;
    };
    const uint64_t hi = word31(wi);
    if (o + cnt <= 31) return (hi >> (2 * (31 - o - cnt))) & low_mask(cnt);
    const uint64_t lo = word31(wi + 1);
    const int n1 = 31 - o, n2 = cnt - n1;
    return ((hi & low_mask(n1)) << (2 * n2)) | (lo >> (2 * (31 - n2)));
}
// bases [p, p + cnt) of a record's full sequence (marker 1: key || ext, marker 2: ext || key)
template <int KW>
__device__ __forceinline__ uint64_t rec_fetch(const SrcRec<KW> &r, int sub, int64_t p, int cnt) {
    const int64_t lenA = r.marker == 1 ? (int64_t)sub : r.len;
    uint64_t x = 0;
    int n1 = 0;
    if (p < lenA) {
        n1 = (int)(lenA - p < cnt ? lenA - p : cnt);
        x = r.marker == 1 ? key_fetch<KW>(r.key, sub, (int)p, n1) : ext_fetch(r.w, r.f, p, n1);
    }
    const int n2 = cnt - n1;
    if (n2 > 0) {
        const int64_t q = p + n1 - lenA;
        const uint64_t y = r.marker == 1 ? ext_fetch(r.w, r.f, q, n2) : key_fetch<KW>(r.key, sub, (int)q, n2);
        x = (x << (2 * n2)) | y;
    }
    return x;
}
// bases [p, p + cnt) of S_out
template <int KW>
__device__ __forceinline__ uint64_t out_fetch(const OutSeq<KW> &s, int sub, int64_t p, int cnt) {
    if (s.type == 1) return rec_fetch<KW>(s.a, sub, p, cnt);
    uint64_t x = 0;
    int n1 = 0;
    if (p < s.lenSa) {
        n1 = (int)(s.lenSa - p < cnt ? s.lenSa - p : cnt);
        x = rec_fetch<KW>(s.a, sub, p, n1);
    }
    const int n2 = cnt - n1;
    if (n2 > 0) x = (x << (2 * n2)) | ext_fetch(s.b.w, s.b.f, p + n1 - s.lenSa, n2);
    return x;
}
// the key made of the `sub` bases of S_out from `from` on
template <int KW>
__device__ __forceinline__ KeyW<KW> out_key(const OutSeq<KW> &s, int sub, int64_t from) {
    KeyW<KW> key;
#pragma unroll
    for (int w = 0; w < KW; w++) {
        const int nb = w < KW - 1 ? 31 : sub - 31 * (KW - 1);
        key.w[w] = out_fetch<KW>(s, sub, from + 31 * w, nb);
    }
    return key;
}

// words [w0, w1) of the output extension (bases q of ext_out = S_out[q + shift])
template <int KW>
__device__ __forceinline__ void emit_words(const OutSeq<KW> &s, int sub, int64_t shift, int64_t L, int64_t w0,
                                           int64_t w1, int64_t wstep, uint64_t *__restrict__ dst) {
    const int64_t nw = (L + 30) / 31;
    const int f = (int)(L - 31 * (nw - 1));
    for (int64_t w = w0; w < w1; w += wstep) {
        if (w == 0) dst[0] = (1ULL << (2 * f)) | out_fetch<KW>(s, sub, shift, f);
        else dst[w] = out_fetch<KW>(s, sub, shift + f + 31 * (w - 1), 31);
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
__device__ __forceinline__ void emit_at(int64_t i, const Desc *__restrict__ desc, const uint32_t *__restrict__ flag,
                       const uint64_t *__restrict__ oidx, const uint64_t *__restrict__ owoff, int64_t n,
                       const int64_t *__restrict__ ps, int P, int sub, int start_marker, const int32_t *__restrict__ carry,
                       const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                       const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                       const uint32_t *__restrict__ len,
                       KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker, int64_t *__restrict__ oext_off,
                       uint64_t *__restrict__ oext, int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    if (i == n) { oext_off[oidx[n]] = (int64_t)owoff[n]; return; }
    if (i > n || !flag[i]) return;
    const Desc d = desc[i];
    const int64_t j = (int64_t)oidx[i];
    const int p = part_of(ps, P, i);
    // randomReflexivMarker starts at 2 in every task (1 once param.scramble == 3 in the k > 31 array loop,
    // P/ReflexivDSMain64.java:7484-7486) and toggles on every emission (:770, :1058-1062, :1242, :1514):
    // orientation = parity of the emission index
    // (several GPUs, rfx_shard.hip: a partition that began on an earlier rank brings the parity of its emissions there --
    // carry = {partition, parity})
    const int64_t cpar = (carry && p == carry[0]) ? (int64_t)carry[1] : 0;
    const int m = (d.type == 1 && d.b) ? (int)d.b : ((j - (int64_t)oidx[ps[p]] + cpar) & 1) ? 3 - start_marker : start_marker;
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
    if constexpr (KW == 2) if (L <= 31 && s.a.len <= 31 && (d.type != 2 || s.b.len <= 31)) {
        // The same for two-word keys (k = 33..63): S_out has at most 62 + 31 = 93 bases, a 192-bit value (h, m_, l) built
        // by appending pieces of at most 31 bases -- ~100 instructions instead of ~90 base lookups of ~20 each.
        const int res = sub - 31;                                    // bases in the key's second word (1..31)
        uint64_t h = 0, m_ = 0, l = 0;
        auto app = [&](uint64_t bits, int nb) __attribute__((always_inline)) {     // S = S << 2 nb | bits  (0 <= nb <= 31)
            const int sh = 2 * nb;
            if (sh) {
                h = (h << sh) | (m_ >> (64 - sh));
                m_ = (m_ << sh) | (l >> (64 - sh));
                l = (l << sh) | (bits & low_mask(nb));
            }
        };
        const int la = (int)s.a.len;
        if (s.a.marker == 1) { app(s.a.key.w[0], 31); app(s.a.key.w[1], res); app(s.a.w[0], la); }
        else { app(s.a.w[0], la); app(s.a.key.w[0], 31); app(s.a.key.w[1], res); }
        if (d.type == 2) app(s.b.w[0], (int)s.b.len);
        // 128 bits of S from bit `sh` up (sh <= 124)
        auto cut = [&](int sh, uint64_t *hi, uint64_t *lo) __attribute__((always_inline)) {
            uint64_t a2 = h, a1 = m_, a0 = l;
            if (sh >= 64) { a0 = a1; a1 = a2; a2 = 0; sh -= 64; }
            *lo = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;
            *hi = sh ? (a1 >> sh) | (a2 << (64 - sh)) : a1;
        };
        uint64_t khi, klo, eb;                                       // the key's `sub` bases as a 128-bit value, the extension
        if (m == 1) { cut(2 * (int)L, &khi, &klo); eb = l & low_mask((int)L); }
        else { uint64_t e1; cut(2 * sub, &e1, &eb); eb &= low_mask((int)L); khi = m_; klo = l; }
        // the key's words: the first 31 bases, then the remaining `res`
        KeyW<KW> kk;
        kk.w[KW - 1] = klo & low_mask(res);
        kk.w[0] = ((klo >> (2 * res)) | (khi << (64 - 2 * res))) & low_mask(31);
        okey[j] = kk;
        oext[wo] = (1ULL << (2 * (int)L)) | eb;
        return;
    }
    // key: first (m == 1) or last (m == 2) k-1 bases of S_out
    const int64_t kshift = m == 1 ? 0 : L;
    if (d.type == 1 && s.a.marker == m) okey[j] = s.a.key;
    else okey[j] = out_key<KW>(s, sub, kshift);
    const int64_t nw = (L + 30) / 31;
    if (nw > EMIT_SHORT) return;                     // k_emit_words: one thread per output word
    if (d.type == 1 && s.a.marker == m) {            // same orientation: the words are unchanged
        for (int64_t w = 0; w < nw; w++) oext[wo + w] = s.a.w[w];
        return;
    }
    emit_words<KW>(s, sub, m == 1 ? sub : 0, L, 0, nw, 1, oext + wo);
}

template <int KW>
__global__ void k_emit(const Desc *__restrict__ desc, const uint32_t *__restrict__ flag,
                       const uint64_t *__restrict__ oidx, const uint64_t *__restrict__ owoff, int64_t n,
                       const int64_t *__restrict__ ps, int P, int sub, int start_marker, const int32_t *__restrict__ carry,
                       const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                       const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                       const uint32_t *__restrict__ len,
                       KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker, int64_t *__restrict__ oext_off,
                       uint64_t *__restrict__ oext, int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    emit_at<KW>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, desc, flag, oidx, owoff, n, ps, P, sub, start_marker, carry, key, marker,
                ext_off, ext, len, okey, omarker, oext_off, oext, oleft, oright);
}

// Extensions longer than EMIT_SHORT words, one thread per OUTPUT WORD: thread t finds the emission
// that owns word t of the output array by a binary search in the word-offset scan (zero-length
// entries share their successor's offset, so the last entry with offset <= t is the owner) and
// writes that one word.  A 2.6 Mbp contig (84 K words) is 84 K threads; no per-record queues.
template <int KW>
__device__ __forceinline__ void emit_word_at(int64_t t, const Desc *__restrict__ desc, const uint64_t *__restrict__ oidx,
                             const uint64_t *__restrict__ owoff, int64_t n, const int64_t *__restrict__ ps, int P, int sub,
                             int start_marker, const int32_t *__restrict__ carry, const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                             const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                             const uint32_t *__restrict__ len, uint64_t *__restrict__ oext) {
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
    // (several GPUs, rfx_shard.hip: a partition that began on an earlier rank brings the parity of its emissions there --
    // carry = {partition, parity})
    const int64_t cpar = (carry && p == carry[0]) ? (int64_t)carry[1] : 0;
    const int m = (d.type == 1 && d.b) ? (int)d.b : ((j - (int64_t)oidx[ps[p]] + cpar) & 1) ? 3 - start_marker : start_marker;
    OutSeq<KW> s;
    s.type = (int)d.type;
    s.a = load_src<KW>(d.a, key, marker, ext_off, ext, len);
    s.lenSa = s.a.len + sub;
    if (d.type == 2) s.b = load_src<KW>(d.b, key, marker, ext_off, ext, len); else s.b = s.a;
    if (d.type == 1 && s.a.marker == m) oext[wo + w] = s.a.w[w];
    else emit_words<KW>(s, sub, m == 1 ? sub : 0, L, w, w + 1, 1, oext + wo);
}

template <int KW>
__global__ void k_emit_words(const Desc *__restrict__ desc, const uint64_t *__restrict__ oidx,
                             const uint64_t *__restrict__ owoff, int64_t n, const int64_t *__restrict__ ps, int P, int sub,
                             int start_marker, const int32_t *__restrict__ carry, const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                             const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext,
                             const uint32_t *__restrict__ len, uint64_t *__restrict__ oext) {
    emit_word_at<KW>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, desc, oidx, owoff, n, ps, P, sub, start_marker, carry, key, marker,
                     ext_off, ext, len, oext);
}


// ------------------------------------------------------------------------------------------------------------------
// Small passes.  Most passes of the loop touch a few thousand records or fewer (the record count
// shrinks by a quarter per pass; on a 4.6 Mbp genome 29 of 54 passes see <= 4096 records), and a pass built from ~35
// launches and one readback costs ~0.2 ms however small it is.  Below SP_N records a pass is TWO launches:
//   k_small_pass   ONE workgroup does everything that is control flow: the stop rule of the driver, the stable sort
//                  of the keys (LDS radix sort of (key word, position) pairs, word by word for multi-word keys), the
//                  logical partition starts, the run resolution, both scans, and the emission of keys, markers and the
//                  short extensions -- the same device functions as the big-pass kernels, fed through the sort's
//                  permutation instead of physically sorted records (a sorted VIEW of the fixed fields is written to
//                  scratch; extension words stay where they are);
//   k_small_words  the whole grid writes the long extensions (late passes: few records, several hundred thousand
//                  words), one thread per output word.
// The driver's state (iteration counter, last count, scramble, partition number, which of the two record sets is
// current, the trace) lives in HBM, so the host queues several passes back to back and reads the state once per batch;
// passes queued after the stop rule fired find `done` set and return at once.
constexpr int SP_N = 4096;
constexpr int SP_T = 1024;
constexpr int SP_W = SP_T / 64;
constexpr int SP_MAXP = 1024;           // logical partitions the single workgroup handles

struct SmallState {
    int64_t n, words, contig_number, nt;
    int32_t iterations, scramble, P, partition_number, cur, io, done, status, start_marker, pad;
};
struct SmallSet { void *key; int32_t *marker; int64_t *ext_off; uint64_t *ext; int32_t *left; int32_t *right; };
struct SmallScratch {
    void *skey; int32_t *smarker, *sleft, *sright; uint32_t *slen; int64_t *sext;
    Desc *desc; uint32_t *flag, *onw; uint64_t *oidx, *owoff; int64_t *ps; int *status;
};
struct SmallRule { int wide, coalesce, min_iter, max_iter, twin, sub, res_bits; int64_t trace_cap; };

template <int KW>
__global__ __launch_bounds__(SP_T) void k_small_pass(SmallSet A, SmallSet B, SmallScratch sc, SmallState *__restrict__ st,
                                                     SmallRule rule, int64_t *__restrict__ trace) {
    __shared__ uint64_t sk[2][SP_N];
    __shared__ uint16_t sv[2][SP_N];
    __shared__ volatile uint32_t wcnt[SP_W][256];
    __shared__ uint32_t wbase[SP_W][256];
    __shared__ uint32_t wsum[SP_W];
    __shared__ int sh_go, sh_P, sh_start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        // the driver's loop head: P/ReflexivMain.java:265-283 (k <= 31), P/ReflexivDSMain64.java:582, 621-655 (k > 31)
        int go = 0;
        if (!st->done) {
            if (st->iterations <= rule.max_iter) {
                st->iterations++;
                go = 1;
                const int64_t current = st->n;
                if (!rule.wide) {
                    if (st->iterations >= rule.min_iter && st->iterations % 3 == 0) {
                        if (st->contig_number == current) go = 0;
                        else {
                            st->contig_number = current;
                            if (rule.coalesce && st->partition_number >= 16 && current / st->partition_number <= 20) {
                                st->partition_number = st->partition_number / 4 + 1;
                                st->P = st->partition_number;
                            }
                        }
                    }
                } else if (st->iterations >= rule.min_iter + 3 && st->iterations % 3 == 0) {
                    if (st->contig_number == current) {
                        if (st->scramble == 2) st->scramble = 3; else go = 0;
                    } else st->contig_number = current;
                }
            }
            if (!go) st->done = 1;
        }
        if (go) { st->io = st->cur; st->start_marker = (rule.wide && st->scramble == 3) ? 1 : 2; }
        sh_go = go; sh_P = st->P; sh_start = st->start_marker;
    }
    __syncthreads();
    if (!sh_go) return;
    const int P = sh_P, start_marker = sh_start, sub = rule.sub;
    const int n = (int)st->n;
    const SmallSet &in = st->io ? B : A;
    const SmallSet &out = st->io ? A : B;
    const KeyW<KW> *ikey = (const KeyW<KW> *)in.key;
    KeyW<KW> *skey = (KeyW<KW> *)sc.skey;

    // ---- stable sort of the keys, last word first: (sk, sv) = (key word, source position)
    int cur = 0;
    const int per_wave = ((n + SP_T - 1) / SP_T) * 64;
    const int rounds = per_wave >> 6;                       // <= SP_N / SP_T = 4
    const uint64_t lt = (1ULL << lane) - 1;
    for (int w = KW - 1; w >= 0; w--) {
        for (int i = tid; i < n; i += SP_T) {
            const int src = (w == KW - 1) ? i : (int)sv[cur][i];
            sk[cur][i] = ikey[src].w[w];
            sv[cur][i] = (uint16_t)src;
        }
        const int bits = (w == KW - 1) ? rule.res_bits : 62;
        for (int sh = 0; sh < bits; sh += 8) {
            for (int i = tid; i < SP_W * 256; i += SP_T) ((volatile uint32_t *)wcnt)[i] = 0;
            __syncthreads();
            uint32_t rank[SP_N / SP_T];
            const int cbase = wave * per_wave;
#pragma unroll
            for (int r = 0; r < SP_N / SP_T; r++) {
                if (r < rounds) {
                    const int idx = cbase + r * 64 + lane;
                    const bool ok = idx < n;
                    const unsigned d = ok ? (unsigned)(sk[cur][idx] >> sh) & 255u : 0u;
                    uint64_t peers = __ballot(ok);
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const uint64_t m = __ballot((d >> b) & 1u);
                        peers &= ((d >> b) & 1u) ? m : ~m;
                    }
                    uint32_t before = 0;
                    if (ok) { before = wcnt[wave][d]; rank[r] = before + (uint32_t)__popcll(peers & lt); }
                    if (ok && (peers & lt) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
                }
            }
            __syncthreads();
            uint32_t tot = 0;
            if (tid < 256) {
#pragma unroll
                for (int q = 0; q < SP_W; q++) tot += wcnt[q][tid];
            }
            uint32_t dbase = block_exclusive_scan(tot, wsum, nullptr);
            if (tid < 256) {
#pragma unroll
                for (int q = 0; q < SP_W; q++) { wbase[q][tid] = dbase; dbase += wcnt[q][tid]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < SP_N / SP_T; r++) {
                if (r < rounds) {
                    const int idx = cbase + r * 64 + lane;
                    if (idx < n) {
                        const uint64_t kk = sk[cur][idx];
                        const uint32_t dst = wbase[wave][(unsigned)(kk >> sh) & 255u] + rank[r];
                        sk[cur ^ 1][dst] = kk;
                        sv[cur ^ 1][dst] = sv[cur][idx];
                    }
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---- sorted view of the fixed fields (extension words are read where they lie)
    for (int i = tid; i < n; i += SP_T) {
        const int src = (int)sv[cur][i];
        skey[i] = ikey[src];
        sc.smarker[i] = in.marker[src]; sc.sleft[i] = in.left[src]; sc.sright[i] = in.right[src];
        const int64_t b = in.ext_off[src], nw = in.ext_off[src + 1] - b;
        sc.slen[i] = (uint32_t)((nw - 1) * 31 + sentinel_len(in.ext[b]));
        sc.sext[i] = b;
    }
    if (tid == 0) sc.status[0] = 0;
    __syncthreads();
    // ---- logical partition starts (order contract B.0)
    for (int p = tid; p <= P; p += SP_T) {
        int64_t s = p == P ? n : (int64_t)(((uint64_t)p * (uint64_t)n) / (uint64_t)P);
        while (p < P && s > 0 && s < n && key_eq(skey[s], skey[s - 1])) s++;
        sc.ps[p] = s;
    }
    // ---- run resolution
    for (int i = tid; i < n; i += SP_T)
        resolve_at<KW>(i, (const KeyW<KW> *)skey, sc.smarker, sc.sleft, sc.sright, sc.slen, n, rule.twin, 2, sc.desc, sc.flag, sc.onw,
                       sc.status);
    __syncthreads();
    // ---- both scans: every thread owns SP_N / SP_T consecutive entries
    {
        constexpr int E = SP_N / SP_T;
        uint32_t f[E], w[E], fs = 0, ws = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int i = tid * E + e;
            f[e] = i < n ? sc.flag[i] : 0u; w[e] = i < n ? sc.onw[i] : 0u;
            fs += f[e]; ws += w[e];
        }
        uint32_t ftot = 0, wtot = 0;
        uint32_t fb = block_exclusive_scan(fs, wsum, &ftot);
        uint32_t wb = block_exclusive_scan(ws, wsum, &wtot);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int i = tid * E + e;
            if (i < n) { sc.oidx[i] = fb; sc.owoff[i] = wb; }
            fb += f[e]; wb += w[e];
        }
        if (tid == 0) { sc.oidx[n] = ftot; sc.owoff[n] = wtot; }
    }
    __syncthreads();
    // ---- keys, markers, left / right, offsets and the short extensions
    for (int i = tid; i <= n; i += SP_T)
        emit_at<KW>(i, sc.desc, sc.flag, sc.oidx, sc.owoff, n, sc.ps, P, sub, start_marker, nullptr, (const KeyW<KW> *)skey, sc.smarker, sc.sext,
                    in.ext, sc.slen, (KeyW<KW> *)out.key, out.marker, out.ext_off, out.ext, out.left, out.right);
    __syncthreads();
    if (tid == 0) {
        const int64_t m = (int64_t)sc.oidx[n];
        st->n = m; st->words = (int64_t)sc.owoff[n];
        st->cur = 1 - st->io;
        st->status |= sc.status[0];
        if (trace && st->nt < rule.trace_cap) trace[st->nt] = m;
        st->nt++;
    }
}

// the long extensions of the pass k_small_pass just resolved (its scans are still in the scratch arrays)
template <int KW>
__global__ void k_small_words(SmallSet A, SmallSet B, SmallScratch sc, const SmallState *__restrict__ st, int sub, int64_t n_in_of_pass_unused) {
    (void)n_in_of_pass_unused;
    if (st->done) return;
    // st->cur was flipped at the end of k_small_pass: this pass read set `io` and wrote the other one
    const SmallSet &in = st->io ? B : A;
    const SmallSet &out = st->io ? A : B;
    // the scans' last entries hold this pass's record count / word total; n_in = number of descriptors = ps[P]
    const int P = st->P;
    const int64_t n_in = sc.ps[P];
    emit_word_at<KW>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, sc.desc, sc.oidx, sc.owoff, n_in, sc.ps, P, sub, st->start_marker, nullptr,
                     (const KeyW<KW> *)sc.skey, sc.smarker, sc.sext, in.ext, sc.slen, out.ext);
}


// ---- the k > 31 from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712): operators that copy or
// re-orient records, expressed as emission descriptors for the same emit kernels as the extend pass
// DSReflexivAndForwardKmer :2153-2168: descriptor 2i = record i as it is, 2i+1 = its other orientation
__global__ void k_desc_double(const int32_t *__restrict__ marker, const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                              const uint32_t *__restrict__ len, int64_t n, Desc *__restrict__ desc, uint32_t *__restrict__ flag,
                              uint32_t *__restrict__ onw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int mk = marker[i];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        Desc d; d.a = (uint32_t)i; d.b = (uint32_t)(t == 0 ? mk : 3 - mk); d.left = left[i]; d.right = right[i]; d.type = 1u; d.len = len[i];
        desc[2 * i + t] = d; flag[2 * i + t] = 1u; onw[2 * i + t] = (len[i] + 30u) / 31u;
    }
}
// DSFilterUnExtendableKmerLeftEnds (m = 1, :3424-3444) / ...RightEnds (m = 2, :4381-4401): every record in orientation m
__global__ void k_desc_flip_all(const int32_t *__restrict__ left, const int32_t *__restrict__ right, const uint32_t *__restrict__ len,
                                int64_t n, int m, Desc *__restrict__ desc, uint32_t *__restrict__ flag, uint32_t *__restrict__ onw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Desc d; d.a = (uint32_t)i; d.b = (uint32_t)m; d.left = left[i]; d.right = right[i]; d.type = 1u; d.len = len[i];
    desc[i] = d; flag[i] = 1u; onw[i] = (len[i] + 30u) / 31u;
}

// The four run filters (op 1 DSFilterExtendableKmerPairs :5305-6375, 2 DSFilterUnExtendableKmer :6377-7444,
// 3 DSFilterStillExtendableKmerFromPairs :3228-3390, 4 DSFilterStillExtendableKmerEnds :3044-3226) on records sorted by key:
// the reference walks a task with a one-record holder that never survives a key change except to be emitted (ops 2-4) or
// dropped (op 1; only a task's LAST holder is emitted there, :5462-5466), so every equal-key run resolves on its own; the
// thread that owns a run head walks it and writes the run's emissions into the run's own index range.
template <int KW>
__global__ void k_key_filter(int op, const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                             const int32_t *__restrict__ left, const int32_t *__restrict__ right, const uint32_t *__restrict__ len,
                             const uint64_t *__restrict__ word0, int64_t n, const int64_t *__restrict__ ps, int P,
                             Desc *__restrict__ desc, uint32_t *__restrict__ flag, uint32_t *__restrict__ onw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KeyW<KW> kk = key[i];
    if (i > 0 && key_eq(key[i - 1], kk)) return;               // not a run head
    int64_t o = i, holder = i, s = i + 1;
#define PUTF(SRC, M) do { \
        Desc d_; d_.a = (uint32_t)(SRC); d_.b = (uint32_t)(M); d_.left = left[SRC]; d_.right = right[SRC]; d_.type = 1u; \
        d_.len = len[SRC]; desc[o] = d_; flag[o] = 1u; onw[o] = (len[SRC] + 30u) / 31u; o++; } while (0)
    for (; s < n && key_eq(key[s], kk); s++) {
        if (holder < 0) { holder = s; continue; }
        const int64_t h = holder;
        if (op == 3) { PUTF(h, marker[h]); holder = -1; continue; }                         // :3289-3312
        if (op == 4) {                                                                      // :3103-3145
            const int64_t sh = (int64_t)len[h] * 31 + sentinel_len(word0[h]), ss = (int64_t)len[s] * 31 + sentinel_len(word0[s]);
            if (sh >= ss) PUTF(h, marker[h]); else PUTF(s, marker[s]);
            holder = -1; continue;
        }
        bool mergeable = false;
        if (marker[s] != marker[h]) {                                                       // :5376-5399 / :6448-6467
            const int32_t a = marker[s] == 1 ? left[s] : right[s], b = marker[s] == 1 ? right[h] : left[h];
            mergeable = (a < 0 && b < 0) || (a >= 0 && b >= 0) || (a >= 0 && a - (int64_t)len[h] >= 0) || (b >= 0 && b - (int64_t)len[s] >= 0);
        }
        if (op == 1) {
            if (mergeable) {
                if (marker[s] == 1) { PUTF(s, 1); PUTF(h, 1); }                             // :5377-5379
                else { PUTF(h, 1); PUTF(s, 1); }                                            // :5416-5418
                holder = -1;
            } else holder = s;
        } else {                                                                            // op 2
            if (mergeable) { holder = -1; continue; }
            PUTF(h, marker[h] == 2 ? 1 : marker[h]);                                        // :6469, :6482 / :6475, :6503
            holder = s;
        }
    }
    if (holder >= 0) {
        // the run ends: a key change emits the holder (ops 2-4) or drops it (op 1), the end of the task always emits it
        bool emit = op != 1;
        if (!emit) { const int p = part_of(ps, P, i); emit = s == ps[p + 1]; }
        if (emit) PUTF(holder, marker[holder]);
    }
    for (; o < s; o++) { flag[o] = 0u; onw[o] = 0u; desc[o].type = 0u; }
#undef PUTF
}

__global__ void k_word0(const int64_t *__restrict__ ext_off, const uint64_t *__restrict__ ext, int64_t n, uint64_t *__restrict__ w0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w0[i] = ext[ext_off[i]];
}

// output partition starts + the pass summary the host reads back in ONE copy:
// summary = {records out, words out, status}
__global__ void k_out_part_start(const int64_t *__restrict__ ps, int P, const uint64_t *__restrict__ oidx,
                                 int64_t *__restrict__ ops, const uint64_t *__restrict__ owoff, int64_t n,
                                 const int *__restrict__ status, uint64_t *__restrict__ summary, volatile uint64_t *mbox, uint64_t seq) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= P) ops[p] = (int64_t)oidx[ps[p]];
    if (p == 0) { summary[0] = oidx[n]; summary[1] = owoff[n]; summary[2] = (uint64_t)(unsigned)status[0]; }
    if (mbox) {                                            // (one workgroup: the host launches it so when it wants the mailbox)
        __syncthreads();
        if (threadIdx.x == 0) {
            mbox[1] = oidx[n]; mbox[2] = owoff[n]; mbox[3] = (uint64_t)(unsigned)status[0];
            __threadfence_system();
            mbox[0] = seq;
        }
    }
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

static DevBuf &ops_tmp_alloc(rfx_ctx *ctx, DevBuf &b, int P) { (void)b.alloc((size_t)(P + 1) * 8, ctx->stream); return b; }

// everything after the descriptors: scans, emission, output partition starts, the 24-byte summary.
// nd = number of descriptor slots (= n of the input except for the doubling operator); single_word: one word per emission.
static int desc_tail(rfx_ctx *ctx, const DevRecords &in, int64_t nd, DevBuf &len, DevBuf &desc, DevBuf &flag, DevBuf &onw, DevBuf &status,
                     const int64_t *d_part_start, int P, int k, bool single_word, int start_marker, int64_t words_bound,
                     DevRecords &out, DevBuf &out_part_start, PartCarry *carry_hook = nullptr) {
    const int sub = k - 1, kw = in.kw;
    DevBuf oidx, owoff;
    RFX_HIP(oidx.alloc((size_t)(nd + 1) * 8, ctx->stream));
    RFX_HIP(owoff.alloc((size_t)(nd + 1) * 8, ctx->stream));
    // single-word stage: every emission has exactly one word (a longer one is the RFX_E_STATE below), so the word
    // offsets ARE the emission indices
    if (single_word) {
        RFX_TRY(exclusive_scan_u32_to_u64(ctx, flag.as<uint32_t>(), oidx.as<uint64_t>(), nd));
        RFX_HIP(hipMemcpyAsync(owoff.p, oidx.p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        RFX_TRY(exclusive_scan2_u32_to_u64(ctx, flag.as<uint32_t>(), onw.as<uint32_t>(), oidx.as<uint64_t>(), owoff.as<uint64_t>(), nd));
    }
    // several GPUs: the emissions of this rank's first partition on earlier ranks (their parity), agreed between the scan
    // and the emission -- on the device, no host wait (rfx_shard.hip)
    const int32_t *d_carry = nullptr;
    if (carry_hook) RFX_TRY(carry_hook->compute(ctx, d_part_start, P, oidx.as<uint64_t>(), nd, &d_carry));
    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_emit<KW>, dim3(grid_for(nd + 1)), dim3(256), 0, ctx->stream, (const Desc *)desc.as<Desc>(),
                       (const uint32_t *)flag.as<uint32_t>(), (const uint64_t *)oidx.as<uint64_t>(),
                       (const uint64_t *)owoff.as<uint64_t>(), nd, d_part_start, P, sub, start_marker, d_carry,
                       (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const int64_t *)in.ext_off.as<int64_t>(), (const uint64_t *)in.ext.as<uint64_t>(),
                       (const uint32_t *)len.as<uint32_t>(), out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(),
                       out.ext_off.as<int64_t>(), out.ext.as<uint64_t>(), out.left.as<int32_t>(),
                       out.right.as<int32_t>()));
    RFX_HIP(hipGetLastError());
    if (words_bound > nd || in.words > in.n) {          // some record may have more than one word
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_emit_words<KW>, dim3(grid_for(words_bound)), dim3(256), 0, ctx->stream, (const Desc *)desc.as<Desc>(),
                           (const uint64_t *)oidx.as<uint64_t>(), (const uint64_t *)owoff.as<uint64_t>(), nd, d_part_start, P,
                           sub, start_marker, d_carry, (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                           (const int64_t *)in.ext_off.as<int64_t>(), (const uint64_t *)in.ext.as<uint64_t>(),
                           (const uint32_t *)len.as<uint32_t>(), out.ext.as<uint64_t>()));
        RFX_HIP(hipGetLastError());
    }
    DevBuf summary;
    RFX_HIP(summary.alloc(24, ctx->stream));
    const uint64_t seq = P + 1 <= 256 ? mailbox_next(ctx) : 0;          // (the posting kernel is one workgroup)
    hipLaunchKernelGGL(k_out_part_start, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream, d_part_start, P,
                       (const uint64_t *)oidx.as<uint64_t>(), out_part_start.as<int64_t>(),
                       (const uint64_t *)owoff.as<uint64_t>(), nd, (const int *)status.as<int>(), summary.as<uint64_t>(),
                       seq ? ctx->mailbox : (volatile uint64_t *)nullptr, seq);
    RFX_HIP(hipGetLastError());
    uint64_t tot[3] = {0, 0, 0};
    if (seq) RFX_TRY(mailbox_wait(ctx, seq, tot, 3));
    else {
        RFX_HIP(hipMemcpyAsync(tot, summary.p, 24, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    const int st = (int)tot[2];
    out.n = (int64_t)tot[0]; out.words = (int64_t)tot[1];
    if (st) { ctx->last_error = "extend pass: single-word stage produced an extension > 31 bases"; return RFX_E_STATE; }
    return RFX_OK;
}

int extend_pass(rfx_ctx *ctx, const DevRecords &in, const int64_t *d_part_start, int P, int k, int twin,
                int stage, DevRecords &out, DevBuf &out_part_start, int start_marker, PartCarry *carry_hook) {
    const int64_t n = in.n;
    const int kw = in.kw;
    if (n > (int64_t)0xFFFFFFFFLL) return RFX_E_LIMIT;
    if (kw != sub_words(k) || (start_marker != 1 && start_marker != 2)) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, in.words, kw));
    RFX_HIP(out_part_start.alloc((size_t)(P + 1) * 8, ctx->stream));
    DevBuf len, desc, flag, onw, status;
    const int64_t a = n ? n : 1;
    RFX_HIP(len.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(desc.alloc((size_t)a * sizeof(Desc), ctx->stream));
    RFX_HIP(flag.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(onw.alloc((size_t)a * 4, ctx->stream));
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
    return desc_tail(ctx, in, n, len, desc, flag, onw, status, d_part_start, P, k, stage == 0, start_marker, in.words, out, out_part_start,
                     carry_hook);
}

// op 0: DSReflexivAndForwardKmer (2n out); 1..4: the run filters (see k_key_filter); 5 / 6: every record forward / reflected
int extras_operator(rfx_ctx *ctx, int op, const DevRecords &in, const int64_t *d_part_start, int P, int k, DevRecords &out,
                    DevBuf &out_part_start) {
    const int64_t n = in.n;
    const int kw = in.kw;
    if (2 * n > (int64_t)0xFFFFFFFFLL) return RFX_E_LIMIT;
    if (kw != sub_words(k) || op < 0 || op > 6) return RFX_E_ARG;
    const int64_t nd = op == 0 ? 2 * n : n, wcap = op == 0 ? 2 * in.words : in.words;
    RFX_TRY(dev_records_alloc(ctx, out, nd, wcap, kw));
    RFX_HIP(out_part_start.alloc((size_t)(P + 1) * 8, ctx->stream));
    DevBuf len, desc, flag, onw, status, w0, ps1;
    const int64_t a = nd ? nd : 1;
    RFX_HIP(len.alloc((size_t)(n ? n : 1) * 4, ctx->stream));
    RFX_HIP(desc.alloc((size_t)a * sizeof(Desc), ctx->stream));
    RFX_HIP(flag.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(onw.alloc((size_t)a * 4, ctx->stream));
    RFX_HIP(status.alloc(4, ctx->stream));
    RFX_HIP(hipMemsetAsync(status.p, 0, 4, ctx->stream));
    const int64_t *ps = d_part_start;
    int Pn = P;
    if (op == 0) {
        // the doubled set keeps the input's partitions: partition p holds the descriptors [2 ps[p], 2 ps[p+1]); nothing in
        // this operator looks at them (every descriptor names its orientation), so one partition over everything will do
        int64_t one[2] = {0, nd};
        RFX_HIP(ps1.alloc(16, ctx->stream));
        RFX_HIP(hipMemcpyAsync(ps1.p, one, 16, hipMemcpyHostToDevice, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        ps = ps1.as<int64_t>(); Pn = 1;
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_ext_len, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const int64_t *)in.ext_off.as<int64_t>(),
                           (const uint64_t *)in.ext.as<uint64_t>(), n, len.as<uint32_t>());
        RFX_HIP(hipGetLastError());
        if (op == 0) {
            hipLaunchKernelGGL(k_desc_double, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const int32_t *)in.marker.as<int32_t>(),
                               (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(), (const uint32_t *)len.as<uint32_t>(),
                               n, desc.as<Desc>(), flag.as<uint32_t>(), onw.as<uint32_t>());
        } else if (op >= 5) {
            hipLaunchKernelGGL(k_desc_flip_all, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const int32_t *)in.left.as<int32_t>(),
                               (const int32_t *)in.right.as<int32_t>(), (const uint32_t *)len.as<uint32_t>(), n, op == 5 ? 1 : 2,
                               desc.as<Desc>(), flag.as<uint32_t>(), onw.as<uint32_t>());
        } else {
            RFX_HIP(w0.alloc((size_t)n * 8, ctx->stream));
            hipLaunchKernelGGL(k_word0, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const int64_t *)in.ext_off.as<int64_t>(),
                               (const uint64_t *)in.ext.as<uint64_t>(), n, w0.as<uint64_t>());
            RFX_HIP(hipGetLastError());
            RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_key_filter<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, op,
                               (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                               (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(),
                               (const uint32_t *)len.as<uint32_t>(), (const uint64_t *)w0.as<uint64_t>(), n, d_part_start, P,
                               desc.as<Desc>(), flag.as<uint32_t>(), onw.as<uint32_t>()));
        }
        RFX_HIP(hipGetLastError());
    }
    DevBuf ops_tmp;
    RFX_TRY(desc_tail(ctx, in, nd, len, desc, flag, onw, status, ps, Pn, k, false, 2, wcap, out, op == 0 ? ops_tmp_alloc(ctx, ops_tmp, Pn) : out_part_start));
    if (op == 0) {
        // output partition starts of the doubled set: twice the input's
        std::vector<int64_t> h((size_t)P + 1);
        RFX_HIP(hipMemcpyAsync(h.data(), d_part_start, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        for (auto &x : h) x *= 2;
        RFX_HIP(hipMemcpyAsync(out_part_start.p, h.data(), (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    return RFX_OK;
}

// Runs the rest of the driver's loop (P/ReflexivMain.java:265-296; k > 31: P/ReflexivDSMain64.java:582-670) on a record set of
// at most SP_N records: see k_small_pass.  io: iterations / contig_number / scramble / P / partition_number as the host
// loop left them; on return they are what the loop would have left, `recs` holds the surviving records and the trace has
// grown by the passes run.
int small_passes(rfx_ctx *ctx, DevRecords &recs, int k, int twin, bool wide, int coalesce, int min_iter, int max_iter,
                 int *iterations, int64_t *contig_number, int *scramble, int *P, int *partition_number,
                 int64_t *trace, int64_t trace_cap, int64_t *nt) {
    const int kw = recs.kw, sub = k - 1;
    if (recs.n > SP_N || *P > SP_MAXP || *P < 1) return RFX_E_ARG;
    const int64_t capn = recs.n > 0 ? recs.n : 1, capw = recs.words > 0 ? recs.words : 1;
    DevRecords other;
    RFX_TRY(dev_records_alloc(ctx, other, capn, capw, kw));
    DevBuf skey, smarker, sleft, sright, slen, sext, desc, flag, onw, oidx, owoff, ps, status, state, dtrace;
    RFX_HIP(skey.alloc((size_t)capn * 8 * kw, ctx->stream));
    RFX_HIP(smarker.alloc((size_t)capn * 4, ctx->stream)); RFX_HIP(sleft.alloc((size_t)capn * 4, ctx->stream));
    RFX_HIP(sright.alloc((size_t)capn * 4, ctx->stream)); RFX_HIP(slen.alloc((size_t)capn * 4, ctx->stream));
    RFX_HIP(sext.alloc((size_t)capn * 8, ctx->stream)); RFX_HIP(desc.alloc((size_t)capn * sizeof(Desc), ctx->stream));
    RFX_HIP(flag.alloc((size_t)capn * 4, ctx->stream)); RFX_HIP(onw.alloc((size_t)capn * 4, ctx->stream));
    RFX_HIP(oidx.alloc((size_t)(capn + 1) * 8, ctx->stream)); RFX_HIP(owoff.alloc((size_t)(capn + 1) * 8, ctx->stream));
    RFX_HIP(ps.alloc((size_t)(SP_MAXP + 1) * 8, ctx->stream)); RFX_HIP(status.alloc(4, ctx->stream));
    RFX_HIP(state.alloc(sizeof(SmallState), ctx->stream));
    const int64_t tcap = max_iter + 8;
    RFX_HIP(dtrace.alloc((size_t)tcap * 8, ctx->stream));
    SmallState h{};
    h.n = recs.n; h.words = recs.words; h.contig_number = *contig_number; h.nt = 0;
    h.iterations = *iterations; h.scramble = *scramble; h.P = *P; h.partition_number = *partition_number;
    h.cur = 0; h.io = 0; h.done = 0; h.status = 0; h.start_marker = 2;
    RFX_HIP(hipMemcpyAsync(state.p, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    const SmallSet A{recs.key.p, recs.marker.as<int32_t>(), recs.ext_off.as<int64_t>(), recs.ext.as<uint64_t>(),
                     recs.left.as<int32_t>(), recs.right.as<int32_t>()};
    const SmallSet B{other.key.p, other.marker.as<int32_t>(), other.ext_off.as<int64_t>(), other.ext.as<uint64_t>(),
                     other.left.as<int32_t>(), other.right.as<int32_t>()};
    const SmallScratch sc{skey.p, smarker.as<int32_t>(), sleft.as<int32_t>(), sright.as<int32_t>(), slen.as<uint32_t>(),
                          sext.as<int64_t>(), desc.as<Desc>(), flag.as<uint32_t>(), onw.as<uint32_t>(), oidx.as<uint64_t>(),
                          owoff.as<uint64_t>(), ps.as<int64_t>(), status.as<int>()};
    const int res = sub - 31 * (kw - 1);
    const SmallRule rule{wide ? 1 : 0, coalesce, min_iter, max_iter, twin, sub, 2 * res, tcap};
    constexpr int BATCH = 6;                 // passes queued between two looks at the state (two checks of the stop rule)
    int64_t words_bound = recs.words;
    for (;;) {
        for (int b = 0; b < BATCH; b++) {
            RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_small_pass<KW>, dim3(1), dim3(SP_T), 0, ctx->stream, A, B, sc, state.as<SmallState>(), rule,
                                                 dtrace.as<int64_t>()));
            RFX_HIP(hipGetLastError());
            if (words_bound > 0) {
                RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_small_words<KW>, dim3(grid_for(words_bound)), dim3(256), 0, ctx->stream, A, B, sc,
                                                     (const SmallState *)state.as<SmallState>(), sub, (int64_t)0));
                RFX_HIP(hipGetLastError());
            }
        }
        RFX_HIP(hipMemcpyAsync(&h, state.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        words_bound = h.words;                // the word total never grows from pass to pass
        if (h.done || h.iterations > max_iter) break;
    }
    if (h.status) { ctx->last_error = "extend pass: impossible record state"; return RFX_E_STATE; }
    std::vector<int64_t> ht((size_t)(h.nt > 0 ? h.nt : 1));
    if (h.nt > 0) {
        RFX_HIP(hipMemcpyAsync(ht.data(), dtrace.p, (size_t)h.nt * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFX_TRY(sync_checked(ctx));
    }
    for (int64_t i = 0; i < h.nt; i++) { if (trace && *nt < trace_cap) trace[*nt] = ht[(size_t)i]; (*nt)++; }
    if (h.cur == 1) {        // the survivors are in the second set: hand its buffers over
        std::swap(recs.key.p, other.key.p); std::swap(recs.key.borrowed, other.key.borrowed); std::swap(recs.key.s, other.key.s);
        std::swap(recs.marker.p, other.marker.p); std::swap(recs.marker.borrowed, other.marker.borrowed); std::swap(recs.marker.s, other.marker.s);
        std::swap(recs.ext_off.p, other.ext_off.p); std::swap(recs.ext_off.borrowed, other.ext_off.borrowed); std::swap(recs.ext_off.s, other.ext_off.s);
        std::swap(recs.ext.p, other.ext.p); std::swap(recs.ext.borrowed, other.ext.borrowed); std::swap(recs.ext.s, other.ext.s);
        std::swap(recs.left.p, other.left.p); std::swap(recs.left.borrowed, other.left.borrowed); std::swap(recs.left.s, other.left.s);
        std::swap(recs.right.p, other.right.p); std::swap(recs.right.borrowed, other.right.borrowed); std::swap(recs.right.s, other.right.s);
    }
    recs.n = h.n; recs.words = h.words;
    *iterations = h.iterations; *contig_number = h.contig_number; *scramble = h.scramble; *P = h.P;
    *partition_number = h.partition_number;
    return RFX_OK;
}

int small_pass_limit() { return SP_N; }
int small_pass_max_partitions() { return SP_MAXP; }

}  // namespace rfx
