// rfx_dynamic.hip -- the dynamic-k record format and passes on the GPU:
// P/ReflexivDSDynamicKmerFirstFour.java (DynamicKmerBinarizerFromReducedToSubKmer :2931-3016, DSkmerRandomReflection
// :2509-2762, DSExtendReflexivKmer :1581-2373, DSBinarySubKmerWithShortExtensionToString :226-263) and
// P/ReflexivDSDynamicKmerIteration.java (its binarizer, DSExtendReflexivKmerToArrayLoop :465-1249, ...LongExtensionToString).
//
// The third record layout -- keys of ANY length as left-aligned 31-base blocks with a 01 terminator, an attribute long
// (marker << 62 | left << 32 | right, negatives as 30000 - v), extensions in the same left-aligned form -- is what a Row
// carries; in HBM a record set is base strings (one byte per base, offsets) + marker / left / right, and blocks are formed
// where the reference's behaviour depends on them: the ORDER of sort("k-1") (array<long>: element by element as signed
// longs, a proper prefix first).
//
// One pass = the scan of SURVEY.md B.5 with a one-row holder, except that a row meets the holder when their keys are EQUAL
// OR ONE IS A PREFIX OF THE OTHER (dynamicSubKmerComparator), so equal-key runs are no longer the unit of work.  What is:
// a FAMILY -- a maximal run of sorted rows that agree on their first Lmin bases, Lmin = the shortest key of the set.  A row
// can only be related to an earlier row through a common prefix of at least Lmin bases, and everything between two related
// rows shares that prefix, so no holder ever survives a family boundary.  Kernels: k_dyn_blocks (the sort keys),
// the library's stable radix sort (block count, then block 3 .. block 0), k_dyn_heads, k_dyn_walk<COUNT / WRITE> (the
// owner of a family head walks it with the reference's rules; the decisions do not depend on the toggling orientation,
// which is the parity of the emission rank inside the partition: a prefix sum, as on the fixed-k path), k_dyn_sizes, and
// k_dyn_emit (one thread per output record writes its bases).
#include <algorithm>
#include <string>
#include <vector>
#include "rfx_internal.h"

using namespace rfx;

namespace {

constexpr int DYN_MAXB = 4;            // blocks per key: keys up to 124 bases (the reference's k-mer list ends at 95)

struct DynDev {                       // a record set in HBM
    int64_t n = 0, nk = 0, ne = 0;    // records, key bases, extension bases
    DevBuf key, key_off, ext, ext_off, marker, left, right;
};

static void buf_swap(DevBuf &x, DevBuf &y) { std::swap(x.p, y.p); std::swap(x.s, y.s); std::swap(x.borrowed, y.borrowed); }
static void dyn_swap(DynDev &x, DynDev &y) {
    std::swap(x.n, y.n); std::swap(x.nk, y.nk); std::swap(x.ne, y.ne);
    buf_swap(x.key, y.key); buf_swap(x.key_off, y.key_off); buf_swap(x.ext, y.ext); buf_swap(x.ext_off, y.ext_off);
    buf_swap(x.marker, y.marker); buf_swap(x.left, y.left); buf_swap(x.right, y.right);
}

static int dyn_alloc(rfx_ctx *ctx, DynDev &d, int64_t n, int64_t nk, int64_t ne) {
    RFX_HIP(d.key.alloc((size_t)std::max<int64_t>(nk, 1), ctx->stream));
    RFX_HIP(d.ext.alloc((size_t)std::max<int64_t>(ne, 1), ctx->stream));
    RFX_HIP(d.key_off.alloc((size_t)(n + 1) * 8, ctx->stream));
    RFX_HIP(d.ext_off.alloc((size_t)(n + 1) * 8, ctx->stream));
    RFX_HIP(d.marker.alloc((size_t)std::max<int64_t>(n, 1) * 4, ctx->stream));
    RFX_HIP(d.left.alloc((size_t)std::max<int64_t>(n, 1) * 4, ctx->stream));
    RFX_HIP(d.right.alloc((size_t)std::max<int64_t>(n, 1) * 4, ctx->stream));
    d.n = n; d.nk = nk; d.ne = ne;
    return RFX_OK;
}

__device__ __forceinline__ uint64_t dyn_block(const uint8_t *s, int n, int j) {
    uint64_t x = 0;
    const int b0 = 31 * j;
    int m = n - b0;
    if (m > 31) m = 31;
    for (int i = 0; i < m; i++) x |= (uint64_t)s[b0 + i] << (2 * (31 - i));
    if (b0 + 31 >= n) x |= 1ULL << (2 * (31 - m));
    return x;
}

// sort keys: block j with the sign bit flipped (Spark compares signed longs), 0 past the last block; the block count
__global__ __launch_bounds__(256) void k_dyn_blocks(const uint8_t *__restrict__ key, const int64_t *__restrict__ off, int64_t n,
                                                    uint64_t *__restrict__ blk /* [DYN_MAXB][n] */, uint64_t *__restrict__ nblk,
                                                    uint32_t *__restrict__ perm, int *__restrict__ too_long, uint32_t *__restrict__ min_len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len = (int)(off[i + 1] - off[i]);
    const int nb = len <= 0 ? 1 : (len - 1) / 31 + 1;
    if (nb > DYN_MAXB) { *too_long = 1; }
    const uint8_t *s = key + off[i];
    for (int j = 0; j < DYN_MAXB; j++) blk[(int64_t)j * n + i] = j < nb ? (dyn_block(s, len, j) ^ 0x8000000000000000ull) : 0ull;
    nblk[i] = (uint64_t)nb;
    perm[i] = (uint32_t)i;
    atomicMin(min_len, (uint32_t)len);
}
__global__ __launch_bounds__(256) void k_dyn_gather_u64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, int64_t n,
                                                        uint64_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// record sizes through a permutation (for the scans of the gather)
__global__ __launch_bounds__(256) void k_dyn_perm_sizes(const int64_t *__restrict__ koff, const int64_t *__restrict__ eoff,
                                                        const uint32_t *__restrict__ perm, int64_t n, uint64_t *__restrict__ ks,
                                                        uint64_t *__restrict__ es) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    ks[i] = (uint64_t)(koff[p + 1] - koff[p]);
    es[i] = (uint64_t)(eoff[p + 1] - eoff[p]);
}
__global__ __launch_bounds__(256) void k_dyn_gather(const uint8_t *__restrict__ key, const int64_t *__restrict__ koff,
                                                    const uint8_t *__restrict__ ext, const int64_t *__restrict__ eoff,
                                                    const int32_t *__restrict__ marker, const int32_t *__restrict__ left,
                                                    const int32_t *__restrict__ right, const uint32_t *__restrict__ perm, int64_t n,
                                                    const uint64_t *__restrict__ nko, const uint64_t *__restrict__ neo, uint8_t *__restrict__ okey,
                                                    int64_t *__restrict__ okoff, uint8_t *__restrict__ oext, int64_t *__restrict__ oeoff,
                                                    int32_t *__restrict__ omarker, int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { okoff[n] = (int64_t)nko[n]; oeoff[n] = (int64_t)neo[n]; return; }
    const uint32_t p = perm[i];
    const int64_t kb = koff[p], kn = koff[p + 1] - kb, eb = eoff[p], en = eoff[p + 1] - eb;
    const int64_t ko = (int64_t)nko[i], eo = (int64_t)neo[i];
    okoff[i] = ko; oeoff[i] = eo;
    for (int64_t j = 0; j < kn; j++) okey[ko + j] = key[kb + j];
    for (int64_t j = 0; j < en; j++) oext[eo + j] = ext[eb + j];
    omarker[i] = marker[p]; oleft[i] = left[p]; oright[i] = right[p];
}

__device__ __forceinline__ bool dyn_keys_equal(const uint8_t *key, const int64_t *off, int64_t a, int64_t b) {
    const int64_t la = off[a + 1] - off[a], lb = off[b + 1] - off[b];
    if (la != lb) return false;
    const uint8_t *x = key + off[a], *y = key + off[b];
    for (int64_t j = 0; j < la; j++) if (x[j] != y[j]) return false;
    return true;
}
// logical partition p starts at floor(p*n/P), moved forward past equal keys (the order contract)
__global__ void k_dyn_part_starts(const uint8_t *__restrict__ key, const int64_t *__restrict__ off, int64_t n, int P, int64_t *__restrict__ ps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t prev = 0;
    for (int p = 0; p < P; p++) {
        int64_t s = (int64_t)p * n / P;
        if (s < prev) s = prev;
        while (s > 0 && s < n && dyn_keys_equal(key, off, s, s - 1)) s++;
        ps[p] = s; prev = s;
    }
    ps[P] = n;
}

// a row heads a family when it opens a partition or differs from its predecessor inside the first lmin bases
__global__ __launch_bounds__(256) void k_dyn_heads(const uint8_t *__restrict__ key, const int64_t *__restrict__ off, int64_t n,
                                                   const int64_t *__restrict__ ps, int P, uint32_t lmin, uint32_t *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    for (int p = 0; p <= P && !h; p++) h = ps[p] == i;
    if (!h) {
        const uint8_t *a = key + off[i], *b = key + off[i - 1];
        for (uint32_t j = 0; j < lmin && !h; j++) h = a[j] != b[j];
    }
    head[i] = h ? 1u : 0u;
}

// an emission: the source record(s) and what is done with them (the orientation is decided by the emission's rank)
struct DynDesc { int32_t kind; int32_t bubble; int64_t a, b; };    // kind 0: flip record a; 1: merge forward a + reflected b

struct DynRow { const uint8_t *key; int klen; int elen; int marker, left, right; int64_t idx; };

__device__ __forceinline__ DynRow dyn_row(const uint8_t *key, const int64_t *koff, const int64_t *eoff, const int32_t *marker,
                                          const int32_t *left, const int32_t *right, int64_t q) {
    DynRow r;
    r.key = key + koff[q]; r.klen = (int)(koff[q + 1] - koff[q]); r.elen = (int)(eoff[q + 1] - eoff[q]);
    r.marker = marker[q]; r.left = left[q]; r.right = right[q]; r.idx = q;
    return r;
}
__device__ __forceinline__ bool dyn_related(const DynRow &a, const DynRow &b) {
    const int n = a.klen < b.klen ? a.klen : b.klen;
    for (int j = 0; j < n; j++) if (a.key[j] != b.key[j]) return false;
    return true;
}

// DSExtendReflexivKmer.call (FirstFour:1603-1763) / DSExtendReflexivKmerToArrayLoop.call (Iteration:487-...) over one family.
// WRITE = false: counts the family's emissions into cnt[head]; WRITE = true: writes descriptors at base[head] + rank.
template <bool WRITE>
__global__ __launch_bounds__(128) void k_dyn_walk(const uint8_t *__restrict__ key, const int64_t *__restrict__ koff,
                                                  const int64_t *__restrict__ eoff, const int32_t *__restrict__ marker,
                                                  const int32_t *__restrict__ left, const int32_t *__restrict__ right, int64_t n,
                                                  const uint32_t *__restrict__ head, int stage, int start_iteration, uint32_t *__restrict__ cnt,
                                                  const uint64_t *__restrict__ base, DynDesc *__restrict__ desc) {
    const int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q0 >= n) return;
    if (!head[q0]) { if (!WRITE) cnt[q0] = 0; return; }
    uint32_t rank = 0;
    const uint64_t b0 = WRITE ? base[q0] : 0;
    auto emit = [&](int kind, int64_t a, int64_t b, int bubble) {
        if (WRITE) desc[b0 + rank] = DynDesc{kind, bubble, a, b};
        rank++;
    };
    bool have = false;
    DynRow h{};
    for (int64_t q = q0; q < n && (q == q0 || !head[q]); q++) {
        const DynRow s = dyn_row(key, koff, eoff, marker, left, right, q);
        if (!have) { h = s; have = true; continue; }
        if (!dyn_related(s, h)) { emit(0, h.idx, -1, 0); h = s; continue; }
        if (s.marker == h.marker) { emit(0, s.idx, -1, 0); continue; }
        if (s.marker == 1) {
            const int extra = h.klen < s.klen ? s.klen - h.klen : 0;
            if (s.klen < h.klen) {                               // the forward row is the shorter one: no merge
                if (stage == 0 || start_iteration < 61) emit(0, s.idx, -1, 0);
                continue;
            }
            int d; bool ok = true;
            if (s.left < 0 && h.right < 0) d = -1;
            else if (s.left >= 0 && h.right >= 0) d = -1;
            else if (s.left >= 0 && s.left - h.elen >= 0) d = s.left - h.elen;
            else if (h.right >= 0 && h.right - s.elen - extra >= 0) d = h.right - s.elen;
            else { ok = false; d = 0; }
            if (!ok) { emit(0, s.idx, -1, 0); continue; }
            emit(1, s.idx, h.idx, d); have = false;
        } else {
            const int extra = s.klen < h.klen ? h.klen - s.klen : 0;
            if (h.klen < s.klen) {                               // the forward HOLDER is the shorter one
                if (stage == 1 && start_iteration >= 61) have = false;
                emit(0, s.idx, -1, 0);
                continue;
            }
            int d; bool ok = true;
            if (s.right < 0 && h.left < 0) d = -1;
            else if (s.right >= 0 && h.left >= 0) d = -1;
            else if (s.right >= 0 && s.right - h.elen - extra >= 0) d = s.right - h.elen;
            else if (h.left >= 0 && h.left - s.elen >= 0) d = h.left - s.elen;
            else { ok = false; d = 0; }
            if (!ok) { emit(0, s.idx, -1, 0); continue; }
            emit(1, h.idx, s.idx, d); have = false;
        }
    }
    if (have) emit(0, h.idx, -1, 0);
    if (!WRITE) cnt[q0] = rank;
}

__device__ __forceinline__ int32_t dyn_clamp(int32_t v) { return v >= 30000 ? 30000 : v <= -30000 ? -30000 : v; }

// emission e of partition p (pbase[p] = rank of the partition's first emission) goes out in orientation
// m = start_marker toggled (e - pbase[p]) times; sizes first, then the bases
__device__ __forceinline__ int dyn_orientation(int64_t e, const uint64_t *pbase, int P, int start_marker) {
    int p = 0;
    for (int t = 1; t < P; t++) if ((int64_t)pbase[t] <= e) p = t;          // the last partition that starts at or before e
    const int64_t r = e - (int64_t)pbase[p];
    return (r & 1) ? 3 - start_marker : start_marker;
}

__global__ __launch_bounds__(256) void k_dyn_sizes(const DynDesc *__restrict__ desc, int64_t ne, const int64_t *__restrict__ koff,
                                                   const int64_t *__restrict__ eoff, uint64_t *__restrict__ ks, uint64_t *__restrict__ es) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const DynDesc d = desc[e];
    if (d.kind == 0) {
        ks[e] = (uint64_t)(koff[d.a + 1] - koff[d.a]);
        es[e] = (uint64_t)(eoff[d.a + 1] - eoff[d.a]);
    } else {
        const int64_t kf = koff[d.a + 1] - koff[d.a], kr = koff[d.b + 1] - koff[d.b];
        ks[e] = (uint64_t)(kf >= kr ? kf : kr);                  // the longer key's length is kept
        es[e] = (uint64_t)((eoff[d.a + 1] - eoff[d.a]) + (eoff[d.b + 1] - eoff[d.b]));
    }
}

// singleKmerRandomizer (FirstFour:1857-1930) / reflexivExtend (:1957-2120) at base level: output record e
__global__ __launch_bounds__(256) void k_dyn_emit(const DynDesc *__restrict__ desc, int64_t ne, const uint8_t *__restrict__ key,
                                                  const int64_t *__restrict__ koff, const uint8_t *__restrict__ ext,
                                                  const int64_t *__restrict__ eoff, const int32_t *__restrict__ marker,
                                                  const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                                  const uint64_t *__restrict__ pbase, int P, int start_marker, const uint64_t *__restrict__ oko,
                                                  const uint64_t *__restrict__ oeo, uint8_t *__restrict__ okey, int64_t *__restrict__ okoff,
                                                  uint8_t *__restrict__ oext, int64_t *__restrict__ oeoff, int32_t *__restrict__ omarker,
                                                  int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > ne) return;
    if (e == ne) { okoff[ne] = (int64_t)oko[ne]; oeoff[ne] = (int64_t)oeo[ne]; return; }
    const DynDesc d = desc[e];
    const int m = dyn_orientation(e, pbase, P, start_marker);
    uint8_t *ok = okey + oko[e], *oe = oext + oeo[e];
    okoff[e] = (int64_t)oko[e]; oeoff[e] = (int64_t)oeo[e];
    if (d.kind == 0) {
        const uint8_t *K = key + koff[d.a], *E = ext + eoff[d.a];
        const int kl = (int)(koff[d.a + 1] - koff[d.a]), el = (int)(eoff[d.a + 1] - eoff[d.a]);
        const int mk = marker[d.a];
        oleft[e] = left[d.a]; oright[e] = right[d.a];
        // (any lengths, as Iteration's array form: combined = key + ext, key' = combined[|ext|:], ext' = combined[:|ext|]; the
        // reflected record the other way round.  FirstFour's single-long form is the same while |ext| <= |key|.)
        if (mk == 1 && m == 2) {
            auto c = [&](int t) -> uint8_t { return t < kl ? K[t] : E[t - kl]; };
            for (int j = 0; j < kl; j++) ok[j] = c(el + j);
            for (int j = 0; j < el; j++) oe[j] = c(j);
            omarker[e] = 2;
        } else if (mk == 2 && m == 1) {
            auto c = [&](int t) -> uint8_t { return t < el ? E[t] : K[t - el]; };
            for (int j = 0; j < kl; j++) ok[j] = c(j);
            for (int j = 0; j < el; j++) oe[j] = c(kl + j);
            omarker[e] = 1;
        } else {
            for (int j = 0; j < kl; j++) ok[j] = K[j];
            for (int j = 0; j < el; j++) oe[j] = E[j];
            omarker[e] = mk;
        }
        return;
    }
    // merge: forward a + reflected b; the longer key L is kept; whole = P + L + S
    const int64_t f = d.a, r = d.b;
    const int kf = (int)(koff[f + 1] - koff[f]), kr = (int)(koff[r + 1] - koff[r]);
    const int S = (int)(eoff[f + 1] - eoff[f]), Pn = (int)(eoff[r + 1] - eoff[r]);
    const uint8_t *L = kf >= kr ? key + koff[f] : key + koff[r];
    const int Ln = kf >= kr ? kf : kr;
    const uint8_t *Sx = ext + eoff[f], *Px = ext + eoff[r];
    const int extra = kf > kr ? kf - kr : 0;
    int lf, rt;
    if (d.bubble < 0) {
        lf = left[r] >= 0 ? left[r] : left[f] - Pn;
        rt = right[f] >= 0 ? right[f] : right[r] - S - extra;
    } else if (left[f] > 0) {
        lf = d.bubble;
        rt = right[f] >= 0 ? right[f] : right[r] - S - extra;
    } else {
        lf = left[r] >= 0 ? left[r] : left[f] - Pn;
        rt = d.bubble - extra;
    }
    oleft[e] = dyn_clamp(lf); oright[e] = dyn_clamp(rt);
    omarker[e] = m;
    // position t of whole = P + L + S
    auto whole = [&](int t) -> uint8_t { return t < Pn ? Px[t] : t < Pn + Ln ? L[t - Pn] : Sx[t - Pn - Ln]; };
    if (m == 2) {                                                  // key' = (L + S)[|S|:], ext' = P + (L + S)[:|S|]
        for (int j = 0; j < Ln; j++) ok[j] = whole(Pn + S + j);
        for (int j = 0; j < Pn + S; j++) oe[j] = whole(j);
    } else {                                                       // key' = (P + L)[:|L|], ext' = (P + L)[|L|:] + S
        for (int j = 0; j < Ln; j++) ok[j] = whole(j);
        for (int j = 0; j < Pn + S; j++) oe[j] = whole(Ln + j);
    }
}

// DSkmerRandomReflection.call (FirstFour:2518-2524): row q of partition p in orientation 2, 1, 2, ... by its rank
__global__ __launch_bounds__(256) void k_dyn_identity_desc(int64_t n, DynDesc *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) desc[i] = DynDesc{0, 0, i, -1};
}
__global__ void k_dyn_copy_u64(const int64_t *__restrict__ src, int n, uint64_t *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint64_t)src[i];
}
// partition bases of the emissions: pbase[p] = base of the family that opens partition p (or the total)
__global__ void k_dyn_pbase(const int64_t *__restrict__ ps, int P, const uint64_t *__restrict__ base, int64_t n, uint64_t total,
                            uint64_t *__restrict__ pbase, int64_t *__restrict__ out_ps) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > P) return;
    const uint64_t v = (p == P || ps[p] >= n) ? total : base[ps[p]];
    pbase[p] = v;
    if (out_ps) out_ps[p] = (int64_t)v;
}

static int dyn_upload(rfx_ctx *ctx, const rfx_dyn_records *h, DynDev &d) {
    const int64_t n = h->n;
    const int64_t nk = n ? h->key_off[n] : 0, ne = n ? h->ext_off[n] : 0;
    RFX_TRY(dyn_alloc(ctx, d, n, nk, ne));
    if (nk) RFX_HIP(hipMemcpyAsync(d.key.p, h->key, (size_t)nk, hipMemcpyHostToDevice, ctx->stream));
    if (ne) RFX_HIP(hipMemcpyAsync(d.ext.p, h->ext, (size_t)ne, hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        RFX_HIP(hipMemcpyAsync(d.key_off.p, h->key_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.ext_off.p, h->ext_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.marker.p, h->marker, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.left.p, h->left, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.right.p, h->right, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    } else {
        RFX_HIP(hipMemsetAsync(d.key_off.p, 0, 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(d.ext_off.p, 0, 8, ctx->stream));
    }
    return RFX_OK;
}
static int dyn_download(rfx_ctx *ctx, const DynDev &d, rfx_dyn_records *h) {
    h->n = d.n; h->need_key = d.nk; h->need_ext = d.ne;
    if (d.n > h->cap_n || d.nk > h->cap_key || d.ne > h->cap_ext) return RFX_E_CAP;
    if (d.nk) RFX_HIP(hipMemcpyAsync(h->key, d.key.p, (size_t)d.nk, hipMemcpyDeviceToHost, ctx->stream));
    if (d.ne) RFX_HIP(hipMemcpyAsync(h->ext, d.ext.p, (size_t)d.ne, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(h->key_off, d.key_off.p, (size_t)(d.n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(h->ext_off, d.ext_off.p, (size_t)(d.n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n) {
        RFX_HIP(hipMemcpyAsync(h->marker, d.marker.p, (size_t)d.n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h->left, d.left.p, (size_t)d.n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h->right, d.right.p, (size_t)d.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

#define GRID(n) dim3((unsigned)ceil_div(std::max<int64_t>((n), 1), 256)), dim3(256), 0, ctx->stream

// sort("k-1") + the cut into P logical partitions: in -> out (sorted), d_ps[P + 1]; *lmin = the shortest key
static int dyn_sort(rfx_ctx *ctx, const DynDev &in, int P, DynDev &out, DevBuf &d_ps, uint32_t *lmin) {
    const int64_t n = in.n;
    RFX_HIP(d_ps.alloc((size_t)(P + 1) * 8, ctx->stream));
    RFX_TRY(dyn_alloc(ctx, out, n, in.nk, in.ne));
    *lmin = 0;
    if (n == 0) {
        RFX_HIP(hipMemsetAsync(d_ps.p, 0, (size_t)(P + 1) * 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(out.key_off.p, 0, 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(out.ext_off.p, 0, 8, ctx->stream));
        return RFX_OK;
    }
    if (n >= ((int64_t)1 << 32)) { ctx->last_error = "dynamic-k sort: more than 2^32 records"; return RFX_E_LIMIT; }
    DevBuf blk, nblk, perm, tk, tv, keys, flags, ks, es, kso, eso;
    RFX_HIP(blk.alloc((size_t)DYN_MAXB * n * 8, ctx->stream)); RFX_HIP(nblk.alloc((size_t)n * 8, ctx->stream));
    RFX_HIP(perm.alloc((size_t)n * 4, ctx->stream)); RFX_HIP(tk.alloc((size_t)n * 8, ctx->stream)); RFX_HIP(tv.alloc((size_t)n * 4, ctx->stream));
    RFX_HIP(keys.alloc((size_t)n * 8, ctx->stream)); RFX_HIP(flags.alloc(16, ctx->stream));
    RFX_HIP(hipMemsetAsync(flags.p, 0, 4, ctx->stream));
    RFX_HIP(hipMemsetAsync((char *)flags.p + 4, 0xFF, 4, ctx->stream));
    hipLaunchKernelGGL(k_dyn_blocks, GRID(n), in.key.as<uint8_t>(), in.key_off.as<int64_t>(), n, blk.as<uint64_t>(), nblk.as<uint64_t>(),
                       perm.as<uint32_t>(), flags.as<int>(), flags.as<uint32_t>() + 1);
    RFX_HIP(hipGetLastError());
    // LSD: the block count (a proper prefix first when every shared block is equal), then block 3 .. block 0
    for (int pass = -1; pass < DYN_MAXB; pass++) {
        const uint64_t *src = pass < 0 ? nblk.as<uint64_t>() : blk.as<uint64_t>() + (int64_t)(DYN_MAXB - 1 - pass) * n;
        hipLaunchKernelGGL(k_dyn_gather_u64, GRID(n), src, (const uint32_t *)perm.as<uint32_t>(), n, keys.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_TRY(sort_pairs(ctx, keys.as<uint64_t>(), perm.as<uint32_t>(), n, pass < 0 ? 8 : 64, tk.as<uint64_t>(), tv.as<uint32_t>()));
    }
    int h_flags[2] = {0, 0};
    RFX_HIP(hipMemcpyAsync(h_flags, flags.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    if (h_flags[0]) { ctx->last_error = "dynamic-k: a key longer than 124 bases"; return RFX_E_LIMIT; }
    *lmin = (uint32_t)h_flags[1];
    // gather the records through the permutation
    RFX_HIP(ks.alloc((size_t)n * 8, ctx->stream)); RFX_HIP(es.alloc((size_t)n * 8, ctx->stream));
    RFX_HIP(kso.alloc((size_t)(n + 1) * 8, ctx->stream)); RFX_HIP(eso.alloc((size_t)(n + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_dyn_perm_sizes, GRID(n), (const int64_t *)in.key_off.as<int64_t>(), (const int64_t *)in.ext_off.as<int64_t>(),
                       (const uint32_t *)perm.as<uint32_t>(), n, ks.as<uint64_t>(), es.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u64(ctx, ks.as<uint64_t>(), kso.as<uint64_t>(), n));
    RFX_TRY(exclusive_scan_u64(ctx, es.as<uint64_t>(), eso.as<uint64_t>(), n));
    hipLaunchKernelGGL(k_dyn_gather, GRID(n + 1), (const uint8_t *)in.key.as<uint8_t>(), (const int64_t *)in.key_off.as<int64_t>(),
                       (const uint8_t *)in.ext.as<uint8_t>(), (const int64_t *)in.ext_off.as<int64_t>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(), (const uint32_t *)perm.as<uint32_t>(), n,
                       (const uint64_t *)kso.as<uint64_t>(), (const uint64_t *)eso.as<uint64_t>(), out.key.as<uint8_t>(), out.key_off.as<int64_t>(),
                       out.ext.as<uint8_t>(), out.ext_off.as<int64_t>(), out.marker.as<int32_t>(), out.left.as<int32_t>(), out.right.as<int32_t>());
    RFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_dyn_part_starts, dim3(1), dim3(1), 0, ctx->stream, (const uint8_t *)out.key.as<uint8_t>(),
                       (const int64_t *)out.key_off.as<int64_t>(), n, P, d_ps.as<int64_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

// descriptors -> the output record set
static int dyn_emit(rfx_ctx *ctx, const DynDev &in, const DevBuf &desc, int64_t ne, const DevBuf &pbase, int P, int start_marker, DynDev &out) {
    DevBuf ks, es, kso, eso;
    RFX_HIP(ks.alloc((size_t)std::max<int64_t>(ne, 1) * 8, ctx->stream)); RFX_HIP(es.alloc((size_t)std::max<int64_t>(ne, 1) * 8, ctx->stream));
    RFX_HIP(kso.alloc((size_t)(ne + 1) * 8, ctx->stream)); RFX_HIP(eso.alloc((size_t)(ne + 1) * 8, ctx->stream));
    if (ne > 0) {
        hipLaunchKernelGGL(k_dyn_sizes, GRID(ne), (const DynDesc *)desc.as<DynDesc>(), ne, (const int64_t *)in.key_off.as<int64_t>(),
                           (const int64_t *)in.ext_off.as<int64_t>(), ks.as<uint64_t>(), es.as<uint64_t>());
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u64(ctx, ks.as<uint64_t>(), kso.as<uint64_t>(), ne));
    RFX_TRY(exclusive_scan_u64(ctx, es.as<uint64_t>(), eso.as<uint64_t>(), ne));
    uint64_t tot[2] = {0, 0};
    RFX_HIP(hipMemcpyAsync(&tot[0], kso.as<uint64_t>() + ne, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(&tot[1], eso.as<uint64_t>() + ne, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    RFX_TRY(dyn_alloc(ctx, out, ne, (int64_t)tot[0], (int64_t)tot[1]));
    hipLaunchKernelGGL(k_dyn_emit, GRID(ne + 1), (const DynDesc *)desc.as<DynDesc>(), ne, (const uint8_t *)in.key.as<uint8_t>(),
                       (const int64_t *)in.key_off.as<int64_t>(), (const uint8_t *)in.ext.as<uint8_t>(), (const int64_t *)in.ext_off.as<int64_t>(),
                       (const int32_t *)in.marker.as<int32_t>(), (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(),
                       (const uint64_t *)pbase.as<uint64_t>(), P, start_marker, (const uint64_t *)kso.as<uint64_t>(), (const uint64_t *)eso.as<uint64_t>(),
                       out.key.as<uint8_t>(), out.key_off.as<int64_t>(), out.ext.as<uint8_t>(), out.ext_off.as<int64_t>(), out.marker.as<int32_t>(),
                       out.left.as<int32_t>(), out.right.as<int32_t>());
    RFX_HIP(hipGetLastError());
    return RFX_OK;
}

// one pass over sorted records: in (sorted), d_ps -> out, d_out_ps (optional)
static int dyn_pass(rfx_ctx *ctx, const DynDev &in, const int64_t *d_ps, int P, uint32_t lmin, int stage, int start_iteration, int start_marker,
                    DynDev &out, int64_t *d_out_ps) {
    const int64_t n = in.n;
    DevBuf head, cnt, base, desc, pbase;
    RFX_HIP(pbase.alloc((size_t)(P + 1) * 8, ctx->stream));
    if (n == 0) {
        RFX_HIP(hipMemsetAsync(pbase.p, 0, (size_t)(P + 1) * 8, ctx->stream));
        if (d_out_ps) RFX_HIP(hipMemsetAsync(d_out_ps, 0, (size_t)(P + 1) * 8, ctx->stream));
        RFX_HIP(desc.alloc(sizeof(DynDesc), ctx->stream));
        return dyn_emit(ctx, in, desc, 0, pbase, P, start_marker, out);
    }
    RFX_HIP(head.alloc((size_t)n * 4, ctx->stream)); RFX_HIP(cnt.alloc((size_t)n * 4, ctx->stream)); RFX_HIP(base.alloc((size_t)(n + 1) * 8, ctx->stream));
    RFX_HIP(desc.alloc((size_t)n * sizeof(DynDesc), ctx->stream));
    hipLaunchKernelGGL(k_dyn_heads, GRID(n), (const uint8_t *)in.key.as<uint8_t>(), (const int64_t *)in.key_off.as<int64_t>(), n, d_ps, P, lmin,
                       head.as<uint32_t>());
    RFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_dyn_walk<false>, dim3((unsigned)ceil_div(n, 128)), dim3(128), 0, ctx->stream, (const uint8_t *)in.key.as<uint8_t>(),
                       (const int64_t *)in.key_off.as<int64_t>(), (const int64_t *)in.ext_off.as<int64_t>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(), n, (const uint32_t *)head.as<uint32_t>(), stage,
                       start_iteration, cnt.as<uint32_t>(), (const uint64_t *)nullptr, (DynDesc *)nullptr);
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, cnt.as<uint32_t>(), base.as<uint64_t>(), n));
    uint64_t ne = 0;
    RFX_HIP(hipMemcpyAsync(&ne, base.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    hipLaunchKernelGGL(k_dyn_walk<true>, dim3((unsigned)ceil_div(n, 128)), dim3(128), 0, ctx->stream, (const uint8_t *)in.key.as<uint8_t>(),
                       (const int64_t *)in.key_off.as<int64_t>(), (const int64_t *)in.ext_off.as<int64_t>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(), n, (const uint32_t *)head.as<uint32_t>(), stage,
                       start_iteration, (uint32_t *)nullptr, (const uint64_t *)base.as<uint64_t>(), desc.as<DynDesc>());
    RFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_dyn_pbase, dim3(1), dim3(64), 0, ctx->stream, d_ps, P, (const uint64_t *)base.as<uint64_t>(), n, ne, pbase.as<uint64_t>(), d_out_ps);
    RFX_HIP(hipGetLastError());
    return dyn_emit(ctx, in, desc, (int64_t)ne, pbase, P, start_marker, out);
}

static int dyn_reflect(rfx_ctx *ctx, const DynDev &in, const int64_t *d_ps, int P, DynDev &out) {
    DevBuf desc, pbase;
    RFX_HIP(desc.alloc((size_t)std::max<int64_t>(in.n, 1) * sizeof(DynDesc), ctx->stream));
    RFX_HIP(pbase.alloc((size_t)(P + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_dyn_identity_desc, GRID(in.n), in.n, desc.as<DynDesc>());
    hipLaunchKernelGGL(k_dyn_copy_u64, dim3(1), dim3(64), 0, ctx->stream, d_ps, P + 1, pbase.as<uint64_t>());
    RFX_HIP(hipGetLastError());
    return dyn_emit(ctx, in, desc, in.n, pbase, P, 2, out);
}

}  // namespace

extern "C" {

int rfx_dyn_sort(rfx_ctx *ctx, const rfx_dyn_records *in, int P, rfx_dyn_records *out, int64_t *part_start) try {
    if (!ctx || !in || !out || !part_start || P < 1 || P > 63) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    DynDev a, b;
    DevBuf ps;
    uint32_t lmin = 0;
    RFX_TRY(dyn_upload(ctx, in, a));
    RFX_TRY(dyn_sort(ctx, a, P, b, ps, &lmin));
    RFX_HIP(hipMemcpyAsync(part_start, ps.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    return dyn_download(ctx, b, out);
} RFX_API_CATCH(ctx)

// ---- DynamicKmerBinarizerFromReducedToSubKmer (FirstFour :2931-3016; Iteration's twin) on the device: the text rows of the hand-over
// files -> records.  A row is its fields joined by ',': form 0 = (k-mer, "m|l|r") -- key = the k-mer without its last base, extension =
// that base, orientation 1 --, form 1 = (sub-k-mer, "m|l|r", extension).  A leading '(' of the first field and a trailing ')' of the
// attribute are dropped (the tuple text Spark writes); left / right are read back clamped to +-30000 (buildingAlongFromThreeInt
// :2340-2366); A0 C1 G2, anything else 3.  One thread per row, twice: sizes (+ the attribute), then the bases.
struct DynRowCut { int64_t f0, f0e, f1, f1e, f2, f2e; };     // the three fields' [begin, end) in the text
__device__ __forceinline__ DynRowCut dyn_row_cut(const char *__restrict__ t, int64_t b, int64_t e) {
    while (e > b && (t[e - 1] == '\n' || t[e - 1] == '\r')) e--;
    int64_t c1 = e, c2 = e;
    for (int64_t i = b; i < e; i++) if (t[i] == ',') { if (c1 == e) c1 = i; else { c2 = i; break; } }
    DynRowCut c{b, c1, c1 < e ? c1 + 1 : e, c2, c2 < e ? c2 + 1 : e, e};
    if (c.f0 < c.f0e && t[c.f0] == '(') c.f0++;
    if (c.f1e > c.f1 && t[c.f1e - 1] == ')') c.f1e--;
    if (c.f2e > c.f2 && t[c.f2e - 1] == ')') c.f2e--;
    return c;
}
__device__ __forceinline__ int dyn_parse_int(const char *__restrict__ t, int64_t &i, int64_t e) {
    bool neg = false;
    if (i < e && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; i++; }
    long long v = 0;
    while (i < e && t[i] >= '0' && t[i] <= '9') { if (v < 100000000LL) v = v * 10 + (t[i] - '0'); i++; }
    if (i < e && t[i] == '|') i++;
    return (int)(neg ? -v : v);
}
__global__ void k_dyn_bin_sizes(const char *__restrict__ text, const int64_t *__restrict__ row_off, int64_t n, int form,
                                uint32_t *__restrict__ klen, uint32_t *__restrict__ elen, int32_t *__restrict__ marker,
                                int32_t *__restrict__ left, int32_t *__restrict__ right) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const DynRowCut c = dyn_row_cut(text, row_off[r], row_off[r + 1]);
    const int64_t l0 = c.f0e - c.f0;
    if (form == 0) { klen[r] = (uint32_t)(l0 > 0 ? l0 - 1 : 0); elen[r] = l0 > 0 ? 1u : 0u; }
    else { klen[r] = (uint32_t)l0; elen[r] = (uint32_t)(c.f2e - c.f2); }
    int64_t i = c.f1;
    const int m = dyn_parse_int(text, i, c.f1e), l = dyn_parse_int(text, i, c.f1e), rr = dyn_parse_int(text, i, c.f1e);
    marker[r] = form == 0 ? 1 : m;
    left[r] = l < -30000 ? -30000 : l > 30000 ? 30000 : l;
    right[r] = rr < -30000 ? -30000 : rr > 30000 ? 30000 : rr;
}
__device__ __forceinline__ uint8_t dyn_code(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }
__global__ void k_dyn_bin_fill(const char *__restrict__ text, const int64_t *__restrict__ row_off, int64_t n, int form,
                               const uint64_t *__restrict__ key_off, const uint64_t *__restrict__ ext_off, uint8_t *__restrict__ key,
                               uint8_t *__restrict__ ext) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const DynRowCut c = dyn_row_cut(text, row_off[r], row_off[r + 1]);
    const int64_t nk = (int64_t)(key_off[r + 1] - key_off[r]), ne = (int64_t)(ext_off[r + 1] - ext_off[r]);
    for (int64_t j = 0; j < nk; j++) key[key_off[r] + j] = dyn_code(text[c.f0 + j]);
    const int64_t es = form == 0 ? c.f0 + nk : c.f2;
    for (int64_t j = 0; j < ne; j++) ext[ext_off[r] + j] = dyn_code(text[es + j]);
}

int rfx_dyn_binarize(rfx_ctx *ctx, const char *text, const int64_t *row_off, int64_t n_rows, int form, rfx_dyn_records *out) try {
    if (!ctx || !out || n_rows < 0 || (n_rows > 0 && (!text || !row_off)) || (form != 0 && form != 1)) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    for (int64_t i = 0; i < n_rows; i++) if (row_off[i + 1] < row_off[i]) return RFX_E_ARG;
    const int64_t nb = n_rows ? row_off[n_rows] - row_off[0] : 0;
    DevBuf d_text, d_off, klen, elen;
    DynDev d;
    RFX_HIP(d_text.alloc((size_t)std::max<int64_t>(nb, 1), ctx->stream));
    RFX_HIP(d_off.alloc((size_t)(n_rows + 1) * 8, ctx->stream));
    RFX_HIP(klen.alloc((size_t)std::max<int64_t>(n_rows, 1) * 4, ctx->stream));
    RFX_HIP(elen.alloc((size_t)std::max<int64_t>(n_rows, 1) * 4, ctx->stream));
    // (records first with room for every byte of the text: the sizes are known only after the first kernel)
    RFX_TRY(dyn_alloc(ctx, d, n_rows, nb, nb));
    if (n_rows == 0) {
        RFX_HIP(hipMemsetAsync(d.key_off.p, 0, 8, ctx->stream));
        RFX_HIP(hipMemsetAsync(d.ext_off.p, 0, 8, ctx->stream));
        d.nk = d.ne = 0;
        return dyn_download(ctx, d, out);
    }
    std::vector<int64_t> rel((size_t)n_rows + 1);
    for (int64_t i = 0; i <= n_rows; i++) rel[(size_t)i] = row_off[i] - row_off[0];
    if (nb) RFX_HIP(hipMemcpyAsync(d_text.p, text + row_off[0], (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    RFX_HIP(hipMemcpyAsync(d_off.p, rel.data(), (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_dyn_bin_sizes, GRID(n_rows), (const char *)d_text.p, (const int64_t *)d_off.as<int64_t>(), n_rows, form,
                       klen.as<uint32_t>(), elen.as<uint32_t>(), d.marker.as<int32_t>(), d.left.as<int32_t>(), d.right.as<int32_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, klen.as<uint32_t>(), d.key_off.as<uint64_t>(), n_rows));
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, elen.as<uint32_t>(), d.ext_off.as<uint64_t>(), n_rows));
    hipLaunchKernelGGL(k_dyn_bin_fill, GRID(n_rows), (const char *)d_text.p, (const int64_t *)d_off.as<int64_t>(), n_rows, form,
                       (const uint64_t *)d.key_off.as<uint64_t>(), (const uint64_t *)d.ext_off.as<uint64_t>(), d.key.as<uint8_t>(), d.ext.as<uint8_t>());
    RFX_HIP(hipGetLastError());
    uint64_t tot[2] = {0, 0};
    RFX_HIP(hipMemcpyAsync(&tot[0], d.key_off.as<uint64_t>() + n_rows, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_HIP(hipMemcpyAsync(&tot[1], d.ext_off.as<uint64_t>() + n_rows, 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    d.nk = (int64_t)tot[0]; d.ne = (int64_t)tot[1];
    return dyn_download(ctx, d, out);
} RFX_API_CATCH(ctx)

int rfx_dyn_random_reflection(rfx_ctx *ctx, const rfx_dyn_records *in, const int64_t *part_start, int P, rfx_dyn_records *out) try {
    if (!ctx || !in || !out || !part_start || P < 1 || P > 63) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    DynDev a, b;
    DevBuf ps;
    RFX_TRY(dyn_upload(ctx, in, a));
    RFX_HIP(ps.alloc((size_t)(P + 1) * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(ps.p, part_start, (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RFX_TRY(dyn_reflect(ctx, a, ps.as<int64_t>(), P, b));
    return dyn_download(ctx, b, out);
} RFX_API_CATCH(ctx)

int rfx_dyn_extend_pass(rfx_ctx *ctx, const rfx_dyn_records *in, const int64_t *part_start, int P, int stage, int start_iteration,
                        int start_marker, rfx_dyn_records *out, int64_t *out_part_start) try {
    if (!ctx || !in || !out || !part_start || P < 1 || P > 63 || (stage != 0 && stage != 1) || (start_marker != 1 && start_marker != 2))
        return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    DynDev a, b;
    DevBuf ps, ops;
    RFX_TRY(dyn_upload(ctx, in, a));
    RFX_HIP(ps.alloc((size_t)(P + 1) * 8, ctx->stream)); RFX_HIP(ops.alloc((size_t)(P + 1) * 8, ctx->stream));
    RFX_HIP(hipMemcpyAsync(ps.p, part_start, (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    int64_t lmin = INT64_MAX;
    for (int64_t i = 0; i < in->n; i++) lmin = std::min(lmin, in->key_off[i + 1] - in->key_off[i]);
    if (in->n == 0) lmin = 0;
    RFX_TRY(dyn_pass(ctx, a, ps.as<int64_t>(), P, (uint32_t)lmin, stage, start_iteration, start_marker, b, ops.as<int64_t>()));
    if (out_part_start) RFX_HIP(hipMemcpyAsync(out_part_start, ops.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    return dyn_download(ctx, b, out);
} RFX_API_CATCH(ctx)

// The drivers, records resident in HBM between the operators.  FirstFour.assemblyFromKmer (:137-224): binarized records
// (key = the k-mer without its last base, extension = that base, orientation 1) -> DSkmerRandomReflection on P equal
// shares -> 4 x (sort, DSExtendReflexivKmer).  Iteration.assemblyFromKmer (:134-205): (end - start + 1) x (sort,
// DSExtendReflexivKmerToArrayLoop).  passes_first_four / start / end select what runs; out = the final records.
int rfx_dyn_run(rfx_ctx *ctx, const rfx_dyn_records *in, int P, int random_reflection, int passes_first_four, int start_iteration,
                int end_iteration, rfx_dyn_records *out, int64_t *trace, int64_t trace_cap, int64_t *n_trace) try {
    if (!ctx || !in || !out || P < 1 || P > 63 || passes_first_four < 0) return RFX_E_ARG;
    RFX_HIP(hipSetDevice(ctx->device));
    DynDev a, b;
    DevBuf ps;
    int64_t nt = 0;
    RFX_TRY(dyn_upload(ctx, in, a));
    if (random_reflection) {
        std::vector<int64_t> st((size_t)P + 1);
        for (int p = 0; p <= P; p++) st[(size_t)p] = p == P ? a.n : (int64_t)(((__int128)p * (__int128)a.n) / P);
        RFX_HIP(ps.alloc((size_t)(P + 1) * 8, ctx->stream));
        RFX_HIP(hipMemcpyAsync(ps.p, st.data(), (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        RFX_TRY(sync_checked(ctx));
        RFX_TRY(dyn_reflect(ctx, a, ps.as<int64_t>(), P, b));
        dyn_swap(a, b);
    }
    auto one = [&](int stage, int start_it) -> int {
        uint32_t lmin = 0;
        DynDev s;
        RFX_TRY(dyn_sort(ctx, a, P, s, ps, &lmin));
        DynDev o;
        RFX_TRY(dyn_pass(ctx, s, ps.as<int64_t>(), P, lmin, stage, start_it, 2, o, nullptr));
        RFX_TRY(sync_checked(ctx));
        dyn_swap(a, o);
        if (trace && nt < trace_cap) trace[nt] = a.n;
        nt++;
        return RFX_OK;
    };
    for (int i = 0; i < passes_first_four; i++) RFX_TRY(one(0, 0));
    for (int it = start_iteration; end_iteration >= start_iteration && it <= end_iteration; it++) RFX_TRY(one(1, start_iteration));
    if (n_trace) *n_trace = nt;
    return dyn_download(ctx, a, out);
} RFX_API_CATCH(ctx)

// ---- the Row form of the third layout (what a JNI shim converts between) -----------------------------------------------

// left-aligned 31-base blocks with a 01 terminator -> base codes; returns the length (currentKmerSizeFromBinaryBlockArray,
// FirstFour:2301-2311), or -1 when cap is short
int rfx_dyn_blocks_to_bases(const int64_t *blocks, int n_blocks, uint8_t *out, int cap) try {
    if (!blocks || n_blocks < 1) return -1;
    const uint64_t last = (uint64_t)blocks[n_blocks - 1];
    const int tz = last ? __builtin_ctzll(last) : 64;
    const int len = (n_blocks - 1) * 31 + (32 - tz / 2 - 1);
    if (len > cap) return -1;
    for (int i = 0; i < len; i++) out[i] = (uint8_t)(((uint64_t)blocks[i / 31] >> (2 * (31 - i % 31))) & 3);
    return len < 0 ? 0 : len;
} RFX_API_CATCH(nullptr)
// base codes -> blocks; returns the number of blocks ((n - 1) / 31 + 1), or -1 when cap is short
int rfx_dyn_bases_to_blocks(const uint8_t *bases, int n, int64_t *out, int cap) try {
    const int nb = n <= 0 ? 1 : (n - 1) / 31 + 1;
    if (nb > cap || !out) return -1;
    for (int j = 0; j < nb; j++) {
        uint64_t x = 0;
        const int b0 = 31 * j;
        int m = n - b0;
        if (m > 31) m = 31;
        if (m < 0) m = 0;
        for (int i = 0; i < m; i++) x |= (uint64_t)bases[b0 + i] << (2 * (31 - i));
        if (b0 + 31 >= n) x |= 1ULL << (2 * (31 - m));
        out[j] = (int64_t)x;
    }
    return nb;
} RFX_API_CATCH(nullptr)
// buildingAlongFromThreeInt (FirstFour:2340-2366) and getReflexivMarker / getLeftMarker / getRightMarker (:2313-2338)
int64_t rfx_dyn_attribute(int marker, int left, int right) try {
    if (left >= 30000) left = 30000; else if (left <= -30000) left = 60000; else if (left < 0) left = 30000 - left;
    if (right >= 30000) right = 30000; else if (right <= -30000) right = 60000; else if (right < 0) right = 30000 - right;
    return (int64_t)(((uint64_t)(uint32_t)marker << 62) | ((uint64_t)(uint32_t)left << 32) | (uint64_t)(uint32_t)right);
} RFX_API_CATCH(nullptr)
void rfx_dyn_attribute_unpack(int64_t a, int *marker, int *left, int *right) try {
    if (marker) *marker = (int)((uint64_t)a >> 62);
    int l = (int)((uint64_t)a >> 32) & ~(3 << 30);
    if (l > 30000) l = 30000 - l;
    int r = (int)a;
    if (r > 30000) r = 30000 - r;
    if (left) *left = l;
    if (right) *right = r;
} RFX_API_CATCH_VOID(nullptr)

}  // extern "C"
