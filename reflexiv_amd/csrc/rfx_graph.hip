// rfx_graph.hip -- the record-building operators between the count stage and the extend
// loop (K4..K9 of SURVEY.md 2.3) plus the record sort that stands in for sortByKey().
//
//   KmerReverseComplement + ForwardSubKmerExtraction   P/ReflexivMain.java:2910-2930, 2709-2730
//   sortByKey()                                        P/ReflexivMain.java:179,191,211,...
//   FilterForkSubKmer[WithErrorCorrection]             P/ReflexivMain.java:2412-2540 (DS :3375-3483)
//   ReflectedSubKmerExtractionFromForward              P/ReflexivMain.java:2742-2768
//   FilterForkReflectedSubKmer[WithErrorCorrection]    P/ReflexivMain.java:2550-2696 (DS :3493-3616)
//   kmerRandomReflection                               P/ReflexivMain.java:2783-2885
// and their k > 31 twins of P/ReflexivDSMain64.java (DSKmerReverseComplement :10706-10755,
// DSForwardSubKmerExtraction :10363-10403, DSFilterFork* :10072-10360, DSReflectedSubKmerExtractionFromForward
// :10426-10475, DSkmerRandomReflection :10491-10690, KmerBinarizer :10772-10836): the same operators on
// (k-1)-mer keys of KW words of 31 bases (templates on KW; KW = 1 keeps the one-word code of k <= 31).
//
// Records live in HBM as flat struct-of-arrays in the reference's layout (include/
// reflexiv_hip.h).  A sort moves (key, index) pairs only; fixed fields are gathered once
// through the permutation and extension words are copied record by record.  The fork
// filters are segmented folds over equal-key runs: the thread that owns a run head walks
// its run (<= 4 records once both strands are present) and writes the survivor at the
// run's compacted position.
#include "rfx_internal.h"
#include "rfx_device.h"

using namespace rfxd;

namespace {

__global__ void k_iota_u32(uint32_t *v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void k_iota_i64(int64_t *v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// ---- K4 + K5
__global__ void k_rc_expand(const uint64_t *__restrict__ kmers, const int32_t *__restrict__ counts, int64_t n,
                            int k, uint64_t *__restrict__ key, int32_t *__restrict__ marker,
                            int64_t *__restrict__ ext_off, uint64_t *__restrict__ ext,
                            int32_t *__restrict__ left, int32_t *__restrict__ right) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = kmers[i];
    uint64_t two[2] = {x, revcomp(x, k)};                 // (kmer, rc) adjacent  :2925-2926
    int32_t c = counts[i];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        int64_t o = 2 * i + t;
        key[o] = two[t] >> 2;                              // :2720
        ext[o] = two[t] & 3;                               // :2719 (no sentinel yet)
        marker[o] = 1; left[o] = c; right[o] = c;          // :2724
        ext_off[o] = o;
    }
    if (i == n - 1) ext_off[2 * n] = 2 * n;
}

// k > 31: k-mers come in the assembler layout (AW = (k-1)/31+1 words of 31 bases, the last word the rest,
// KmerBinarizer :10812-10819); base-wise, as DSKmerReverseComplement does (:10727-10743)
template <int KW>
__global__ void k_rc_expand_w(const uint64_t *__restrict__ kmers, int AW, const int32_t *__restrict__ counts, int64_t n,
                              int k, KeyW<KW> *__restrict__ key, int32_t *__restrict__ marker,
                              int64_t *__restrict__ ext_off, uint64_t *__restrict__ ext,
                              int32_t *__restrict__ left, int32_t *__restrict__ right) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *km = kmers + i * AW;
    const int lastb = k - 31 * (AW - 1);
    auto fwd = [&](int t) -> unsigned {
        const int wi = t / 31, j = t - 31 * wi, nb = wi < AW - 1 ? 31 : lastb;
        return (unsigned)((km[wi] >> (2 * (nb - 1 - j))) & 3);
    };
    auto rc = [&](int t) -> unsigned { return fwd(k - 1 - t) ^ 3u; };
    const int32_t c = counts[i];
    const int sub = k - 1;
    key[2 * i] = build_key<KW>(sub, fwd);                  // :10381-10395
    key[2 * i + 1] = build_key<KW>(sub, rc);
    ext[2 * i] = fwd(k - 1); ext[2 * i + 1] = rc(k - 1);   // :10383 / :10390 (no sentinel yet)
#pragma unroll
    for (int t = 0; t < 2; t++) {
        int64_t o = 2 * i + t;
        marker[o] = 1; left[o] = c; right[o] = c; ext_off[o] = o;          // :10399
    }
    if (i == n - 1) ext_off[2 * n] = 2 * n;
}

// KmerBinarizer + the count filter of the from-counts driver (P/ReflexivDSMain64.java:10772-10836, :473-478) for
// k-mers that never left HBM: counter layout (k/32+1 words of 32 bases, the last word k%32 right-aligned,
// P/ReflexivDataFrameCounter64.java:429-437) -> assembler layout; the count is read as the text would be
// (10 digits or more -> 1000000000, :10801-10806)
__global__ void k_counter_keep(const int64_t *__restrict__ counts, int64_t n, int min_cov, int max_cov,
                               uint32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t c64 = counts[i];
    const int32_t c = c64 >= 1000000000LL ? 1000000000 : (int32_t)c64;
    flag[i] = (c >= min_cov && c <= max_cov) ? 1u : 0u;
}
__global__ void k_counter_to_asm(const uint64_t *__restrict__ k32, const int64_t *__restrict__ counts, int64_t n, int k,
                                 const uint32_t *__restrict__ flag, const uint64_t *__restrict__ pos,
                                 uint64_t *__restrict__ k31, int32_t *__restrict__ ocounts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int W32 = k / 32 + 1, W31 = (k - 1) / 31 + 1, res = k % 32;
    const uint64_t *src = k32 + i * W32;
    uint64_t *dst = k31 + (int64_t)pos[i] * W31;
    auto base = [&](int t) -> uint64_t {
        const int wi = t >> 5, j = t & 31;
        return wi < W32 - 1 ? (src[wi] >> (2 * (31 - j))) & 3 : (src[wi] >> (2 * (res - 1 - j))) & 3;
    };
    int t = 0;
    for (int w = 0; w < W31; w++) {
        const int nb = w < W31 - 1 ? 31 : k - 31 * (W31 - 1);
        uint64_t x = 0;
        for (int j = 0; j < nb; j++) x = (x << 2) | base(t++);
        dst[w] = x;
    }
    const int64_t c64 = counts[i];
    ocounts[pos[i]] = c64 >= 1000000000LL ? 1000000000 : (int32_t)c64;
}

// ---- sort support
template <int KW>
__global__ void k_key_word(const KeyW<KW> *__restrict__ key, const uint32_t *__restrict__ perm, int64_t n, int w,
                           uint64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = key[perm ? perm[i] : (uint32_t)i].w[w];
}
// Two-word keys (k = 33..63) sorted in ONE radix sort: on the key's first 64 bits (word 0's 31 bases and the first base
// of word 1) -- k_key_prefix2 -- after which k_tie_fix2 orders the runs of equal prefixes by the rest of word 1.  Equal
// prefixes are mostly equal KEYS (the sort exists to bring them together: runs of 2..4), so the mending is a stable
// insertion sort of a few permutation entries by the owner of the run's head; a run longer than TIE_MAX (low-complexity
// keys) raises a flag and the caller takes the two-pass LSD form instead.
constexpr int TIE_MAX = 32;
__global__ void k_key_prefix2(const KeyW<2> *__restrict__ key, int64_t n, int res, uint64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (key[i].w[0] << 2) | (key[i].w[1] >> (2 * res - 2));
}
__global__ void k_tie_fix2(const uint64_t *__restrict__ prefix, uint32_t *__restrict__ perm, int64_t n,
                           const KeyW<2> *__restrict__ key, int res, int *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = prefix[i];
    if ((i > 0 && prefix[i - 1] == p) || i + 1 >= n || prefix[i + 1] != p) return;      // not the head of a run of >= 2
    int r = 2;
    while (i + r < n && r <= TIE_MAX && prefix[i + r] == p) r++;
    if (r > TIE_MAX) { *flag = 1; return; }
    const uint64_t rm = res > 1 ? ~((~0ULL) << (2 * res - 2)) : 0ULL;
    // (nearly every run is already in order -- equal keys: checked on the fly, the arrays below stay untouched)
    bool sorted = true;
    uint64_t prev = 0;
    for (int j = 0; j < r; j++) {
        const uint64_t x = key[perm[i + j]].w[1] & rm;
        if (j && x < prev) sorted = false;
        prev = x;
    }
    if (sorted) return;
    uint32_t idx[TIE_MAX];
    uint64_t rest[TIE_MAX];
    for (int j = 0; j < r; j++) { idx[j] = perm[i + j]; rest[j] = key[idx[j]].w[1] & rm; }
    for (int j = 1; j < r; j++) {                     // stable insertion sort by the rest of word 1
        const uint64_t x = rest[j];
        const uint32_t xi = idx[j];
        int q = j - 1;
        while (q >= 0 && rest[q] > x) { rest[q + 1] = rest[q]; idx[q + 1] = idx[q]; q--; }
        rest[q + 1] = x; idx[q + 1] = xi;
    }
    for (int j = 0; j < r; j++) perm[i + j] = idx[j];
}
template <int KW>
__global__ void k_gather_key(const KeyW<KW> *__restrict__ key, const uint32_t *__restrict__ perm, int64_t n,
                             KeyW<KW> *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = key[perm[i]];
}

__global__ void k_gather_fixed(const uint32_t *__restrict__ perm, int64_t n, const int32_t *__restrict__ marker,
                               const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                               const int64_t *__restrict__ ext_off, int32_t *__restrict__ omarker,
                               int32_t *__restrict__ oleft, int32_t *__restrict__ oright,
                               uint32_t *__restrict__ onw) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = perm[i];
    omarker[i] = marker[s]; oleft[i] = left[s]; oright[i] = right[s];
    onw[i] = (uint32_t)(ext_off[s + 1] - ext_off[s]);
}

// every record holds exactly one extension word (the single-word stages): fixed fields and that word in one go, the
// output offsets are the identity -- no scan of word counts, no second gather
__global__ void k_gather_single(const uint32_t *__restrict__ perm, int64_t n, const int32_t *__restrict__ marker,
                                const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                const uint64_t *__restrict__ ext, int32_t *__restrict__ omarker,
                                int32_t *__restrict__ oleft, int32_t *__restrict__ oright,
                                int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    oext_off[i] = i;
    if (i == n) return;
    uint32_t s = perm[i];
    omarker[i] = marker[s]; oleft[i] = left[s]; oright[i] = right[s]; oext[i] = ext[s];
}

constexpr int SHORT_WORDS = 8;

// copy extension words record by record; records longer than SHORT_WORDS are queued as chunks of
// LONG_CHUNK words -- (record, chunk) items -- so that a 2.6 Mbp contig (84 K words) is copied by
// forty workgroups at once instead of one (84 us per sort in the late passes)
constexpr int LONG_CHUNK = 2048;
__global__ void k_gather_ext(const uint32_t *__restrict__ perm, int64_t n, const int64_t *__restrict__ ext_off,
                             const uint64_t *__restrict__ ext, const uint64_t *__restrict__ oext_off_u,
                             int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext,
                             uint64_t *__restrict__ long_list, unsigned long long *__restrict__ long_n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int64_t o = (int64_t)oext_off_u[i];
    oext_off[i] = o;
    if (i == n) return;
    uint32_t s = perm[i];
    int64_t b = ext_off[s], nw = ext_off[s + 1] - b;
    if (nw <= SHORT_WORDS) {
        for (int64_t w = 0; w < nw; w++) oext[o + w] = ext[b + w];
    } else {
        const unsigned long long nc = (unsigned long long)((nw + LONG_CHUNK - 1) / LONG_CHUNK);
        const unsigned long long at = atomicAdd(long_n, nc);
        for (unsigned long long c = 0; c < nc; c++) long_list[at + c] = ((uint64_t)c << 32) | (uint64_t)i;
    }
}

__global__ void k_gather_ext_long(const uint32_t *__restrict__ perm, const int64_t *__restrict__ ext_off,
                                  const uint64_t *__restrict__ ext, const int64_t *__restrict__ oext_off,
                                  uint64_t *__restrict__ oext, const uint64_t *__restrict__ long_list,
                                  const unsigned long long *__restrict__ long_n) {
    unsigned long long cnt = *long_n;
    for (unsigned long long e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint64_t item = long_list[e];
        const uint32_t i = (uint32_t)item;
        const int64_t c0 = (int64_t)(item >> 32) * LONG_CHUNK;
        const uint32_t s = perm[i];
        const int64_t b = ext_off[s], nw = ext_off[s + 1] - b, o = oext_off[i];
        const int64_t c1 = c0 + LONG_CHUNK < nw ? c0 + LONG_CHUNK : nw;
        for (int64_t w = c0 + threadIdx.x; w < c1; w += blockDim.x) oext[o + w] = ext[b + w];
    }
}

// Order contract B.0: start[p] = floor(p*n/P) moved forward so equal keys never split.
template <int KW>
__global__ void k_partition_starts(const KeyW<KW> *__restrict__ skey, int64_t n, int P,
                                   int64_t *__restrict__ start) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > P) return;
    if (p == P) { start[P] = n; return; }
    int64_t s = (int64_t)(((uint64_t)p * (uint64_t)n) / (uint64_t)P);   // n < 2^32, p < 2^31
    while (s > 0 && s < n && key_eq(skey[s], skey[s - 1])) s++;
    start[p] = s;
}

// ---- run heads
template <int KW>
__global__ void k_head_flags(const KeyW<KW> *__restrict__ key, int64_t n, uint32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || !key_eq(key[i], key[i - 1])) ? 1u : 0u;
}

// a-7: forward fork filter, one thread per run head
template <int KW>
__global__ void k_fork_forward(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                               const uint64_t *__restrict__ ext, const int32_t *__restrict__ left, int64_t n,
                               const uint32_t *__restrict__ flag, const uint64_t *__restrict__ pos,
                               int sub, int min_err, int ds_ec,
                               KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker,
                               int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext,
                               int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) { oext_off[pos[n]] = (int64_t)pos[n]; return; }
    if (i > n || !flag[i]) return;
    const KeyW<KW> kk = key[i];
    int32_t hm = marker[i], hc = left[i];
    uint64_t he = ext[i];
    int32_t hr = ds_ec ? (-1 - hc) : -1;                                   // :2478 / DS :3436 / 64 :10149
    for (int64_t j = i + 1; j < n && key_eq(key[j], kk); j++) {
        int32_t cs = left[j];
        if (cs > hc) {                                                     // :2483
            bool err = min_err != 0 && hc <= min_err && cs >= 2 * hc;      // :2484
            hm = marker[j]; he = ext[j]; hc = cs;
            hr = err ? (ds_ec ? (-1 - cs) : -1) : sub;
        } else if (cs == hc) {                                             // :2497
            if ((int64_t)ext[j] > (int64_t)he) { hm = marker[j]; he = ext[j]; }
            hr = sub;
        } else {                                                           // :2512
            bool err = min_err != 0 && cs <= min_err && hc >= 2 * cs;      // :2513
            hr = err ? (ds_ec ? (-1 - hc) : -1) : sub;
        }
    }
    uint64_t o = pos[i];
    okey[o] = kk; omarker[o] = hm; oext[o] = he; oleft[o] = hc; oright[o] = hr; oext_off[o] = (int64_t)o;
}

// a-9: reflected fork filter
template <int KW>
__global__ void k_fork_reflected(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                                 const uint64_t *__restrict__ ext, const int32_t *__restrict__ left,
                                 const int32_t *__restrict__ right, int64_t n,
                                 const uint32_t *__restrict__ flag, const uint64_t *__restrict__ pos,
                                 int sub, int min_err, int ds_ec,
                                 KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker,
                                 int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext,
                                 int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) { oext_off[pos[n]] = (int64_t)pos[n]; return; }
    if (i > n || !flag[i]) return;
    const KeyW<KW> kk = key[i];
    int32_t last_cov = left[i];                                            // HighCoverLastCoverage :2623
    int32_t hm = marker[i], hr = right[i];
    uint64_t he = ext[i];
    int32_t hl = ds_ec ? (-1 - last_cov) : -1;                             // :2626 / DS :3556 / 64 :10292
    for (int64_t j = i + 1; j < n && key_eq(key[j], kk); j++) {
        int32_t cs = left[j];
        if (cs > last_cov) {                                               // :2631
            bool err = min_err != 0 && last_cov <= min_err && cs >= 2 * last_cov;
            last_cov = cs;
            hm = marker[j]; he = ext[j]; hr = right[j];
            hl = err ? (ds_ec ? (-1 - cs) : -1) : sub;
        } else if (cs == last_cov) {                                       // :2647-2653
            int ls = sentinel_len(ext[j]), lh = sentinel_len(he);
            uint64_t a = ext[j] >> ((2 * (ls - 1)) & 63);
            uint64_t b = he >> ((2 * lh) & 63);
            if ((int64_t)a > (int64_t)b) { hm = marker[j]; he = ext[j]; hr = right[j]; }
            hl = sub;
        } else {                                                           // :2667
            bool err = min_err != 0 && cs <= min_err && last_cov >= 2 * cs;
            if (err) { if (!ds_ec) hl = -1; }                              // :2672 / DS :3595 keeps left
            else hl = sub;
        }
    }
    uint64_t o = pos[i];
    okey[o] = kk; omarker[o] = hm; oext[o] = he; oleft[o] = hl; oright[o] = hr; oext_off[o] = (int64_t)o;
}

__global__ void k_map_part_start(const int64_t *__restrict__ ps, int P, const uint64_t *__restrict__ pos,
                                 int64_t *__restrict__ ops) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= P) ops[p] = (int64_t)pos[ps[p]];
}

// a-8
template <int KW>
__global__ void k_reflect(const KeyW<KW> *__restrict__ key, const uint64_t *__restrict__ ext,
                          const int32_t *__restrict__ left, const int32_t *__restrict__ right, int64_t n, int sub,
                          KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker,
                          int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext,
                          int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) { oext_off[n] = n; return; }
    if (i > n) return;
    const KeyW<KW> kk = key[i];
    uint64_t first;
    KeyW<KW> nk;
    if (KW == 1) {
        first = (kk.w[0] >> (2 * (sub - 1))) & 3;                              // :2752-2754
        nk.w[0] = ((kk.w[0] << 2) & low_mask(sub)) | ext[i];                   // :2757-2758
    } else {
        // every word moves one base to the left, taking the top base of its right neighbour; the suffix base
        // enters the last word  (P/ReflexivDSMain64.java:10448-10466)
        const int res = sub - 31 * (KW - 1);
        first = kk.w[0] >> 60;                                                 // :10448
        uint64_t in = ext[i] & 3;
#pragma unroll
        for (int w = KW - 1; w >= 0; w--) {
            const int nb = w < KW - 1 ? 31 : res;
            const uint64_t top = kk.w[w] >> (2 * (nb - 1));
            nk.w[w] = ((kk.w[w] << 2) & low_mask(nb)) | in;
            in = top;
        }
    }
    okey[i] = nk;
    oext[i] = first | 4;                                                   // :2755 / :10450
    omarker[i] = 2; oleft[i] = left[i]; oright[i] = right[i]; oext_off[i] = i;
}

// partition of sorted position i: last p with ps[p] <= i
__device__ __forceinline__ int part_of(const int64_t *__restrict__ ps, int P, int64_t i) {
    int lo = 0, hi = P;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ps[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// a-10: single-word flip to the orientation the arrival index asks for
template <int KW>
__global__ void k_random_reflection(const KeyW<KW> *__restrict__ key, const int32_t *__restrict__ marker,
                                    const uint64_t *__restrict__ ext, const int32_t *__restrict__ left,
                                    const int32_t *__restrict__ right, int64_t n,
                                    const int64_t *__restrict__ ps, int P, int sub, const int32_t *__restrict__ carry,
                                    KeyW<KW> *__restrict__ okey, int32_t *__restrict__ omarker,
                                    int64_t *__restrict__ oext_off, uint64_t *__restrict__ oext,
                                    int32_t *__restrict__ oleft, int32_t *__restrict__ oright) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) { oext_off[n] = n; return; }
    if (i > n) return;
    int p = part_of(ps, P, i);
    // (several GPUs: a partition that began on an earlier rank brings the parity of its record count there, rfx_shard.hip)
    const int64_t cpar = (carry && p == carry[0]) ? (int64_t)carry[1] : 0;
    int m = ((i - ps[p] + cpar) & 1) ? 1 : 2;                               // :2777, :2880-2884
    KeyW<KW> kk = key[i];
    uint64_t e = ext[i];
    int mk = marker[i];
    if (mk != m) {
        int L = sentinel_len(e);
        uint64_t eb = e & low_mask(L);
        if (KW == 1) {
            unsigned __int128 S;
            if (mk == 1) {                      // key||ext  -> keyed at its last k-1 bases
                S = ((unsigned __int128)kk.w[0] << (2 * L)) | eb;
                kk.w[0] = (uint64_t)(S & (unsigned __int128)low_mask(sub));
                e = (uint64_t)(S >> (2 * sub)) | (1ULL << (2 * L));
            } else {                            // ext||key  -> keyed at its first k-1 bases
                S = ((unsigned __int128)eb << (2 * sub)) | kk.w[0];
                kk.w[0] = (uint64_t)(S >> (2 * L));
                e = (uint64_t)(S & (unsigned __int128)low_mask(L)) | (1ULL << (2 * L));
            }
        } else {
            // base-wise (P/ReflexivDSMain64.java:10527-10680): S = key||ext or ext||key, re-cut at the other end
            const KeyW<KW> src = kk;
            auto sb = [&](int t) -> unsigned {      // base t of S
                if (mk == 1) return t < sub ? key_base_w<KW>(src, sub, t) : (unsigned)((eb >> (2 * (L - 1 - (t - sub)))) & 3);
                return t < L ? (unsigned)((eb >> (2 * (L - 1 - t))) & 3) : key_base_w<KW>(src, sub, t - L);
            };
            uint64_t ne = 1;
            if (m == 1) {
                kk = build_key<KW>(sub, sb);
                for (int t = 0; t < L; t++) ne = (ne << 2) | sb(sub + t);
            } else {
                kk = build_key<KW>(sub, [&](int t) { return sb(L + t); });
                for (int t = 0; t < L; t++) ne = (ne << 2) | sb(t);
            }
            e = ne;
        }
    }
    okey[i] = kk; omarker[i] = m; oext[i] = e; oleft[i] = left[i]; oright[i] = right[i]; oext_off[i] = i;
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

int dev_records_alloc(rfx_ctx *ctx, DevRecords &r, int64_t cap_n, int64_t cap_words, int kw) {
    if (cap_n < 1) cap_n = 1;
    if (cap_words < 1) cap_words = 1;
    if (kw < 1 || kw > MAX_KEY_WORDS) return RFX_E_ARG;
    r.kw = kw;
    RFX_HIP(r.key.alloc((size_t)cap_n * 8 * kw, ctx->stream));
    RFX_HIP(r.marker.alloc((size_t)cap_n * 4, ctx->stream));
    RFX_HIP(r.ext_off.alloc((size_t)(cap_n + 1) * 8, ctx->stream));
    RFX_HIP(r.ext.alloc((size_t)cap_words * 8, ctx->stream));
    RFX_HIP(r.left.alloc((size_t)cap_n * 4, ctx->stream));
    RFX_HIP(r.right.alloc((size_t)cap_n * 4, ctx->stream));
    r.n = 0; r.words = 0;
    return RFX_OK;
}

int dev_records_upload(rfx_ctx *ctx, const rfx_records *h, DevRecords &d) {
    const int64_t n = h->n;
    const int64_t words = n > 0 ? h->ext_off[n] : 0;
    const int kw = h->key_words > 1 ? h->key_words : 1;
    RFX_TRY(dev_records_alloc(ctx, d, n, words, kw));
    if (n > 0) {
        RFX_HIP(hipMemcpyAsync(d.key.p, h->key, (size_t)n * 8 * kw, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.marker.p, h->marker, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.left.p, h->left, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.right.p, h->right, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        RFX_HIP(hipMemcpyAsync(d.ext_off.p, h->ext_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (words > 0)
            RFX_HIP(hipMemcpyAsync(d.ext.p, h->ext, (size_t)words * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        int64_t zero = 0;
        RFX_HIP(hipMemcpyAsync(d.ext_off.p, &zero, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    RFX_TRY(sync_checked(ctx));
    d.n = n; d.words = words;
    return RFX_OK;
}

int dev_records_download(rfx_ctx *ctx, const DevRecords &d, rfx_records *h) {
    h->need_n = d.n; h->need_words = d.words;
    if (d.n > h->cap_n || d.words > h->cap_words) return RFX_E_CAP;
    const int64_t n = d.n;
    h->n = n;
    h->key_words = d.kw;
    if (n > 0) {
        RFX_HIP(hipMemcpyAsync(h->key, d.key.p, (size_t)n * 8 * d.kw, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h->marker, d.marker.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h->left, d.left.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFX_HIP(hipMemcpyAsync(h->right, d.right.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (d.words > 0)
            RFX_HIP(hipMemcpyAsync(h->ext, d.ext.p, (size_t)d.words * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    RFX_HIP(hipMemcpyAsync(h->ext_off, d.ext_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RFX_TRY(sync_checked(ctx));
    return RFX_OK;
}

int rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n, int k,
                      DevRecords &out) {
    const int kw = sub_words(k);
    RFX_TRY(dev_records_alloc(ctx, out, 2 * n, 2 * n, kw));
    if (n > 0 && k <= 31) {
        hipLaunchKernelGGL(k_rc_expand, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_kmers, d_counts, n, k,
                           out.key.as<uint64_t>(), out.marker.as<int32_t>(), out.ext_off.as<int64_t>(),
                           out.ext.as<uint64_t>(), out.left.as<int32_t>(), out.right.as<int32_t>());
        RFX_HIP(hipGetLastError());
    } else if (n > 0) {
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_rc_expand_w<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_kmers,
                                             asm_words(k), d_counts, n, k, out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(),
                                             out.ext_off.as<int64_t>(), out.ext.as<uint64_t>(), out.left.as<int32_t>(),
                                             out.right.as<int32_t>()));
        RFX_HIP(hipGetLastError());
    } else {
        RFX_HIP(hipMemsetAsync(out.ext_off.p, 0, 8, ctx->stream));
    }
    out.n = 2 * n; out.words = 2 * n;
    return RFX_OK;
}

int sort_records(rfx_ctx *ctx, const DevRecords &in, int P, int key_bits, DevRecords &out, DevBuf &part_start, int k) {
    const int64_t n = in.n;
    const int kw = in.kw;
    if (n > (int64_t)0xFFFFFFFFLL) return RFX_E_LIMIT;
    if (kw > 1 && k != 0 && sub_words(k) != kw) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, in.words, kw));
    RFX_HIP(part_start.alloc((size_t)(P + 1) * 8, ctx->stream));
    DevBuf perm, tk, tv, nw, wscan, long_list, long_n;
    RFX_HIP(perm.alloc((size_t)(n ? n : 1) * 4, ctx->stream));
    RFX_HIP(tk.alloc((size_t)(n ? n : 1) * 8, ctx->stream));
    RFX_HIP(tv.alloc((size_t)(n ? n : 1) * 4, ctx->stream));
    RFX_HIP(nw.alloc((size_t)(n ? n : 1) * 4, ctx->stream));
    RFX_HIP(wscan.alloc((size_t)(n + 1) * 8, ctx->stream));
    // chunk items: at most one per record plus one per LONG_CHUNK words
    RFX_HIP(long_list.alloc((size_t)((n ? n : 1) + in.words / LONG_CHUNK + 1) * 8, ctx->stream));
    RFX_HIP(long_n.alloc(8, ctx->stream));
    RFX_HIP(hipMemsetAsync(long_n.p, 0, 8, ctx->stream));
    if (n > 0 && kw == 1) {
        RFX_HIP(hipMemcpyAsync(out.key.p, in.key.p, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n)), dim3(256), 0, ctx->stream, perm.as<uint32_t>(), n);
        RFX_HIP(hipGetLastError());
        RFX_TRY(sort_pairs(ctx, out.key.as<uint64_t>(), perm.as<uint32_t>(), n, key_bits, tk.as<uint64_t>(),
                           tv.as<uint32_t>()));
    } else if (n > 0) {
        // stable LSD over the key words, last word first, through the permutation: word w of the records in
        // their current order is gathered, sorted together with the permutation, and so on; then the whole
        // keys are gathered once
        DevBuf wk;
        RFX_HIP(wk.alloc((size_t)n * 8, ctx->stream));
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n)), dim3(256), 0, ctx->stream, perm.as<uint32_t>(), n);
        RFX_HIP(hipGetLastError());
        const int res = k > 0 ? (k - 1) - 31 * (kw - 1) : 31;      // k unknown: every word may use its 62 bits
        bool done = false;
        static const bool tie_off = getenv("RFX_SORT_TIEFIX") && atoi(getenv("RFX_SORT_TIEFIX")) == 0;
        if (kw == 2 && !tie_off) {       // (k unknown: res = 31, a prefix that splits less but orders the same)
            // one sort on the first 64 bits, then the runs of equal prefixes mended (see k_tie_fix2)
            DevBuf flag;
            RFX_HIP(flag.alloc(4, ctx->stream));
            RFX_HIP(hipMemsetAsync(flag.p, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_key_prefix2, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const KeyW<2> *)in.key.as<KeyW<2>>(), n, res,
                               wk.as<uint64_t>());
            RFX_HIP(hipGetLastError());
            const int pbits = 62 + 2 * res < 64 ? 62 + 2 * res : 64;
            RFX_TRY(sort_pairs(ctx, wk.as<uint64_t>(), perm.as<uint32_t>(), n, pbits, tk.as<uint64_t>(), tv.as<uint32_t>()));
            hipLaunchKernelGGL(k_tie_fix2, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const uint64_t *)wk.as<uint64_t>(),
                               perm.as<uint32_t>(), n, (const KeyW<2> *)in.key.as<KeyW<2>>(), res, flag.as<int>());
            RFX_HIP(hipGetLastError());
            int h_flag = 0;
            RFX_TRY(small_readback(ctx, &h_flag, flag.p, 4));
            done = h_flag == 0;
            if (!done) {                       // a long run of equal prefixes: start over with the two passes below
                hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n)), dim3(256), 0, ctx->stream, perm.as<uint32_t>(), n);
                RFX_HIP(hipGetLastError());
            }
        }
        for (int w = kw - 1; w >= 0 && !done; w--) {
            const uint32_t *pp = w == kw - 1 ? nullptr : perm.as<uint32_t>();
            RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_key_word<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                                                 (const KeyW<KW> *)in.key.as<KeyW<KW>>(), pp, n, w, wk.as<uint64_t>()));
            RFX_HIP(hipGetLastError());
            RFX_TRY(sort_pairs(ctx, wk.as<uint64_t>(), perm.as<uint32_t>(), n, 2 * (w == kw - 1 ? res : 31), tk.as<uint64_t>(),
                               tv.as<uint32_t>()));
        }
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_gather_key<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                                             (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const uint32_t *)perm.as<uint32_t>(), n,
                                             out.key.as<KeyW<KW>>()));
        RFX_HIP(hipGetLastError());
    }
    if (in.words == n) {
        // (ext_off is the identity on both sides)
        hipLaunchKernelGGL(k_gather_single, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream, (const uint32_t *)perm.as<uint32_t>(), n,
                           (const int32_t *)in.marker.as<int32_t>(), (const int32_t *)in.left.as<int32_t>(),
                           (const int32_t *)in.right.as<int32_t>(), (const uint64_t *)in.ext.as<uint64_t>(), out.marker.as<int32_t>(),
                           out.left.as<int32_t>(), out.right.as<int32_t>(), out.ext_off.as<int64_t>(), out.ext.as<uint64_t>());
        RFX_HIP(hipGetLastError());
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_partition_starts<KW>, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream,
                                             (const KeyW<KW> *)out.key.as<KeyW<KW>>(), n, P, part_start.as<int64_t>()));
        RFX_HIP(hipGetLastError());
        out.n = n; out.words = in.words;
        return RFX_OK;
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_gather_fixed, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                           (const uint32_t *)perm.as<uint32_t>(), n, (const int32_t *)in.marker.as<int32_t>(),
                           (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(),
                           (const int64_t *)in.ext_off.as<int64_t>(), out.marker.as<int32_t>(),
                           out.left.as<int32_t>(), out.right.as<int32_t>(), nw.as<uint32_t>());
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, nw.as<uint32_t>(), wscan.as<uint64_t>(), n));
    hipLaunchKernelGGL(k_gather_ext, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream,
                       (const uint32_t *)perm.as<uint32_t>(), n, (const int64_t *)in.ext_off.as<int64_t>(),
                       (const uint64_t *)in.ext.as<uint64_t>(), (const uint64_t *)wscan.as<uint64_t>(),
                       out.ext_off.as<int64_t>(), out.ext.as<uint64_t>(), long_list.as<uint64_t>(),
                       long_n.as<unsigned long long>());
    RFX_HIP(hipGetLastError());
    if (in.words > n) {     // some record has more than one word: a long one may exist
        hipLaunchKernelGGL(k_gather_ext_long, dim3(1024), dim3(256), 0, ctx->stream,
                           (const uint32_t *)perm.as<uint32_t>(), (const int64_t *)in.ext_off.as<int64_t>(),
                           (const uint64_t *)in.ext.as<uint64_t>(), (const int64_t *)out.ext_off.as<int64_t>(),
                           out.ext.as<uint64_t>(), (const uint64_t *)long_list.as<uint64_t>(),
                           (const unsigned long long *)long_n.as<unsigned long long>());
        RFX_HIP(hipGetLastError());
    }
    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_partition_starts<KW>, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream,
                                         (const KeyW<KW> *)out.key.as<KeyW<KW>>(), n, P, part_start.as<int64_t>()));
    RFX_HIP(hipGetLastError());
    out.n = n; out.words = in.words;
    return RFX_OK;
}

int fork_filter(rfx_ctx *ctx, bool reflected, const DevRecords &in, const int64_t *d_part_start, int P, int k,
                int min_error_cov, int twin, DevRecords &out, DevBuf &out_part_start) {
    const int64_t n = in.n;
    if (in.words != n) return RFX_E_ARG;             // single-word records only
    const int kw = in.kw;
    if (kw != sub_words(k)) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, n, kw));
    RFX_HIP(out_part_start.alloc((size_t)(P + 1) * 8, ctx->stream));
    DevBuf flag, pos;
    RFX_HIP(flag.alloc((size_t)(n ? n : 1) * 4, ctx->stream));
    RFX_HIP(pos.alloc((size_t)(n + 1) * 8, ctx->stream));
    if (n > 0) {
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_head_flags<KW>, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                                             (const KeyW<KW> *)in.key.as<KeyW<KW>>(), n, flag.as<uint32_t>()));
        RFX_HIP(hipGetLastError());
    }
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, flag.as<uint32_t>(), pos.as<uint64_t>(), n));
    const int ds_ec = (twin == RFX_TWIN_DS && min_error_cov != 0) ? 1 : 0;
    if (!reflected) {
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_fork_forward<KW>, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream,
                           (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                           (const uint64_t *)in.ext.as<uint64_t>(), (const int32_t *)in.left.as<int32_t>(), n,
                           (const uint32_t *)flag.as<uint32_t>(), (const uint64_t *)pos.as<uint64_t>(), k - 1,
                           min_error_cov, ds_ec, out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(),
                           out.ext_off.as<int64_t>(), out.ext.as<uint64_t>(), out.left.as<int32_t>(),
                           out.right.as<int32_t>()));
    } else {
        RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_fork_reflected<KW>, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream,
                           (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                           (const uint64_t *)in.ext.as<uint64_t>(), (const int32_t *)in.left.as<int32_t>(),
                           (const int32_t *)in.right.as<int32_t>(), n, (const uint32_t *)flag.as<uint32_t>(),
                           (const uint64_t *)pos.as<uint64_t>(), k - 1, min_error_cov, ds_ec,
                           out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(), out.ext_off.as<int64_t>(),
                           out.ext.as<uint64_t>(), out.left.as<int32_t>(), out.right.as<int32_t>()));
    }
    RFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_map_part_start, dim3(grid_for(P + 1)), dim3(256), 0, ctx->stream, d_part_start, P,
                       (const uint64_t *)pos.as<uint64_t>(), out_part_start.as<int64_t>());
    RFX_HIP(hipGetLastError());
    uint64_t m = 0;
    RFX_TRY(small_readback(ctx, &m, pos.as<uint64_t>() + n, 8));
    out.n = (int64_t)m; out.words = (int64_t)m;
    return RFX_OK;
}

int reflect_from_forward(rfx_ctx *ctx, const DevRecords &in, int k, DevRecords &out) {
    const int64_t n = in.n;
    if (in.words != n) return RFX_E_ARG;
    const int kw = in.kw;
    if (kw != sub_words(k)) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, n, kw));
    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_reflect<KW>, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream,
                       (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const uint64_t *)in.ext.as<uint64_t>(),
                       (const int32_t *)in.left.as<int32_t>(), (const int32_t *)in.right.as<int32_t>(), n, k - 1,
                       out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(), out.ext_off.as<int64_t>(),
                       out.ext.as<uint64_t>(), out.left.as<int32_t>(), out.right.as<int32_t>()));
    RFX_HIP(hipGetLastError());
    out.n = n; out.words = n;
    return RFX_OK;
}

int random_reflection(rfx_ctx *ctx, const DevRecords &in, const int64_t *d_part_start, int P, int k,
                      DevRecords &out, const int32_t *d_carry) {
    const int64_t n = in.n;
    if (in.words != n) return RFX_E_ARG;
    const int kw = in.kw;
    if (kw != sub_words(k)) return RFX_E_ARG;
    RFX_TRY(dev_records_alloc(ctx, out, n, n, kw));
    RFX_KW_SWITCH(kw, hipLaunchKernelGGL(k_random_reflection<KW>, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream,
                       (const KeyW<KW> *)in.key.as<KeyW<KW>>(), (const int32_t *)in.marker.as<int32_t>(),
                       (const uint64_t *)in.ext.as<uint64_t>(), (const int32_t *)in.left.as<int32_t>(),
                       (const int32_t *)in.right.as<int32_t>(), n, d_part_start, P, k - 1, d_carry,
                       out.key.as<KeyW<KW>>(), out.marker.as<int32_t>(), out.ext_off.as<int64_t>(),
                       out.ext.as<uint64_t>(), out.left.as<int32_t>(), out.right.as<int32_t>()));
    RFX_HIP(hipGetLastError());
    out.n = n; out.words = n;
    return RFX_OK;
}

// KmerBinarizer (P/ReflexivDSMain64.java:10772-10836) + filter(count >= min && count <= max) (:473-478)
int counter_to_asm(rfx_ctx *ctx, const uint64_t *d_keys32, const int64_t *d_counts64, int64_t n, int k, int min_cov,
                   int max_cov, uint64_t *d_out31, int32_t *d_out_counts, int64_t *out_n) {
    *out_n = 0;
    if (n == 0) return RFX_OK;
    DevBuf flag, pos;
    RFX_HIP(flag.alloc((size_t)n * 4, ctx->stream));
    RFX_HIP(pos.alloc((size_t)(n + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_counter_keep, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_counts64, n, min_cov, max_cov,
                       flag.as<uint32_t>());
    RFX_HIP(hipGetLastError());
    RFX_TRY(exclusive_scan_u32_to_u64(ctx, flag.as<uint32_t>(), pos.as<uint64_t>(), n));
    hipLaunchKernelGGL(k_counter_to_asm, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_keys32, d_counts64, n, k,
                       (const uint32_t *)flag.as<uint32_t>(), (const uint64_t *)pos.as<uint64_t>(), d_out31, d_out_counts);
    RFX_HIP(hipGetLastError());
    uint64_t m = 0;
    RFX_TRY(small_readback(ctx, &m, pos.as<uint64_t>() + n, 8));
    *out_n = (int64_t)m;
    return RFX_OK;
}

}  // namespace rfx
