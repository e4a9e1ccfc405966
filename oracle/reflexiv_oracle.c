/*
 * reflexiv_oracle.c -- CPU restatement of Reflexiv's fixed-k (k <= 31) assembly
 * hot path: k-mer extraction -> count/filter -> RC expand -> fork filters ->
 * reflexible extend-and-merge passes -> contigs.
 *
 * TEST INFRASTRUCTURE ONLY (see reflexiv_oracle.h).  Written from the Java of
 * the reference, cited per function as P/<file>:<lines>, with
 * P = src/main/java/uni/bielefeld/cmg/reflexiv/pipeline.  The flip/merge
 * arithmetic is restated at sequence level (unpack -> concatenate -> repack in
 * the reference's word layout); SURVEY.md C.9 records that for k = 31 this is
 * equal to the reference's bit code in every stage.
 *
 * Parity pin: docs/example.html:303,320-343 (tests/test_oracle_example.py).
 */
#include "reflexiv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

/* Threads for the CPU-baseline leg of bench.py ("OpenMP, one logical partition per
 * thread ~ local[N]").  1 (the default) keeps every function below strictly serial, which is what the
 * parity tests use; with more, the partition-wise operators run one logical partition per task -- the
 * serial function applied to that partition alone, exactly as a Spark task would -- sorts and gathers
 * split by index range, and the results are identical (tests/test_oracle_omp.py). */
static int g_threads = 1;
void orc_set_threads(int t) {
#ifdef _OPENMP
    g_threads = t < 1 ? 1 : t;
#else
    (void)t; g_threads = 1;
#endif
}
int orc_get_threads(void) { return g_threads; }
int orc_host_cores(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "reflexiv_oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}

/* Java Long.numberOfLeadingZeros: 64 for 0 */
static int nlz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

/* "Long.SIZE/2 - (Long.numberOfLeadingZeros(x)/2 + 1)"  P/ReflexivMain.java:2800 */
static int sentinel_len(uint64_t w) { return 32 - (nlz64(w) / 2 + 1); }

/* nucleotideValue  P/ReflexivMain.java:3062-3074: A0 C1 G2, anything else 3 */
static inline uint64_t nuc_value(char c) {
    if (c == 'A') return 0;
    if (c == 'C') return 1;
    if (c == 'G') return 2;
    return 3;
}

static inline uint64_t low_mask(int bases) {   /* ~((~0L) << 2*bases), bases <= 31 */
    return ~((~0ULL) << (2 * bases));
}

void orc_default_params(orc_params *p) {
    p->k = 31; p->min_cov = 2; p->max_cov = 10000000; p->min_error_cov = 8;
    p->min_contig = 500; p->min_iter = 15; p->max_iter = 150;
    p->front_clip = 0; p->end_clip = 0; p->partitions = 8;
    p->twin = ORC_TWIN_DS; p->coalesce = 0; p->extras = 1;
}

/* ---------------------------------------------------------- a-1 fastq filter */

int64_t orc_fastq_group(const char *text, int64_t len,
                        int64_t *seq_off, int32_t *seq_len, int64_t cap) {
    /* FastqFilterWithQual.call  P/ReflexivMain.java:3092-3112.  lineMark counts
     * the lines of the record being assembled; the checks are made in the
     * reference's order (lineMark==2, ==3, startsWith("@"), ==1). */
    int lineMark = 0;
    int64_t n = 0, cur_off = 0; int32_t cur_len = 0;
    int64_t pos = 0;
    while (pos < len) {
        int64_t e = pos;
        while (e < len && text[e] != '\n') e++;
        int64_t l = e - pos;
        if (l > 0 && text[e - 1] == '\r') l--;       /* textFile strips \r\n too */
        if (lineMark == 2) {
            lineMark++;
        } else if (lineMark == 3) {
            lineMark++;
            if (n < cap) { seq_off[n] = cur_off; seq_len[n] = cur_len; }
            n++;
        } else if (l > 0 && text[pos] == '@') {
            lineMark = 1;
        } else if (lineMark == 1) {
            cur_off = pos; cur_len = (int32_t)l;
            lineMark++;
        }
        pos = e + 1;
    }
    return n;
}

static int check_seq(char a) { return a == 'A' || a == 'T' || a == 'C' || a == 'G' || a == 'N'; }   /* :268-289 */

int64_t orc_fastq_only_seq(const char *text, int64_t len,
                           int64_t *seq_off, int32_t *seq_len, int64_t cap) {
    int64_t n = 0, pos = 0;
    while (pos < len) {
        int64_t e = pos;
        while (e < len && text[e] != '\n') e++;
        int64_t l = e - pos;
        if (l > 0 && text[e - 1] == '\r') l--;
        const char *s = text + pos;
        if (l > 20 && s[0] != '@' && s[0] != '+' &&                              /* :245-250 */
            check_seq(s[0]) && check_seq(s[4]) && check_seq(s[9]) && check_seq(s[14]) && check_seq(s[19])) {
            if (n < cap) { seq_off[n] = pos; seq_len[n] = (int32_t)l; }
            n++;
        }
        pos = e + 1;
    }
    return n;
}

/* -------------------------------------------------------------- a-2 extract */

int64_t orc_extract_canon(const char *bases, const int64_t *read_off, int64_t n_reads,
                          int k, int front_clip, int end_clip,
                          uint64_t *out, int64_t cap) {
    const uint64_t mask = low_mask(k);               /* maxKmerBits :3003 */
    int64_t n = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const char *read = bases + read_off[r];
        int64_t len = read_off[r + 1] - read_off[r];
        if (len - k - end_clip <= 1 || front_clip > len) continue;          /* :3020 */
        uint64_t fwd = 0, rc = 0;
        for (int64_t i = front_clip; i < len - end_clip; i++) {             /* :3027 */
            int64_t j = i - front_clip;
            uint64_t v = nuc_value(read[i]);
            fwd = (fwd << 2) | v;                                            /* :3032-3033 */
            if (j >= k) fwd &= mask;                                         /* :3034-3036 */
            uint64_t c = v ^ 3;                                              /* :3039 */
            if (j >= k) { rc >>= 2; c <<= 2 * (k - 1); }                     /* :3041-3043 */
            else        { c <<= 2 * j; }                                     /* :3045 */
            rc |= c;                                                         /* :3047 */
            if (j >= k - 1) {                                                /* :3050 */
                uint64_t canon = ((int64_t)fwd < (int64_t)rc) ? fwd : rc;    /* :3051-3055 */
                if (n < cap) out[n] = canon;
                n++;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------- a-2w extract, k > 31 */

int orc_words_w(int k) { return k / 32 + 1; }

/* compareLongArrayBlocks :652-687: base by base from the left; equal -> true (forward) */
static int fwd_not_after_rc(const uint64_t *f, const uint64_t *r, int W, int res) {
    for (int i = 0; i < W; i++) {
        const int nb = i < W - 1 ? 32 : res;
        for (int j = 0; j < nb; j++) {
            const unsigned a = (unsigned)(f[i] >> (2 * (nb - 1 - j))) & 3u;
            const unsigned b = (unsigned)(r[i] >> (2 * (nb - 1 - j))) & 3u;
            if (a < b) return 1;
            if (a > b) return 0;
        }
    }
    return 1;
}

int64_t orc_extract_canon_w(const char *bases, const int64_t *read_off, int64_t n_reads,
                            int k, int front_clip, int end_clip, uint64_t *out, int64_t cap) {
    const int W = k / 32 + 1, res = k % 32;             /* kmerBinarySlots, kmerSizeResidue */
    if (k <= 32 || res == 0 || W > 8) return -1;
    const uint64_t mask = ~((~0ULL) << (2 * res));      /* maxKmerBits :391 */
    int64_t n = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const char *read = bases + read_off[r];
        const int64_t len = read_off[r + 1] - read_off[r];
        if (len - k - end_clip + 1 <= 0 || front_clip > len) continue;       /* :410 */
        uint64_t acc = 0, racc = 0, f[8] = {0}, rc[8] = {0};
        for (int64_t i = front_clip; i < len - end_clip; i++) {              /* :419 */
            const int64_t j = i - front_clip;
            const uint64_t v = nuc_value(read[i]);
            if (j <= k - 1) {                                                /* :425-441 */
                acc = (acc << 2) | v;
                if ((j + 1) % 32 == 0) { f[(j + 1) / 32 - 1] = acc; acc = 0; }
                if (j == k - 1) { acc &= mask; f[(j + 1) / 32] = acc; acc = 0; }
            } else {                                                         /* :442-460 */
                uint64_t t1 = f[W - 1] >> (2 * (res - 1)), t2;
                f[W - 1] = ((f[W - 1] << 2) | v) & mask;
                for (int q = W - 2; q >= 0; q--) {
                    t2 = f[q] >> 62;
                    f[q] = (f[q] << 2) | t1;
                    t1 = t2;
                }
            }
            uint64_t c = v ^ 3;                                              /* :463 */
            if (j <= k - 1) {                                                /* :465-491 */
                if (j < res - 1) { racc |= c << (2 * j); }
                else if (j == res - 1) { racc |= c << (2 * j); rc[W - 1] = racc; racc = 0; }
                else if ((j - res + 1) % 32 == 0) {
                    racc |= c << (2 * ((j - res) % 32));
                    rc[W - (j - res + 1) / 32 - 1] = racc; racc = 0;
                } else { racc |= c << (2 * ((j - res) % 32)); }
            } else {                                                         /* :492-512 */
                uint64_t t1 = rc[0] << 62, t2;
                rc[0] = (rc[0] >> 2) | (c << 62);
                for (int q = 1; q < W - 1; q++) {
                    t2 = rc[q] << 62;
                    rc[q] = (rc[q] >> 2) | t1;
                    t1 = t2;
                }
                rc[W - 1] >>= 2;
                rc[W - 1] |= t1 >> (2 * (31 - res + 1));
            }
            if (j >= k - 1) {                                                /* :524, :609-627 */
                const uint64_t *src = fwd_not_after_rc(f, rc, W, res) ? f : rc;
                if (n < cap) for (int q = 0; q < W; q++) out[n * W + q] = src[q];
                n++;
            }
        }
    }
    return n;
}

static int g_cmp_w = 1;
static int cmp_words(const void *a, const void *b) {
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    for (int i = 0; i < g_cmp_w; i++) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
    }
    return 0;
}

int64_t orc_count_filter_w(uint64_t *kmers, int64_t n, int k, int min_cov, int max_cov,
                           uint64_t *out_keys, int64_t *out_counts, int64_t cap, int64_t *n_distinct) {
    const int W = k / 32 + 1;
    g_cmp_w = W;
    qsort(kmers, (size_t)n, (size_t)W * 8, cmp_words);
    int64_t m = 0, d = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i + 1;
        while (j < n && cmp_words(kmers + i * W, kmers + j * W) == 0) j++;
        const int64_t c = j - i;
        d++;
        int keep = 1;
        if (min_cov > 1 && c < min_cov) keep = 0;                            /* :197-200 */
        if (max_cov < 10000000 && c > max_cov) keep = 0;                     /* :202-205 */
        if (keep) {
            if (m < cap) {
                for (int q = 0; q < W; q++) out_keys[m * W + q] = kmers[i * W + q];
                out_counts[m] = c;
            }
            m++;
        }
        i = j;
    }
    if (n_distinct) *n_distinct = d;
    return m;
}

void orc_kmer_text_w(const uint64_t *kmer, int k, char *out) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    const int res = k % 32;
    for (int i = 0; i < (k / 32) * 32; i++) out[i] = nt[(kmer[i / 32] >> (2 * (31 - i % 32))) & 3];      /* :346-352 */
    for (int i = (k / 32) * 32; i < k; i++) out[i] = nt[(kmer[i / 32] >> (2 * (res - 1 - i % 32))) & 3];  /* :354-360 */
}

/* ---- a-2w + count + filter over all host cores, k > 32, W = k/32+1 = 2 words (k = 33..63), with the same range
 * restriction: buckets = top 12 bits of word 0.  Same output as orc_extract_canon_w + orc_count_filter_w
 * (tests/test_oracle_omp.py); used by tests/golden/make_c2_full.py for the full-size k = 63 pin. */
typedef struct { uint64_t w0, w1; } kw2;

static inline void read_kmers_w2(const char *read, int64_t len, int k, int front_clip, int end_clip,
                                 int b_lo, int b_hi, int64_t *hist, kw2 *arr, int64_t *cursor) {
    const int W = 2, res = k % 32;
    const uint64_t mask = ~((~0ULL) << (2 * res));
    if (len - k - end_clip + 1 <= 0 || front_clip > len) return;             /* :410 */
    uint64_t acc = 0, racc = 0, f[2] = {0, 0}, rc[2] = {0, 0};
    for (int64_t i = front_clip; i < len - end_clip; i++) {                  /* :419 (the loop of orc_extract_canon_w) */
        const int64_t j = i - front_clip;
        const uint64_t v = nuc_value(read[i]);
        if (j <= k - 1) {
            acc = (acc << 2) | v;
            if ((j + 1) % 32 == 0) { f[(j + 1) / 32 - 1] = acc; acc = 0; }
            if (j == k - 1) { acc &= mask; f[(j + 1) / 32] = acc; acc = 0; }
        } else {
            uint64_t t1 = f[W - 1] >> (2 * (res - 1));
            f[W - 1] = ((f[W - 1] << 2) | v) & mask;
            f[0] = (f[0] << 2) | t1;
        }
        uint64_t c = v ^ 3;
        if (j <= k - 1) {
            if (j < res - 1) { racc |= c << (2 * j); }
            else if (j == res - 1) { racc |= c << (2 * j); rc[W - 1] = racc; racc = 0; }
            else if ((j - res + 1) % 32 == 0) { racc |= c << (2 * ((j - res) % 32)); rc[W - (j - res + 1) / 32 - 1] = racc; racc = 0; }
            else { racc |= c << (2 * ((j - res) % 32)); }
        } else {
            uint64_t t1 = rc[0] << 62;
            rc[0] = (rc[0] >> 2) | (c << 62);
            rc[W - 1] >>= 2;
            rc[W - 1] |= t1 >> (2 * (31 - res + 1));
        }
        if (j >= k - 1) {
            const uint64_t *src = fwd_not_after_rc(f, rc, W, res) ? f : rc;
            const int bkt = (int)(src[0] >> 52);
            if (bkt < b_lo || bkt >= b_hi) continue;
            if (arr) { kw2 e = { src[0], src[1] }; arr[cursor[bkt]++] = e; } else hist[bkt]++;
        }
    }
}

/* LSD radix sort of 16-byte elements by (w0 low bits, w1), 11-bit digits */
static void radix_sort_kw2(kw2 *a, int64_t n, int bits0, int bits1, kw2 *tmp) {
    if (n < 2) return;
    int64_t hist[2048];
    kw2 *src = a, *dst = tmp;
    for (int word = 1; word >= 0; word--) {
        const int bits = word ? bits1 : bits0;
        for (int sh = 0; sh < bits; sh += 11) {
            memset(hist, 0, sizeof hist);
            if (word) { for (int64_t i = 0; i < n; i++) hist[(src[i].w1 >> sh) & 2047]++; }
            else      { for (int64_t i = 0; i < n; i++) hist[(src[i].w0 >> sh) & 2047]++; }
            const uint64_t d0 = word ? (src[0].w1 >> sh) & 2047 : (src[0].w0 >> sh) & 2047;
            if (hist[d0] == n) continue;
            int64_t s = 0;
            for (int d = 0; d < 2048; d++) { int64_t c = hist[d]; hist[d] = s; s += c; }
            if (word) { for (int64_t i = 0; i < n; i++) dst[hist[(src[i].w1 >> sh) & 2047]++] = src[i]; }
            else      { for (int64_t i = 0; i < n; i++) dst[hist[(src[i].w0 >> sh) & 2047]++] = src[i]; }
            kw2 *t = src; src = dst; dst = t;
        }
    }
    if (src != a) memcpy(a, src, (size_t)n * sizeof(kw2));
}

int64_t orc_count_reads_w2_range_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                                     int k, int front_clip, int end_clip, int min_cov, int max_cov,
                                     int b_lo, int b_hi,
                                     uint64_t *out_keys, int64_t *out_counts, int64_t cap,
                                     int64_t *n_distinct, int64_t *n_instances) {
    if (k <= 32 || k >= 64) return -1;
    const int T = g_threads, NB = 4096, res = k % 32;
    int64_t *hist = (int64_t *)xmalloc((size_t)T * NB * sizeof(int64_t));
    memset(hist, 0, (size_t)T * NB * sizeof(int64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n_reads * t / T, hi = n_reads * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++)
            read_kmers_w2(bases + read_off[r], read_off[r + 1] - read_off[r], k, front_clip, end_clip, b_lo, b_hi,
                          hist + (size_t)t * NB, NULL, NULL);
    }
    int64_t *bstart = (int64_t *)xmalloc((size_t)(NB + 1) * sizeof(int64_t));
    int64_t N = 0;
    for (int b = 0; b < NB; b++) {
        bstart[b] = N;
        for (int t = 0; t < T; t++) { int64_t c = hist[(size_t)t * NB + b]; hist[(size_t)t * NB + b] = N; N += c; }
    }
    bstart[NB] = N;
    if (n_instances) *n_instances = N;
    kw2 *arr = (kw2 *)xmalloc((size_t)(N ? N : 1) * sizeof(kw2));
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n_reads * t / T, hi = n_reads * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++)
            read_kmers_w2(bases + read_off[r], read_off[r + 1] - read_off[r], k, front_clip, end_clip, b_lo, b_hi,
                          NULL, arr, hist + (size_t)t * NB);
    }
    int64_t *bm = (int64_t *)xmalloc((size_t)(NB + 1) * sizeof(int64_t)), *bd = (int64_t *)xmalloc((size_t)NB * sizeof(int64_t));
    int64_t maxb = 1;
    for (int b = 0; b < NB; b++) if (bstart[b + 1] - bstart[b] > maxb) maxb = bstart[b + 1] - bstart[b];
    kw2 *scratch = (kw2 *)xmalloc((size_t)T * (size_t)maxb * sizeof(kw2));
    for (int sweep = 0; sweep < 2; sweep++) {
        if (sweep == 1) {
            int64_t M = 0, D = 0;
            for (int b = 0; b < NB; b++) { int64_t c = bm[b]; bm[b] = M; M += c; D += bd[b]; }
            bm[NB] = M;
            if (n_distinct) *n_distinct = D;
            if (M > cap) break;
        }
#ifdef _OPENMP
#pragma omp parallel for num_threads(T) schedule(dynamic, 4)
#endif
        for (int b = 0; b < NB; b++) {
            kw2 *a = arr + bstart[b];
            const int64_t n = bstart[b + 1] - bstart[b];
            if (sweep == 0) {
#ifdef _OPENMP
                radix_sort_kw2(a, n, 52, 2 * res, scratch + (size_t)omp_get_thread_num() * (size_t)maxb);
#else
                radix_sort_kw2(a, n, 52, 2 * res, scratch);
#endif
            }
            int64_t m = 0, d = 0, o = sweep ? bm[b] : 0;
            for (int64_t i = 0; i < n;) {
                int64_t j = i + 1;
                while (j < n && a[j].w0 == a[i].w0 && a[j].w1 == a[i].w1) j++;
                const int64_t c = j - i;
                d++;
                int keep = 1;
                if (min_cov > 1 && c < min_cov) keep = 0;                    /* P/ReflexivDataFrameCounter64.java:197-200 */
                if (max_cov < 10000000 && c > max_cov) keep = 0;             /* :202-205 */
                if (keep) {
                    if (sweep) { out_keys[2 * o] = a[i].w0; out_keys[2 * o + 1] = a[i].w1; out_counts[o] = c; o++; }
                    m++;
                }
                i = j;
            }
            if (!sweep) { bm[b] = m; bd[b] = d; }
        }
    }
    const int64_t M = bm[NB];
    free(hist); free(bstart); free(arr); free(bm); free(bd); free(scratch);
    return M;
}

/* ---------------------------------------------------------- radix sort u64 */

static void radix_sort_u64(uint64_t *a, int64_t n) {
    if (n < 2) return;
    uint64_t *tmp = (uint64_t *)xmalloc((size_t)n * 8);
    int64_t *hist = (int64_t *)xmalloc(65536 * sizeof(int64_t));
    uint64_t *src = a, *dst = tmp;
    for (int pass = 0; pass < 4; pass++) {
        int sh = 16 * pass;
        memset(hist, 0, 65536 * sizeof(int64_t));
        for (int64_t i = 0; i < n; i++) hist[(src[i] >> sh) & 0xFFFF]++;
        if (hist[(src[0] >> sh) & 0xFFFF] == n) continue;   /* all equal digit */
        int64_t s = 0;
        for (int d = 0; d < 65536; d++) { int64_t c = hist[d]; hist[d] = s; s += c; }
        for (int64_t i = 0; i < n; i++) dst[hist[(src[i] >> sh) & 0xFFFF]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, (size_t)n * 8);
    free(tmp); free(hist);
}

/* LSD radix sort of the low `bits` bits, 11-bit digits, caller's scratch (the buckets of
 * orc_count_reads_omp: the top bits are equal inside a bucket) */
static void radix_sort_low(uint64_t *a, int64_t n, int bits, uint64_t *tmp) {
    if (n < 2) return;
    int64_t hist[2048];
    uint64_t *src = a, *dst = tmp;
    for (int sh = 0; sh < bits; sh += 11) {
        memset(hist, 0, sizeof hist);
        for (int64_t i = 0; i < n; i++) hist[(src[i] >> sh) & 2047]++;
        if (hist[(src[0] >> sh) & 2047] == n) continue;
        int64_t s = 0;
        for (int d = 0; d < 2048; d++) { int64_t c = hist[d]; hist[d] = s; s += c; }
        for (int64_t i = 0; i < n; i++) dst[hist[(src[i] >> sh) & 2047]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, (size_t)n * 8);
}

/* ----------------------------------------------------- a-3/a-4 count, filter */

int64_t orc_count_filter(uint64_t *kmers, int64_t n, int min_cov, int max_cov, int twin,
                         uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                         int64_t *n_distinct) {
    radix_sort_u64(kmers, n);
    /* the RDD twin filters only when minKmerCoverage > 1  P/ReflexivMain.java:160;
     * the DS twin always does  P/ReflexivDSMain.java:211-216 */
    int apply = !(twin == ORC_TWIN_RDD && min_cov <= 1);
    int64_t m = 0, d = 0, i = 0;
    while (i < n) {
        int64_t j = i + 1;
        while (j < n && kmers[j] == kmers[i]) j++;
        int64_t c64 = j - i;
        int32_t c = (int32_t)c64;                      /* i1 + i2 on Integer :2897 */
        d++;
        if (!apply || (c >= min_cov && c <= max_cov)) {                     /* :3117 */
            if (m < cap) { out_keys[m] = kmers[i]; out_counts[m] = c; }
            m++;
        }
        i = j;
    }
    if (n_distinct) *n_distinct = d;
    return m;
}

/* ---- a-2 + a-3 + a-4 over all host cores (bench.py's cpu_baseline) ----
 * The same operators as orc_extract_canon + orc_count_filter, organised the way Spark's local[N] runs
 * them: every thread extracts the canonical k-mers of its own slice of the reads (one map task per
 * slice) straight into range buckets of the k-mer space (the shuffle write; bucket = top 12 bits of the
 * k-mer), then buckets are sorted, run-length counted and filtered by whichever thread is free (the
 * reduce side of reduceByKey, P/ReflexivMain.java:155, 2895-2899, 3115-3119).  Buckets are ranges, so their
 * concatenation is the ascending order of the order contract.  Result identical to the serial pair. */
int64_t orc_count_reads_range_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                                  int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                                  int b_lo, int b_hi,
                                  uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                                  int64_t *n_distinct, int64_t *n_instances);

static inline void read_kmers(const char *read, int64_t len, int k, int front_clip, int end_clip, uint64_t mask,
                              int shift, int b_lo, int b_hi, int64_t *hist, uint64_t *arr, int64_t *cursor) {
    if (len - k - end_clip <= 1 || front_clip > len) return;                 /* :3020 */
    uint64_t fwd = 0, rc = 0;
    for (int64_t i = front_clip; i < len - end_clip; i++) {                  /* :3027 */
        int64_t j = i - front_clip;
        uint64_t v = nuc_value(read[i]);
        fwd = (fwd << 2) | v;                                                /* :3032-3033 */
        if (j >= k) fwd &= mask;                                             /* :3034-3036 */
        uint64_t c = v ^ 3;                                                  /* :3039 */
        if (j >= k) { rc >>= 2; c <<= 2 * (k - 1); }                         /* :3041-3043 */
        else        { c <<= 2 * j; }                                         /* :3045 */
        rc |= c;                                                             /* :3047 */
        if (j >= k - 1) {                                                    /* :3050 */
            uint64_t canon = ((int64_t)fwd < (int64_t)rc) ? fwd : rc;        /* :3051-3055 */
            const int bkt = (int)(canon >> shift);
            if (bkt < b_lo || bkt >= b_hi) continue;         /* another pass's share of the k-mer space */
            if (arr) arr[cursor[bkt]++] = canon; else hist[bkt]++;
        }
    }
}

int64_t orc_count_reads_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                            int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                            uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                            int64_t *n_distinct, int64_t *n_instances) {
    return orc_count_reads_range_omp(bases, read_off, n_reads, k, front_clip, end_clip, min_cov, max_cov, twin, 0, 1 << 30,
                                     out_keys, out_counts, cap, n_distinct, n_instances);
}

/* the same, restricted to the range buckets [b_lo, b_hi) of the 4096 (top 12 bits of the k-mer): a read set whose
 * instances do not fit in memory at once is counted in several passes over disjoint shares of the k-mer space;
 * the passes' outputs concatenate to the full ascending list (tests/golden/make_c2_full.py) */
int64_t orc_count_reads_range_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                                  int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                                  int b_lo, int b_hi,
                                  uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                                  int64_t *n_distinct, int64_t *n_instances) {
    const int T = g_threads;
    const int BB = k >= 6 ? 12 : 2 * k;                 /* bucket bits */
    const int NB = 1 << BB, shift = 2 * k - BB;
    const uint64_t mask = low_mask(k);
    int64_t *hist = (int64_t *)xmalloc((size_t)T * NB * sizeof(int64_t));
    memset(hist, 0, (size_t)T * NB * sizeof(int64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n_reads * t / T, hi = n_reads * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++)
            read_kmers(bases + read_off[r], read_off[r + 1] - read_off[r], k, front_clip, end_clip, mask, shift, b_lo, b_hi,
                       hist + (size_t)t * NB, NULL, NULL);
    }
    int64_t *bstart = (int64_t *)xmalloc((size_t)(NB + 1) * sizeof(int64_t));
    int64_t N = 0;
    for (int b = 0; b < NB; b++) {
        bstart[b] = N;
        for (int t = 0; t < T; t++) { int64_t c = hist[(size_t)t * NB + b]; hist[(size_t)t * NB + b] = N; N += c; }
    }
    bstart[NB] = N;
    if (n_instances) *n_instances = N;
    uint64_t *arr = (uint64_t *)xmalloc((size_t)(N ? N : 1) * 8);
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n_reads * t / T, hi = n_reads * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++)
            read_kmers(bases + read_off[r], read_off[r + 1] - read_off[r], k, front_clip, end_clip, mask, shift, b_lo, b_hi,
                       NULL, arr, hist + (size_t)t * NB);
    }
    /* reduce side: sort + run-length + filter per bucket; two sweeps so that survivors land in order */
    const int apply = !(twin == ORC_TWIN_RDD && min_cov <= 1);                /* :160 vs DS :211-216 */
    int64_t *bm = (int64_t *)xmalloc((size_t)(NB + 1) * sizeof(int64_t)), *bd = (int64_t *)xmalloc((size_t)NB * sizeof(int64_t));
    int64_t maxb = 1;
    for (int b = 0; b < NB; b++) if (bstart[b + 1] - bstart[b] > maxb) maxb = bstart[b + 1] - bstart[b];
    uint64_t *scratch = (uint64_t *)xmalloc((size_t)T * (size_t)maxb * 8);
#ifdef _OPENMP
#pragma omp parallel for num_threads(T) schedule(dynamic, 4)
#endif
    for (int b = 0; b < NB; b++) {
        uint64_t *a = arr + bstart[b];
        const int64_t n = bstart[b + 1] - bstart[b];
#ifdef _OPENMP
        radix_sort_low(a, n, shift, scratch + (size_t)omp_get_thread_num() * (size_t)maxb);
#else
        radix_sort_low(a, n, shift, scratch);
#endif
        int64_t m = 0, d = 0;
        for (int64_t i = 0; i < n;) {
            int64_t j = i + 1;
            while (j < n && a[j] == a[i]) j++;
            const int32_t c = (int32_t)(j - i);                              /* i1 + i2 on Integer :2897 */
            d++;
            if (!apply || (c >= min_cov && c <= max_cov)) m++;               /* :3117 */
            i = j;
        }
        bm[b] = m; bd[b] = d;
    }
    int64_t M = 0, D = 0;
    for (int b = 0; b < NB; b++) { int64_t c = bm[b]; bm[b] = M; M += c; D += bd[b]; }
    bm[NB] = M;
    if (n_distinct) *n_distinct = D;
    if (M <= cap) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(T) schedule(dynamic, 4)
#endif
        for (int b = 0; b < NB; b++) {
            const uint64_t *a = arr + bstart[b];
            const int64_t n = bstart[b + 1] - bstart[b];
            int64_t o = bm[b];
            for (int64_t i = 0; i < n;) {
                int64_t j = i + 1;
                while (j < n && a[j] == a[i]) j++;
                const int32_t c = (int32_t)(j - i);
                if (!apply || (c >= min_cov && c <= max_cov)) { out_keys[o] = a[i]; out_counts[o] = c; o++; }
                i = j;
            }
        }
    }
    free(hist); free(bstart); free(arr); free(bm); free(bd); free(scratch);
    return M;
}

/* ------------------------------------------------------------ a-5/a-6 expand */

static void pack_key_fwd(const uint8_t *b, int kw, int sub, uint64_t *key);

uint64_t orc_revcomp(uint64_t kmer, int k) {
    /* KmerReverseComplement.call  P/ReflexivMain.java:2916-2923 */
    uint64_t rc = 0;
    for (int i = 0; i < k; i++) {
        rc <<= 2;
        rc |= (kmer & 3) ^ 3;
        kmer >>= 2;
    }
    return rc;
}

void orc_rc_expand_subkmer(const uint64_t *kmers, const int32_t *counts, int64_t n, int k,
                           uint64_t *key, int32_t *marker, uint64_t *ext,
                           int32_t *left, int32_t *right) {
    for (int64_t i = 0; i < n; i++) {
        uint64_t two[2] = { kmers[i], orc_revcomp(kmers[i], k) };           /* :2925-2926 */
        for (int t = 0; t < 2; t++) {
            int64_t o = 2 * i + t;
            key[o] = two[t] >> 2;                                            /* :2720 */
            ext[o] = two[t] & 3;                                             /* :2719 */
            marker[o] = 1; left[o] = counts[i]; right[o] = counts[i];        /* :2724 */
        }
    }
}

/* ---- k > 31: the assembler's 31-bases-per-word layout  P/ReflexivDSMain64.java */

int orc_sub_words(int k) { return k <= 32 ? 1 : (k - 2) / 31 + 1; }    /* subKmerBinarySlots    U/DefaultParam.java:94  */
int orc_asm_words(int k) { return k <= 31 ? 1 : (k - 1) / 31 + 1; }    /* kmerBinarySlotsAssemble  U/DefaultParam.java:85 */

/* KmerBinarizer.call :10772-10836 on one CSV row: the k-mer text (an optional leading '(' is
 * dropped, :10790-10792) and the count text (an optional trailing ')' is dropped; 10 or more digits
 * read as 1000000000, :10794-10806).  words: (k-1)/31+1, 31 bases each, the last one the rest
 * (:10812-10819).  Returns 0, or -1 when the text is shorter than k. */
int orc_kmer_binarize_w(const char *kmer_text, const char *count_text, int k, uint64_t *words, int32_t *cover) {
    if (kmer_text[0] == '(') kmer_text++;
    size_t cl = strlen(count_text);
    if (cl && count_text[cl - 1] == ')') *cover = cl >= 11 ? 1000000000 : (int32_t)strtol(count_text, NULL, 10);
    else *cover = cl >= 10 ? 1000000000 : (int32_t)strtol(count_text, NULL, 10);
    if ((int)strlen(kmer_text) < k) return -1;
    const int W = (k - 1) / 31 + 1;
    for (int w = 0; w < W; w++) words[w] = 0;
    for (int i = 0; i < k; i++) {
        words[i / 31] <<= 2;
        words[i / 31] |= nuc_value(kmer_text[i]);
    }
    return 0;
}

/* What the text round trip counter -> CSV -> KmerBinarizer does to a k-mer
 * (DSBinaryKmerToString P/ReflexivDataFrameCounter64.java:340-369, then the above): from k/32+1
 * words of 32 bases to (k-1)/31+1 words of 31. */
void orc_counter_to_asm_w(const uint64_t *kmers32, int64_t n, int k, uint64_t *kmers31) {
    const int W32 = k / 32 + 1, W31 = (k - 1) / 31 + 1;
    char *txt = (char *)xmalloc((size_t)k + 1);
    for (int64_t i = 0; i < n; i++) {
        orc_kmer_text_w(kmers32 + i * W32, k, txt);
        txt[k] = 0;
        int32_t c;
        orc_kmer_binarize_w(txt, "1", k, kmers31 + i * W31, &c);
    }
    free(txt);
}

/* DSKmerReverseComplement.call :10706-10755 + DSForwardSubKmerExtraction.call :10363-10403:
 * n (k-mer, count) -> 2n records (k-mer then its reverse complement), key = the first k-1 bases in
 * (k-2)/31+1 words, ext = the last base (no sentinel), marker 1, left = right = count. */
void orc_rc_expand_subkmer_w(const uint64_t *kmers, const int32_t *counts, int64_t n, int k,
                             uint64_t *key, int32_t *marker, uint64_t *ext,
                             int32_t *left, int32_t *right) {
    if (k <= 31) { orc_rc_expand_subkmer(kmers, counts, n, k, key, marker, ext, left, right); return; }
    const int W = (k - 1) / 31 + 1, kw = orc_sub_words(k), sub = k - 1;
    const int lastb = k - 31 * (W - 1);               /* bases in the last k-mer word */
    uint8_t *b = (uint8_t *)xmalloc((size_t)k + 8), *r = (uint8_t *)xmalloc((size_t)k + 8);
    for (int64_t i = 0; i < n; i++) {
        const uint64_t *km = kmers + (size_t)i * W;
        int o = 0;
        for (int w = 0; w < W; w++) {
            const int nb = w < W - 1 ? 31 : lastb;
            for (int j = 0; j < nb; j++) b[o++] = (uint8_t)((km[w] >> (2 * (nb - 1 - j))) & 3);
        }
        for (int j = 0; j < k; j++) r[j] = (uint8_t)(b[k - 1 - j] ^ 3);     /* :10727-10743 */
        const uint8_t *two[2] = { b, r };                                    /* :10749-10750 */
        for (int t = 0; t < 2; t++) {
            int64_t q = 2 * i + t;
            pack_key_fwd(two[t], kw, sub, key + (size_t)q * kw);              /* :10381-10395 */
            ext[q] = two[t][k - 1];                                          /* :10383 / :10390 */
            marker[q] = 1; left[q] = counts[i]; right[q] = counts[i];        /* :10399 */
        }
    }
    free(b); free(r);
}

/* ------------------------------------------------------------ order contract */

/* keys are kw consecutive words per record (kw = 1 for k <= 32): a Spark sort on an
 * array<long> column of equal-length arrays compares element by element, which for 31-base
 * words is the order of the base strings */
static inline int key_eq(const uint64_t *a, const uint64_t *b, int kw) {
    for (int w = 0; w < kw; w++) if (a[w] != b[w]) return 0;
    return 1;
}

#ifdef _OPENMP
/* the same stable LSD radix sort with the index range cut into T chunks: per-chunk histograms, offsets
 * in (digit, chunk) order, every chunk scatters its own elements in order -> the same permutation */
static void sort_perm_par(const uint64_t *key, int64_t n, int kw, int64_t *perm, int T) {
    int64_t *tmp = (int64_t *)xmalloc((size_t)n * sizeof(int64_t));
    int64_t *hist = (int64_t *)xmalloc((size_t)T * 65536 * sizeof(int64_t));
#pragma omp parallel for num_threads(T) schedule(static)
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    int64_t *src = perm, *dst = tmp;
    for (int w = kw - 1; w >= 0; w--) {
        for (int pass = 0; pass < 4; pass++) {
            const int sh = 16 * pass;
            int same = 0;
#pragma omp parallel num_threads(T)
            {
                const int t = omp_get_thread_num();
                const int64_t lo = n * t / T, hi = n * (t + 1) / T;
                int64_t *h = hist + (size_t)t * 65536;
                memset(h, 0, 65536 * sizeof(int64_t));
                for (int64_t i = lo; i < hi; i++) h[(key[src[i] * kw + w] >> sh) & 0xFFFF]++;
#pragma omp barrier
#pragma omp single
                {
                    int64_t tot0 = 0;
                    const int d0 = (int)((key[src[0] * kw + w] >> sh) & 0xFFFF);
                    for (int q = 0; q < T; q++) tot0 += hist[(size_t)q * 65536 + d0];
                    same = tot0 == n;
                    if (!same) {
                        int64_t sum = 0;
                        for (int d = 0; d < 65536; d++)
                            for (int q = 0; q < T; q++) { int64_t c = hist[(size_t)q * 65536 + d]; hist[(size_t)q * 65536 + d] = sum; sum += c; }
                    }
                }
                if (!same) for (int64_t i = lo; i < hi; i++) dst[h[(key[src[i] * kw + w] >> sh) & 0xFFFF]++] = src[i];
            }
            if (!same) { int64_t *x = src; src = dst; dst = x; }
        }
    }
    if (src != perm) memcpy(perm, src, (size_t)n * sizeof(int64_t));
    free(tmp); free(hist);
}
#endif

void orc_sort_perm_w(const uint64_t *key, int64_t n, int kw, int64_t *perm) {
    /* stable LSD radix sort of (key, index), last word first; ties keep arrival order (B.0) */
#ifdef _OPENMP
    if (g_threads > 1 && n >= (1 << 18)) { sort_perm_par(key, n, kw, perm, g_threads); return; }
#endif
    int64_t *tmp = (int64_t *)xmalloc((size_t)(n ? n : 1) * sizeof(int64_t));
    int64_t *hist = (int64_t *)xmalloc(65536 * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    int64_t *src = perm, *dst = tmp;
    for (int w = kw - 1; w >= 0 && n > 1; w--) {
        for (int pass = 0; pass < 4; pass++) {
            int sh = 16 * pass;
            memset(hist, 0, 65536 * sizeof(int64_t));
            for (int64_t i = 0; i < n; i++) hist[(key[src[i] * kw + w] >> sh) & 0xFFFF]++;
            if (hist[(key[src[0] * kw + w] >> sh) & 0xFFFF] == n) continue;
            int64_t s = 0;
            for (int d = 0; d < 65536; d++) { int64_t c = hist[d]; hist[d] = s; s += c; }
            for (int64_t i = 0; i < n; i++) dst[hist[(key[src[i] * kw + w] >> sh) & 0xFFFF]++] = src[i];
            int64_t *t = src; src = dst; dst = t;
        }
    }
    if (src != perm) memcpy(perm, src, (size_t)n * sizeof(int64_t));
    free(tmp); free(hist);
}

void orc_sort_perm(const uint64_t *key, int64_t n, int64_t *perm) { orc_sort_perm_w(key, n, 1, perm); }

void orc_partition_starts_w(const uint64_t *sorted_key, int64_t n, int kw, int P, int64_t *start) {
    int64_t prev = 0;
    for (int p = 0; p < P; p++) {
        /* floor(p*n/P) without overflow for n < 2^62/P */
        int64_t s = (int64_t)(((__int128)p * (__int128)n) / P);
        if (s < prev) s = prev;
        while (s > 0 && s < n && key_eq(sorted_key + s * kw, sorted_key + (s - 1) * kw, kw)) s++;
        start[p] = s; prev = s;
    }
    start[P] = n;
}

void orc_partition_starts(const uint64_t *sorted_key, int64_t n, int P, int64_t *start) {
    orc_partition_starts_w(sorted_key, n, 1, P, start);
}

/* --------------------------------------------------------- a-7 forward filter */

/* Generic over the key width kw (words per (k-1)-mer key, AoS).  k <= 31: P/ReflexivMain.java /
 * P/ReflexivDSMain.java (twin selects the arithmetic); k > 31: P/ReflexivDSMain64.java, whose
 * filters are the DS twin's statement for statement with subKmerSlotComparator (:10124-10132)
 * in place of the key comparison. */
#define KEY(a, i) ((a) + (size_t)(i) * (size_t)kw)
#define KEYCPY(d, s_) memcpy((d), (s_), (size_t)kw * 8)

typedef int64_t (*fork_fn)(const uint64_t *, const int32_t *, const uint64_t *, const int32_t *, const int32_t *, int64_t,
                           const int64_t *, int, int, int, int, uint64_t *, int32_t *, uint64_t *, int32_t *, int32_t *,
                           int64_t *);

/* one task per logical partition: the serial filter applied to that partition alone, results concatenated */
static int64_t fork_by_tasks(fork_fn fn, const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                             const int32_t *left, const int32_t *right,
                             const int64_t *part_start, int P, int k, int min_error_cov, int twin,
                             uint64_t *okey, int32_t *omarker, uint64_t *oext,
                             int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    const int kw = orc_sub_words(k);
    int64_t *cnt = (int64_t *)xmalloc((size_t)(P + 1) * 8);
    /* a partition's survivors are no more than its records: write them at the partition's own start */
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
#endif
    for (int p = 0; p < P; p++) {
        const int64_t b = part_start[p], np_ = part_start[p + 1] - b;
        int64_t one[2] = { 0, np_ }, oone[2];
        cnt[p] = fn(key + (size_t)b * kw, marker + b, ext + b, left + b, right + b, np_, one, 1, k, min_error_cov, twin,
                    okey + (size_t)b * kw, omarker + b, oext + b, oleft + b, oright + b, oone);
    }
    int64_t m = 0;
    for (int p = 0; p < P; p++) {                          /* compact towards the front, in order */
        const int64_t b = part_start[p], c = cnt[p];
        out_part_start[p] = m;
        if (b != m && c > 0) {
            memmove(okey + (size_t)m * kw, okey + (size_t)b * kw, (size_t)c * 8 * kw);
            memmove(omarker + m, omarker + b, (size_t)c * 4); memmove(oext + m, oext + b, (size_t)c * 8);
            memmove(oleft + m, oleft + b, (size_t)c * 4); memmove(oright + m, oright + b, (size_t)c * 4);
        }
        m += c;
    }
    out_part_start[P] = m;
    free(cnt);
    return m;
}

static int64_t fork_filter_forward_serial(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                  const int32_t *left, const int32_t *right, int64_t n,
                                  const int64_t *part_start, int P,
                                  int k, int min_error_cov, int twin,
                                  uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                  int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    (void)right; (void)n;
    const int kw = orc_sub_words(k);
    const int32_t sub = k - 1;                              /* param.subKmerSize */
    int64_t m = 0;
    for (int p = 0; p < P; p++) {
        out_part_start[p] = m;
        int64_t first = m;                                   /* list empty per task */
        for (int64_t i = part_start[p]; i < part_start[p + 1]; i++) {
            /* free-end value: RDD -1 (:2478), DS -1-coverage (DS :3436, 64 :10149) */
#define FREE_OF(cov) ((twin == ORC_TWIN_DS && min_error_cov != 0) ? (-1 - (cov)) : -1)
            if (m == first || !key_eq(KEY(key, i), KEY(okey, m - 1), kw)) {      /* :2475,:2529 */
                KEYCPY(KEY(okey, m), KEY(key, i)); omarker[m] = marker[i]; oext[m] = ext[i];
                oleft[m] = left[i]; oright[m] = FREE_OF(left[i]); m++;
                continue;
            }
            int64_t h = m - 1;                               /* the run's survivor */
            int32_t cs = left[i], ch = oleft[h];             /* coverages (_3) */
            if (cs > ch) {                                                   /* :2483 */
                int err = (min_error_cov != 0) && ch <= min_error_cov && cs >= 2 * ch;  /* :2484 */
                omarker[h] = marker[i]; oext[h] = ext[i]; oleft[h] = cs;
                oright[h] = err ? FREE_OF(cs) : sub;
            } else if (cs == ch) {                                           /* :2497 */
                if ((int64_t)ext[i] > (int64_t)oext[h]) {                    /* :2498 */
                    omarker[h] = marker[i]; oext[h] = ext[i]; oleft[h] = cs;
                }
                oright[h] = sub;
            } else {                                                         /* :2512 */
                int err = (min_error_cov != 0) && cs <= min_error_cov && ch >= 2 * cs;  /* :2513 */
                oright[h] = err ? FREE_OF(ch) : sub;
            }
#undef FREE_OF
        }
    }
    out_part_start[P] = m;
    return m;
}

int64_t orc_fork_filter_forward_w(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                  const int32_t *left, const int32_t *right, int64_t n,
                                  const int64_t *part_start, int P,
                                  int k, int min_error_cov, int twin,
                                  uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                  int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    if (g_threads > 1 && P > 1)
        return fork_by_tasks(fork_filter_forward_serial, key, marker, ext, left, right, part_start, P, k, min_error_cov, twin,
                             okey, omarker, oext, oleft, oright, out_part_start);
    return fork_filter_forward_serial(key, marker, ext, left, right, n, part_start, P, k, min_error_cov, twin,
                                      okey, omarker, oext, oleft, oright, out_part_start);
}

int64_t orc_fork_filter_forward(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                const int32_t *left, const int32_t *right, int64_t n,
                                const int64_t *part_start, int P,
                                int k, int min_error_cov, int twin,
                                uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    return orc_fork_filter_forward_w(key, marker, ext, left, right, n, part_start, P, k, min_error_cov, twin,
                                     okey, omarker, oext, oleft, oright, out_part_start);
}

/* --------------------------------------------- sequence-level record helpers */

static int64_t ext_len_words(const uint64_t *w, int64_t nw) {
    /* (length-1)*31 + firstBlockLength  P/ReflexivMain.java:820-823 */
    return (nw - 1) * 31 + sentinel_len(w[0]);
}

/* key: words 0..kw-2 hold 31 bases each, the last word the remaining sub - 31*(kw-1) bases
 * (subKmerSizeResidue), right-aligned  (P/ReflexivDSMain64.java:1928-1940; kw = 1: the
 * single long of the k <= 31 classes) */
static void unpack_key(const uint64_t *key, int kw, int sub, uint8_t *b) {
    int o = 0;
    for (int w = 0; w < kw; w++) {
        const int nb = w < kw - 1 ? 31 : sub - 31 * (kw - 1);
        for (int j = 0; j < nb; j++) b[o++] = (uint8_t)((key[w] >> (2 * (nb - 1 - j))) & 3);
    }
}
static void pack_key(const uint8_t *b, int kw, int sub, uint64_t *key) {
    int o = 0;
    for (int w = 0; w < kw; w++) {
        const int nb = w < kw - 1 ? 31 : sub - 31 * (kw - 1);
        uint64_t x = 0;
        for (int j = 0; j < nb; j++) x = (x << 2) | b[o++];
        key[w] = x;
    }
}
static void pack_key_fwd(const uint8_t *b, int kw, int sub, uint64_t *key) { pack_key(b, kw, sub, key); }
static void unpack_ext(const uint64_t *w, int64_t nw, uint8_t *b) {
    int f = sentinel_len(w[0]);
    int64_t o = 0;
    for (int j = 0; j < f; j++) b[o++] = (uint8_t)((w[0] >> (2 * (f - 1 - j))) & 3);
    for (int64_t i = 1; i < nw; i++)
        for (int j = 0; j < 31; j++) b[o++] = (uint8_t)((w[i] >> (2 * (30 - j))) & 3);
}
static int64_t ext_words_for(int64_t len) { return (len + 30) / 31; }
static void pack_ext(const uint8_t *b, int64_t len, uint64_t *w) {
    int64_t nw = ext_words_for(len);
    int f = (int)(len - 31 * (nw - 1));
    uint64_t x = 1;                                   /* the "C marker" sentinel */
    int64_t o = 0;
    for (int j = 0; j < f; j++) x = (x << 2) | b[o++];
    w[0] = x;
    for (int64_t i = 1; i < nw; i++) {
        x = 0;
        for (int j = 0; j < 31; j++) x = (x << 2) | b[o++];
        w[i] = x;
    }
}

/* full sequence of a record: marker 1 = key||ext, marker 2 = ext||key */
static int64_t record_seq(const uint64_t *key, int kw, int marker, const uint64_t *w, int64_t nw, int sub,
                          uint8_t *b) {
    int64_t L = ext_len_words(w, nw);
    if (marker == 1) { unpack_key(key, kw, sub, b); unpack_ext(w, nw, b + sub); }
    else             { unpack_ext(w, nw, b); unpack_key(key, kw, sub, b + L); }
    return L + sub;
}

/* store sequence b[0..len) in orientation m at output slot */
typedef struct {
    uint64_t *key; int32_t *marker; int64_t *ext_off; uint64_t *ext;
    int32_t *left; int32_t *right; int64_t n; int kw;
} out_set;

static void emit_seq(out_set *o, const uint8_t *b, int64_t len, int sub, int m,
                     int32_t left, int32_t right) {
    int64_t i = o->n++;
    int64_t L = len - sub;
    const int kw = o->kw;
    uint64_t *w = o->ext + o->ext_off[i];
    if (m == 1) { pack_key(b, kw, sub, KEY(o->key, i));       pack_ext(b + sub, L, w); }
    else        { pack_key(b + L, kw, sub, KEY(o->key, i));   pack_ext(b, L, w); }
    o->marker[i] = m; o->left[i] = left; o->right[i] = right;
    o->ext_off[i + 1] = o->ext_off[i] + ext_words_for(L);
}

/* ------------------------------------------------------- a-8 reflect records */

void orc_reflect_from_forward_w(const uint64_t *key, const uint64_t *ext, int64_t n, int k,
                                uint64_t *okey, int32_t *omarker, uint64_t *oext) {
    /* ReflectedSubKmerExtractionFromForward  P/ReflexivMain.java:2742-2768; 64: :10426-10475 (the
     * first base leaves word 0, every word shifts left by one base taking the top base of its
     * right neighbour, the suffix base enters the last word): key' = key[1..] || suffix. */
    const int kw = orc_sub_words(k), sub = k - 1;
    uint8_t b[4 * 31 * 4 + 8];
    uint8_t *bb = kw <= 16 ? b : (uint8_t *)xmalloc((size_t)sub + 8);
    for (int64_t i = 0; i < n; i++) {
        unpack_key(KEY(key, i), kw, sub, bb);
        const uint64_t first = bb[0];                                        /* :2752-2754 / :10448 */
        bb[sub] = (uint8_t)(ext[i] & 3);                                     /* :2757-2758 / :10455 */
        pack_key(bb + 1, kw, sub, KEY(okey, i));
        omarker[i] = 2; oext[i] = first | 4;                                 /* :2755,:2762 / :10450 */
    }
    if (bb != b) free(bb);
}

void orc_reflect_from_forward(const uint64_t *key, const uint64_t *ext, int64_t n, int k,
                              uint64_t *okey, int32_t *omarker, uint64_t *oext) {
    orc_reflect_from_forward_w(key, ext, n, k, okey, omarker, oext);
}

/* ------------------------------------------------------- a-9 reflected filter */

static int64_t fork_filter_reflected_serial(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                    const int32_t *left, const int32_t *right, int64_t n,
                                    const int64_t *part_start, int P,
                                    int k, int min_error_cov, int twin,
                                    uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                    int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    (void)n;
    const int kw = orc_sub_words(k);
    const int32_t sub = k - 1;
    const int ds_ec = (twin == ORC_TWIN_DS && min_error_cov != 0);
    int64_t m = 0;
    for (int p = 0; p < P; p++) {
        out_part_start[p] = m;
        int64_t first = m;
        int32_t last_cov = 0;                                /* HighCoverLastCoverage :2614 */
        for (int64_t i = part_start[p]; i < part_start[p + 1]; i++) {
            int32_t cs = left[i];
            if (m == first || !key_eq(KEY(key, i), KEY(okey, m - 1), kw)) {      /* :2622,:2684 */
                last_cov = cs;
                KEYCPY(KEY(okey, m), KEY(key, i)); omarker[m] = marker[i]; oext[m] = ext[i];
                oleft[m] = ds_ec ? (-1 - cs) : -1;                            /* :2626 / DS :3556 / 64 :10292 */
                oright[m] = right[i]; m++;
                continue;
            }
            int64_t h = m - 1;
            if (cs > last_cov) {                                             /* :2631 */
                int err = (min_error_cov != 0) && last_cov <= min_error_cov && cs >= 2 * last_cov;
                last_cov = cs;
                omarker[h] = marker[i]; oext[h] = ext[i]; oright[h] = right[i];
                oleft[h] = err ? (ds_ec ? (-1 - cs) : -1) : sub;              /* :2636,:2643 */
            } else if (cs == last_cov) {                                     /* :2647 */
                /* compares (ext >>> 2*(len-1)) of the new record with (ext >>> 2*len)
                 * of the survivor, lengths from the sentinel  :2648-2653 */
                int ls = sentinel_len(ext[i]);
                int lh = sentinel_len(oext[h]);
                /* Java shift counts are taken mod 64 */
                uint64_t a = ext[i] >> ((2 * (ls - 1)) & 63);
                uint64_t b = oext[h] >> ((2 * lh) & 63);
                if ((int64_t)a > (int64_t)b) {
                    omarker[h] = marker[i]; oext[h] = ext[i]; oright[h] = right[i];
                }
                oleft[h] = sub;                                              /* :2656,:2663 */
            } else {                                                         /* :2667 */
                int err = (min_error_cov != 0) && cs <= min_error_cov && last_cov >= 2 * cs;
                if (err) {
                    /* RDD :2672 stores -1; DS :3595 / 64 :10330 keep the previous left */
                    if (!ds_ec) oleft[h] = -1;
                } else {
                    oleft[h] = sub;                                          /* :2679 */
                }
            }
        }
    }
    out_part_start[P] = m;
    return m;
}

int64_t orc_fork_filter_reflected_w(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                    const int32_t *left, const int32_t *right, int64_t n,
                                    const int64_t *part_start, int P,
                                    int k, int min_error_cov, int twin,
                                    uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                    int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    if (g_threads > 1 && P > 1)
        return fork_by_tasks(fork_filter_reflected_serial, key, marker, ext, left, right, part_start, P, k, min_error_cov, twin,
                             okey, omarker, oext, oleft, oright, out_part_start);
    return fork_filter_reflected_serial(key, marker, ext, left, right, n, part_start, P, k, min_error_cov, twin,
                                        okey, omarker, oext, oleft, oright, out_part_start);
}

int64_t orc_fork_filter_reflected(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                  const int32_t *left, const int32_t *right, int64_t n,
                                  const int64_t *part_start, int P,
                                  int k, int min_error_cov, int twin,
                                  uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                  int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    return orc_fork_filter_reflected_w(key, marker, ext, left, right, n, part_start, P, k, min_error_cov, twin,
                                       okey, omarker, oext, oleft, oright, out_part_start);
}

/* --------------------------------------------------- a-10 random reflection */

void orc_random_reflection_w(uint64_t *key, int32_t *marker, uint64_t *ext, int64_t n,
                             const int64_t *part_start, int P, int k) {
    /* kmerRandomReflection  P/ReflexivMain.java:2783-2885; 64: DSkmerRandomReflection :10491-10690 */
    (void)n;
    const int kw = orc_sub_words(k), sub = k - 1;
    if (g_threads > 1 && P > 1) {                    /* one task per logical partition */
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
#endif
        for (int p = 0; p < P; p++) {
            const int64_t b0 = part_start[p];
            int64_t one[2] = { 0, part_start[p + 1] - b0 };
            /* P = 1 below: the serial branch */
            orc_random_reflection_w(key + (size_t)b0 * kw, marker + b0, ext + b0, one[1], one, 1, k);
        }
        return;
    }
    uint8_t *b = (uint8_t *)xmalloc((size_t)sub + 96);
    for (int p = 0; p < P; p++) {
        int m = 2;                                   /* randomReflexivMarker = 2 :2777 */
        for (int64_t i = part_start[p]; i < part_start[p + 1]; i++) {
            if (marker[i] != m) {                    /* singleKmerRandomizer :2792-2876 */
                int64_t len = record_seq(KEY(key, i), kw, marker[i], &ext[i], 1, sub, b);
                int64_t L = len - sub;
                if (m == 1) { pack_key(b, kw, sub, KEY(key, i));     pack_ext(b + sub, L, &ext[i]); }
                else        { pack_key(b + L, kw, sub, KEY(key, i)); pack_ext(b, L, &ext[i]); }
                marker[i] = m;
            }
            m = 3 - m;                                                      /* :2880-2884 */
        }
    }
    free(b);
}

void orc_random_reflection(uint64_t *key, int32_t *marker, uint64_t *ext, int64_t n,
                           const int64_t *part_start, int P, int k) {
    orc_random_reflection_w(key, marker, ext, n, part_start, P, k);
}

/* ------------------------------------------------------ a-11..13 extend pass */

#define ORC_BLOCK (INT32_MIN)

static int64_t extend_pass_serial(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                          const uint64_t *ext, const int32_t *left, const int32_t *right,
                          int64_t n, const int64_t *part_start, int P, int k, int twin, int start_marker,
                          uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                          int32_t *oleft, int32_t *oright, int64_t *out_part_start);

/* one task per logical partition (P/ReflexivMain.java: one call() per partition): the serial pass applied
 * to that partition alone into private buffers, then concatenated in partition order */
static int64_t extend_by_tasks(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                               const uint64_t *ext, const int32_t *left, const int32_t *right,
                               const int64_t *part_start, int P, int k, int twin, int start_marker,
                               uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                               int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    const int kw = orc_sub_words(k);
    orc_records *tmp = (orc_records *)xmalloc((size_t)P * sizeof(orc_records));
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
#endif
    for (int p = 0; p < P; p++) {
        const int64_t b = part_start[p], np_ = part_start[p + 1] - b;
        const int64_t wp = ext_off[b + np_] - ext_off[b];
        orc_records *t = &tmp[p];
        t->key = (uint64_t *)xmalloc((size_t)(np_ ? np_ : 1) * 8 * kw);
        t->marker = (int32_t *)xmalloc((size_t)(np_ ? np_ : 1) * 4);
        t->ext_off = (int64_t *)xmalloc((size_t)(np_ + 1) * 8);
        t->ext = (uint64_t *)xmalloc((size_t)(wp ? wp : 1) * 8);
        t->left = (int32_t *)xmalloc((size_t)(np_ ? np_ : 1) * 4);
        t->right = (int32_t *)xmalloc((size_t)(np_ ? np_ : 1) * 4);
        int64_t one[2] = { 0, np_ }, oone[2];
        /* ext_off + b keeps absolute word offsets into ext */
        t->n = extend_pass_serial(key + (size_t)b * kw, marker + b, ext_off + b, ext, left + b, right + b, np_, one, 1, k,
                                  twin, start_marker, t->key, t->marker, t->ext_off, t->ext, t->left, t->right, oone);
    }
    int64_t m = 0, w = 0;
    oext_off[0] = 0;
    for (int p = 0; p < P; p++) {
        orc_records *t = &tmp[p];
        out_part_start[p] = m;
        const int64_t c = t->n, cw = t->ext_off[c];
        memcpy(okey + (size_t)m * kw, t->key, (size_t)c * 8 * kw);
        memcpy(omarker + m, t->marker, (size_t)c * 4); memcpy(oleft + m, t->left, (size_t)c * 4);
        memcpy(oright + m, t->right, (size_t)c * 4); memcpy(oext + w, t->ext, (size_t)cw * 8);
        for (int64_t i = 0; i < c; i++) oext_off[m + i + 1] = w + t->ext_off[i + 1];
        m += c; w += cw;
        free(t->key); free(t->marker); free(t->ext_off); free(t->ext); free(t->left); free(t->right);
    }
    out_part_start[P] = m;
    free(tmp);
    return m;
}

int64_t orc_extend_pass_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                          const uint64_t *ext, const int32_t *left, const int32_t *right,
                          int64_t n, const int64_t *part_start, int P, int k, int twin, int start_marker,
                          uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                          int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    if (g_threads > 1 && P > 1)
        return extend_by_tasks(key, marker, ext_off, ext, left, right, part_start, P, k, twin, start_marker,
                               okey, omarker, oext_off, oext, oleft, oright, out_part_start);
    return extend_pass_serial(key, marker, ext_off, ext, left, right, n, part_start, P, k, twin, start_marker,
                              okey, omarker, oext_off, oext, oleft, oright, out_part_start);
}

static int64_t extend_pass_serial(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                          const uint64_t *ext, const int32_t *left, const int32_t *right,
                          int64_t n, const int64_t *part_start, int P, int k, int twin, int start_marker,
                          uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                          int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    const int kw = orc_sub_words(k), sub = k - 1;
    int64_t maxw = 1;
    for (int64_t i = 0; i < n; i++) { int64_t w = ext_off[i + 1] - ext_off[i]; if (w > maxw) maxw = w; }
    /* scratch big enough for a merged sequence */
    uint8_t *bs = (uint8_t *)xmalloc((size_t)(2 * maxw * 31 + 2 * sub + 64));
    out_set o = { okey, omarker, oext_off, oext, oleft, oright, 0, kw };
    oext_off[0] = 0;

#define EXTW(i)   (ext + ext_off[i])
#define EXTN(i)   (ext_off[(i) + 1] - ext_off[i])
#define FLIP_EMIT(i) do { /* singleKmerRandomizer  P/ReflexivMain.java:910-1063 (64 :7605-8170) */ \
        int64_t len_ = record_seq(KEY(key, i), kw, marker[i], EXTW(i), EXTN(i), sub, bs); \
        emit_seq(&o, bs, len_, sub, m, left[i], right[i]); m = 3 - m; } while (0)

    for (int p = 0; p < P; p++) {
        out_part_start[p] = o.n;
        /* randomReflexivMarker = 2 :770; the k > 31 array loop starts at 1 once param.scramble == 3
         * (P/ReflexivDSMain64.java:7484-7486) */
        int m = start_marker;
        int64_t holder = -1;                         /* tmpReflexivKmerExtendList (<= 1 element) */
        for (int64_t s = part_start[p]; s < part_start[p + 1]; s++) {
            if (holder < 0) { holder = s; continue; }                        /* :797-799,:813-814 */
            if (!key_eq(KEY(key, s), KEY(key, holder), kw)) {                /* :886-893 */
                FLIP_EMIT(holder); holder = s; continue;
            }
            if (marker[s] == marker[holder]) {                               /* :845-854 */
                FLIP_EMIT(s); continue;
            }
            int64_t F = (marker[s] == 1) ? s : holder;       /* forward record   */
            int64_t R = (marker[s] == 1) ? holder : s;       /* reflected record */
            int32_t a = left[F], b = right[R];               /* junction-side markers */
            int64_t lenF = ext_len_words(EXTW(F), EXTN(F));
            int64_t lenR = ext_len_words(EXTW(R), EXTN(R));
            int64_t d;
            if ((a < 0 && b < 0) || (a >= 0 && b >= 0)) {                     /* :825-832 */
                d = -1;
            } else if (s == F) {                                             /* :833-840 */
                if (a >= 0 && a - lenR >= 0) d = a - lenR;
                else if (b >= 0 && b - lenF >= 0) d = b - lenF;
                else d = ORC_BLOCK;
            } else {                                                         /* :868-875 */
                if (b >= 0 && b - lenF >= 0) d = b - lenF;
                else if (twin == ORC_TWIN_RDD) {
                    /* RDD twin tests left but subtracts from right  :872-873 */
                    int32_t hr = right[F];
                    if (a >= 0 && hr - lenR >= 0) d = hr - lenR; else d = ORC_BLOCK;
                } else {
                    if (a >= 0 && a - lenR >= 0) d = a - lenR; else d = ORC_BLOCK; /* DS :1856-1857, 64 :7567 */
                }
            }
            if (d == ORC_BLOCK) { FLIP_EMIT(s); continue; }                   /* :841-843 */
            /* reflexivExtend: R.ext || key || F.ext   P/ReflexivMain.java:1077-1519 (64 :8199-8679) */
            int64_t lr = record_seq(KEY(key, R), kw, 2, EXTW(R), EXTN(R), sub, bs);   /* R.ext||key */
            unpack_ext(EXTW(F), EXTN(F), bs + lr);
            int64_t len = lr + lenF;
            int32_t L, Rt;
            if (d < 0)             { L = left[R];   Rt = right[F]; }          /* :1214-1218 / 64 :8657-8660 */
            else if (left[F] > 0)  { L = (int32_t)d; Rt = right[F]; }         /* :1220-1226 / 64 :8662-8665 */
            else                   { L = left[R];   Rt = (int32_t)d; }        /* :1227-1233 / 64 :8667-8670 */
            emit_seq(&o, bs, len, sub, m, L, Rt); m = 3 - m;                 /* :1242,:1514 */
            holder = -1;
        }
        if (holder >= 0) FLIP_EMIT(holder);                                   /* :902 */
    }
    out_part_start[P] = o.n;
    free(bs);
    return o.n;
#undef EXTW
#undef EXTN
#undef FLIP_EMIT
}

int64_t orc_extend_pass(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                        const uint64_t *ext, const int32_t *left, const int32_t *right,
                        int64_t n, const int64_t *part_start, int P, int k, int twin,
                        uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                        int32_t *oleft, int32_t *oright, int64_t *out_part_start) {
    return orc_extend_pass_w(key, marker, ext_off, ext, left, right, n, part_start, P, k, twin, 2,
                             okey, omarker, oext_off, oext, oleft, oright, out_part_start);
}

void orc_gather_w(const int64_t *perm, int64_t n, int kw,
                  const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                  const uint64_t *ext, const int32_t *left, const int32_t *right,
                  uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                  int32_t *oleft, int32_t *oright) {
    oext_off[0] = 0;
    for (int64_t i = 0; i < n; i++) { int64_t s = perm[i]; oext_off[i + 1] = oext_off[i] + (ext_off[s + 1] - ext_off[s]); }
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1 && n >= (1 << 16))
#endif
    for (int64_t i = 0; i < n; i++) {
        int64_t s = perm[i];
        KEYCPY(KEY(okey, i), KEY(key, s)); omarker[i] = marker[s]; oleft[i] = left[s]; oright[i] = right[s];
        int64_t nw = ext_off[s + 1] - ext_off[s];
        memcpy(oext + oext_off[i], ext + ext_off[s], (size_t)nw * 8);
    }
}

void orc_gather(const int64_t *perm, int64_t n,
                const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                const uint64_t *ext, const int32_t *left, const int32_t *right,
                uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                int32_t *oleft, int32_t *oright) {
    orc_gather_w(perm, n, 1, key, marker, ext_off, ext, left, right, okey, omarker, oext_off, oext, oleft, oright);
}


/* ------------------------------------------- k > 31 from-counts extras */

int64_t orc_double_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                     const int32_t *left, const int32_t *right, int64_t n, int k,
                     uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft, int32_t *oright) {
    /* DSReflexivAndForwardKmer.call :2153-2168: add s, then singleKmerRandomizer(s) with the marker s does not have */
    const int kw = orc_sub_words(k), sub = k - 1;
    int64_t maxw = 1;
    for (int64_t i = 0; i < n; i++) { int64_t w = ext_off[i + 1] - ext_off[i]; if (w > maxw) maxw = w; }
    uint8_t *bs = (uint8_t *)xmalloc((size_t)(maxw * 31 + sub + 64));
    out_set o = { okey, omarker, oext_off, oext, oleft, oright, 0, kw };
    oext_off[0] = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t nw = ext_off[i + 1] - ext_off[i];
        int64_t len = record_seq(KEY(key, i), kw, marker[i], ext + ext_off[i], nw, sub, bs);
        emit_seq(&o, bs, len, sub, marker[i], left[i], right[i]);                 /* :2157 */
        emit_seq(&o, bs, len, sub, marker[i] == 1 ? 2 : 1, left[i], right[i]);    /* :2158-2163 */
    }
    free(bs);
    return o.n;
}

int64_t orc_flip_all_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                       const int32_t *left, const int32_t *right, int64_t n, int k, int m,
                       uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft, int32_t *oright) {
    /* DSFilterUnExtendableKmerLeftEnds.call :3424-3444 (randomReflexivMarker constantly 1) /
     * DSFilterUnExtendableKmerRightEnds.call :4381-4401 (constantly 2) */
    const int kw = orc_sub_words(k), sub = k - 1;
    int64_t maxw = 1;
    for (int64_t i = 0; i < n; i++) { int64_t w = ext_off[i + 1] - ext_off[i]; if (w > maxw) maxw = w; }
    uint8_t *bs = (uint8_t *)xmalloc((size_t)(maxw * 31 + sub + 64));
    out_set o = { okey, omarker, oext_off, oext, oleft, oright, 0, kw };
    oext_off[0] = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t len = record_seq(KEY(key, i), kw, marker[i], ext + ext_off[i], ext_off[i + 1] - ext_off[i], sub, bs);
        emit_seq(&o, bs, len, sub, m, left[i], right[i]);
    }
    free(bs);
    return o.n;
}

int64_t orc_key_filter_w(int op, const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                         const int32_t *left, const int32_t *right, int64_t n, const int64_t *part_start, int P, int k,
                         uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft,
                         int32_t *oright, int64_t *out_part_start) {
    (void)n;
    const int kw = orc_sub_words(k), sub = k - 1;
    int64_t maxw = 1;
    for (int64_t i = 0; i < n; i++) { int64_t w = ext_off[i + 1] - ext_off[i]; if (w > maxw) maxw = w; }
    uint8_t *bs = (uint8_t *)xmalloc((size_t)(maxw * 31 + sub + 64));
    out_set o = { okey, omarker, oext_off, oext, oleft, oright, 0, kw };
    oext_off[0] = 0;
#define EXTW(i)   (ext + ext_off[i])
#define EXTN(i)   (ext_off[(i) + 1] - ext_off[i])
#define ADD(i, M) do { int64_t len_ = record_seq(KEY(key, i), kw, marker[i], EXTW(i), EXTN(i), sub, bs); \
                       emit_seq(&o, bs, len_, sub, (M), left[i], right[i]); } while (0)
#define ADD_ASIS(i) ADD(i, marker[i])
    for (int p = 0; p < P; p++) {
        out_part_start[p] = o.n;
        int64_t holder = -1;                         /* tmpReflexivKmerExtendList (<= 1 element) */
        for (int64_t s = part_start[p]; s < part_start[p + 1]; s++) {
            if (holder < 0) { holder = s; continue; }                         /* lineMarker == 1 / size() == 0 */
            const int64_t h = holder;
            if (!key_eq(KEY(key, s), KEY(key, h), kw)) {                     /* new sub-kmer group */
                if (op != ORC_OP_EXTENDABLE_PAIRS) ADD_ASIS(h);              /* :6518 / :3328 / :3160; the pairs filter drops it (:5455) */
                holder = s; continue;
            }
            if (op == ORC_OP_FIRST_OF_KEY) { ADD_ASIS(h); holder = -1; continue; }          /* :3289-3312 */
            const int64_t lenH = ext_len_words(EXTW(h), EXTN(h)), lenS = ext_len_words(EXTW(s), EXTN(s));
            if (op == ORC_OP_LONGER_OF_KEY) {                                /* :3103-3145 */
                const int64_t fh = sentinel_len(EXTW(h)[0]), fs = sentinel_len(EXTW(s)[0]);
                if (lenH * 31 + fh >= lenS * 31 + fs) ADD_ASIS(h); else ADD_ASIS(s);
                holder = -1; continue;
            }
            /* the merge test of the extend pass on (forward, reflected) of one key:  :5376-5399 / :6448-6467 */
            int mergeable = 0, opposite = marker[s] != marker[h];
            if (opposite) {
                const int32_t a = marker[s] == 1 ? left[s] : right[s];       /* current's junction side  */
                const int32_t b = marker[s] == 1 ? right[h] : left[h];       /* holder's junction side   */
                mergeable = (a < 0 && b < 0) || (a >= 0 && b >= 0) || (a >= 0 && a - lenH >= 0) || (b >= 0 && b - lenS >= 0);
            }
            if (op == ORC_OP_EXTENDABLE_PAIRS) {
                if (mergeable) {
                    if (marker[s] == 1) { ADD_ASIS(s); ADD(h, 1); }          /* :5377-5379 */
                    else                { ADD_ASIS(h); ADD(s, 1); }          /* :5416-5418 */
                    holder = -1;
                } else holder = s;                                           /* resetSubKmerGroup(s) :5397, :5402, :5408, :5436 */
            } else {                                                         /* ORC_OP_UNEXTENDABLE */
                if (mergeable) { holder = -1; continue; }                    /* "already extended": both dropped */
                if (marker[h] == 2) ADD(h, 1); else ADD_ASIS(h);             /* :6469, :6482 (singleKmerRandomizer, marker 1) / :6475, :6503 */
                holder = s;
            }
        }
        if (holder >= 0) ADD_ASIS(holder);                                   /* :5462-5466 / :6532-6536 / :3335 / :3167 */
    }
    out_part_start[P] = o.n;
    free(bs);
    return o.n;
#undef EXTW
#undef EXTN
#undef ADD
#undef ADD_ASIS
}

/* --------------------------------------------------------------- a-15 contigs */

/* header: 0 = RDD twin ">Contig-<len>-<idx>" (also P/ReflexivDSMain64.java:830-866),
 *         1 = DS twin  ">Contig-<len>-(<left>,<right>)-<idx>", which also skips records whose
 *             markers are both <= -10,000,000 (P/ReflexivDSMain.java:749) */
static int64_t contigs_text_impl(const uint64_t *key, int kw, const int32_t *marker, const int64_t *ext_off,
                                 const uint64_t *ext, const int32_t *left, const int32_t *right,
                                 int64_t n, int k, int min_contig, int ds_header,
                                 char *out, int64_t cap, int64_t *n_contigs) {
    static const char NUC[4] = { 'A', 'C', 'G', 'T' };
    const int sub = k - 1;
    int64_t pos = 0, idx = 0;
    uint8_t *b = NULL; int64_t bcap = 0;
#define PUTC(c) do { if (pos < cap) out[pos] = (c); pos++; } while (0)
    for (int64_t i = 0; i < n; i++) {
        if (ds_header && left[i] <= -10000000 && right[i] <= -10000000) continue;
        int64_t nw = ext_off[i + 1] - ext_off[i];
        int64_t len = ext_len_words(ext + ext_off[i], nw) + sub;
        if (len < min_contig) continue;                                      /* :596,:606 */
        if (len + 64 > bcap) { free(b); bcap = 2 * len + 64; b = (uint8_t *)xmalloc((size_t)bcap); }
        record_seq(KEY(key, i), kw, marker[i], ext + ext_off[i], nw, sub, b);
        char hdr[96];
        int hl;
        if (ds_header)                                                       /* DS :755,:722 */
            hl = snprintf(hdr, sizeof hdr, ">Contig-%lld-(%d,%d)-%lld\n", (long long)len,
                          left[i], right[i], (long long)idx);
        else                                                                 /* :597,:578 */
            hl = snprintf(hdr, sizeof hdr, ">Contig-%lld-%lld\n", (long long)len, (long long)idx);
        for (int j = 0; j < hl; j++) PUTC(hdr[j]);
        for (int64_t j = 0; j < len; j++) {                                  /* changeLine :616-637 */
            if (j > 0 && j % 100 == 0) PUTC('\n');
            PUTC(NUC[b[j]]);
        }
        PUTC('\n');                                  /* saveAsTextFile line terminator */
        idx++;
    }
#undef PUTC
    free(b);
    if (n_contigs) *n_contigs = idx;
    return pos;
}

int64_t orc_contigs_text(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                         const uint64_t *ext, const int32_t *left, const int32_t *right,
                         int64_t n, int k, int min_contig, int twin,
                         char *out, int64_t cap, int64_t *n_contigs) {
    return contigs_text_impl(key, 1, marker, ext_off, ext, left, right, n, k, min_contig, twin == ORC_TWIN_DS,
                             out, cap, n_contigs);
}

int64_t orc_contigs_text_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                           const uint64_t *ext, const int32_t *left, const int32_t *right,
                           int64_t n, int k, int min_contig,
                           char *out, int64_t cap, int64_t *n_contigs) {
    return contigs_text_impl(key, orc_sub_words(k), marker, ext_off, ext, left, right, n, k, min_contig, 0,
                             out, cap, n_contigs);
}

/* ---------------------------------------------------------------- a-14 driver */

static void rec_alloc(orc_records *r, int64_t n, int64_t words, int kw) {
    r->n = 0;
    r->key = (uint64_t *)xmalloc((size_t)n * 8 * (size_t)kw);
    r->marker = (int32_t *)xmalloc((size_t)n * 4);
    r->ext_off = (int64_t *)xmalloc((size_t)(n + 1) * 8);
    r->ext = (uint64_t *)xmalloc((size_t)words * 8);
    r->left = (int32_t *)xmalloc((size_t)n * 4);
    r->right = (int32_t *)xmalloc((size_t)n * 4);
    r->ext_off[0] = 0;
}
void orc_free_records(orc_records *r) {
    free(r->key); free(r->marker); free(r->ext_off); free(r->ext); free(r->left); free(r->right);
    memset(r, 0, sizeof *r);
}

/* stable sort by key (sortByKey / sort("k-1")); consumes *cur */
static void sort_records_w(orc_records *cur, int kw) {
    int64_t n = cur->n, words = cur->ext_off[n];
    int64_t *perm = (int64_t *)xmalloc((size_t)(n ? n : 1) * 8);
    orc_sort_perm_w(cur->key, n, kw, perm);
    orc_records srt; rec_alloc(&srt, n ? n : 1, words ? words : 1, kw);
    orc_gather_w(perm, n, kw, cur->key, cur->marker, cur->ext_off, cur->ext, cur->left, cur->right,
                 srt.key, srt.marker, srt.ext_off, srt.ext, srt.left, srt.right);
    srt.n = n;
    free(perm);
    orc_free_records(cur);
    *cur = srt;
}

/* sort + one mapPartitions(extend pass); consumes *cur, returns the new set */
static void sort_and_extend(orc_records *cur, int P, int k, int twin, int start_marker) {
    const int kw = orc_sub_words(k);
    sort_records_w(cur, kw);                                  /* sortByKey  :235,:247,:286 */
    int64_t n = cur->n, words = cur->ext_off[n];
    int64_t *ps = (int64_t *)xmalloc((size_t)(P + 1) * 8), *ops = (int64_t *)xmalloc((size_t)(P + 1) * 8);
    orc_partition_starts_w(cur->key, n, kw, P, ps);
    orc_records out; rec_alloc(&out, n ? n : 1, words ? words : 1, kw);
    out.n = orc_extend_pass_w(cur->key, cur->marker, cur->ext_off, cur->ext, cur->left, cur->right, n,
                              ps, P, k, twin, start_marker,
                              out.key, out.marker, out.ext_off, out.ext, out.left, out.right, ops);
    free(ps); free(ops);
    orc_free_records(cur);
    *cur = out;
}

/* RC expand ... random reflection (the operators before the extend loop), shared by both drivers;
 * kmers: n k-mers of (k-1)/31+1 words each in the assembler layout (one word for k <= 31) */
static void records_before_loop(const uint64_t *kmers, const int32_t *counts, int64_t n, const orc_params *prm,
                                int P, orc_records *cur) {
    const int k = prm->k, twin = prm->twin, kw = orc_sub_words(k);
    int64_t n2 = 2 * n;
    int64_t a = n2 ? n2 : 1;
    /* RC expand + forward sub-kmers  :168-176 */
    uint64_t *key = (uint64_t *)xmalloc((size_t)a * 8 * kw), *ext = (uint64_t *)xmalloc((size_t)a * 8);
    int32_t *marker = (int32_t *)xmalloc((size_t)a * 4), *left = (int32_t *)xmalloc((size_t)a * 4),
            *right = (int32_t *)xmalloc((size_t)a * 4);
    uint64_t *key2 = (uint64_t *)xmalloc((size_t)a * 8 * kw), *ext2 = (uint64_t *)xmalloc((size_t)a * 8);
    int32_t *marker2 = (int32_t *)xmalloc((size_t)a * 4), *left2 = (int32_t *)xmalloc((size_t)a * 4),
            *right2 = (int32_t *)xmalloc((size_t)a * 4);
    int64_t *perm = (int64_t *)xmalloc((size_t)a * 8);
    int64_t *ps = (int64_t *)xmalloc((size_t)(P + 1) * 8), *ps2 = (int64_t *)xmalloc((size_t)(P + 1) * 8);
    orc_rc_expand_subkmer_w(kmers, counts, n, k, key, marker, ext, left, right);
    int64_t m = n2;

#define SORT_FIXED() do { \
        orc_sort_perm_w(key, m, kw, perm); \
        for (int64_t i_ = 0; i_ < m; i_++) { int64_t s_ = perm[i_]; KEYCPY(KEY(key2, i_), KEY(key, s_)); \
            marker2[i_] = marker[s_]; ext2[i_] = ext[s_]; left2[i_] = left[s_]; right2[i_] = right[s_]; } \
        orc_partition_starts_w(key2, m, kw, P, ps); } while (0)

    /* sortByKey + forward fork filter  :179-186 */
    SORT_FIXED();
    m = orc_fork_filter_forward_w(key2, marker2, ext2, left2, right2, m, ps, P, k,
                                  prm->min_error_cov, twin, key, marker, ext, left, right, ps2);
    /* reflected extraction  :188-189 */
    orc_reflect_from_forward_w(key, ext, m, k, key2, marker2, ext2);
    memcpy(key, key2, (size_t)m * 8 * kw); memcpy(ext, ext2, (size_t)m * 8); memcpy(marker, marker2, (size_t)m * 4);
    /* sortByKey + reflected fork filter  :191-198 */
    SORT_FIXED();
    m = orc_fork_filter_reflected_w(key2, marker2, ext2, left2, right2, m, ps, P, k,
                                    prm->min_error_cov, twin, key, marker, ext, left, right, ps2);
    /* random reflection on the filter's output partitions  :204-205 */
    orc_random_reflection_w(key, marker, ext, m, ps2, P, k);
#undef SORT_FIXED

    /* move to the variable-length record set (single word == 1-word array) */
    rec_alloc(cur, m ? m : 1, m ? m : 1, kw);
    for (int64_t i = 0; i < m; i++) {
        KEYCPY(KEY(cur->key, i), KEY(key, i)); cur->marker[i] = marker[i]; cur->left[i] = left[i]; cur->right[i] = right[i];
        cur->ext[i] = ext[i]; cur->ext_off[i + 1] = i + 1;
    }
    cur->n = m;
    free(key); free(ext); free(marker); free(left); free(right);
    free(key2); free(ext2); free(marker2); free(left2); free(right2);
    free(perm); free(ps); free(ps2);
}

int64_t orc_assemble_from_counts(const uint64_t *kmers, const int32_t *counts, int64_t n,
                                 const orc_params *prm,
                                 char *out, int64_t cap, int64_t *n_contigs,
                                 int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                                 orc_records *rec_out) {
    const int k = prm->k, twin = prm->twin;
    int P = prm->partitions > 0 ? prm->partitions : 1;
    int64_t nt = 0;
    orc_records cur;
    records_before_loop(kmers, counts, n, prm, P, &cur);

#define TRACE() do { if (trace && nt < trace_cap) trace[nt] = cur.n; nt++; } while (0)
    /* 1 + 3 single-word passes, then the first-array pass  :211-254 */
    int iterations = 0;
    sort_and_extend(&cur, P, k, twin, 2); TRACE();
    for (int i = 1; i < 4; i++) { iterations++; sort_and_extend(&cur, P, k, twin, 2); TRACE(); }
    iterations++;
    sort_and_extend(&cur, P, k, twin, 2); TRACE();
    /* array loop with the stop rule  :263-296 */
    int partitionNumber = P;
    int64_t contigNumber = 0;
    while (iterations <= prm->max_iter) {
        iterations++;
        if (iterations >= prm->min_iter && iterations % 3 == 0) {
            int64_t current = cur.n;                                          /* count() :270 */
            if (contigNumber == current) break;
            contigNumber = current;
            if (prm->coalesce && partitionNumber >= 16 && current / partitionNumber <= 20) {
                partitionNumber = partitionNumber / 4 + 1;                    /* :277-281 */
                P = partitionNumber;
            }
        }
        sort_and_extend(&cur, P, k, twin, 2); TRACE();
    }
    if (n_trace) *n_trace = nt;
    int64_t len = orc_contigs_text(cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right,
                                   cur.n, k, prm->min_contig, twin, out, cap, n_contigs);
    if (rec_out) *rec_out = cur; else orc_free_records(&cur);
    return len;
}

/* k > 31: ReflexivDSMain64.assemblyFromKmer  P/ReflexivDSMain64.java:374-826.  prm->extras (default 1)
 * selects the from-counts extras: at iteration minimumIteration + 3 every record is doubled
 * into both orientations (:588), the doubled set is split into the members of mergeable pairs ("extendable")
 * and the rest ("unextendable", :593-605), only the extendable set is iterated further (:659-660), and after the
 * loop the two are united and records that share an end with a longer one are dropped (:672-712).  With extras
 * = 0 the loop of :621-661 iterates ALL records and nothing is split or filtered.  What differs
 * from the k <= 31 driver: the stop rule starts at minimumIteration + 3 (:621), the first time the
 * count repeats param.scramble goes 2 -> 3 instead of stopping (:639-645) and every later array pass
 * starts its emission marker at 1 (:7484-7486; Spark evaluates the passes defined after a count() at
 * the next action, so they see the new value), the coalesce of :651-655 is assigned to a variable the
 * loop does not read (no effect), the surviving records are sorted once more before they become
 * text (:714), and the header is ">Contig-<len>-<idx>" (:830-866).  prm->twin is ignored: this class
 * has the DS arithmetic only. */
int64_t orc_assemble_from_counts_w(const uint64_t *kmers, const int32_t *counts, int64_t n,
                                   const orc_params *prm,
                                   char *out, int64_t cap, int64_t *n_contigs,
                                   int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                                   orc_records *rec_out) {
    orc_params q = *prm; q.twin = ORC_TWIN_DS;
    const int k = q.k, kw = orc_sub_words(k);
    const int P = q.partitions > 0 ? q.partitions : 1;
    int64_t nt = 0;
    orc_records cur;
    records_before_loop(kmers, counts, n, &q, P, &cur);                       /* :458-531 */
    int iterations = 0;
    sort_and_extend(&cur, P, k, ORC_TWIN_DS, 2); TRACE();                     /* :533-540 */
    for (int i = 1; i < 4; i++) { iterations++; sort_and_extend(&cur, P, k, ORC_TWIN_DS, 2); TRACE(); }   /* :543-548 */
    iterations++;                                                             /* :553 */
    sort_and_extend(&cur, P, k, ORC_TWIN_DS, 2); TRACE();                     /* :550-563 */
    int64_t contigNumber = 0;
    int scramble = 2;                                                         /* U/DefaultParam.java:131 */
    orc_records unext; memset(&unext, 0, sizeof unext);
    int have_split = 0;
    while (iterations <= q.max_iter) {                                        /* :582 */
        iterations++;
        if (q.extras && iterations == q.min_iter + 3) {                       /* :584-619 */
            sort_records_w(&cur, kw);                                         /* :587 */
            {   /* DSReflexivAndForwardKmer :588 */
                orc_records d; rec_alloc(&d, 2 * cur.n + 1, 2 * cur.ext_off[cur.n] + 1, kw);
                d.n = orc_double_w(cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right, cur.n, k,
                                   d.key, d.marker, d.ext_off, d.ext, d.left, d.right);
                orc_free_records(&cur); cur = d;
            }
            sort_records_w(&cur, kw);                                         /* :590 */
            int64_t *ps = (int64_t *)xmalloc((size_t)(P + 1) * 8), *ops = (int64_t *)xmalloc((size_t)(P + 1) * 8);
            orc_partition_starts_w(cur.key, cur.n, kw, P, ps);
            orc_records pe, pu;
            rec_alloc(&pe, cur.n + 1, cur.ext_off[cur.n] + 1, kw); rec_alloc(&pu, cur.n + 1, cur.ext_off[cur.n] + 1, kw);
            pe.n = orc_key_filter_w(ORC_OP_EXTENDABLE_PAIRS, cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right, cur.n,
                                    ps, P, k, pe.key, pe.marker, pe.ext_off, pe.ext, pe.left, pe.right, ops);     /* :593 */
            pu.n = orc_key_filter_w(ORC_OP_UNEXTENDABLE, cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right, cur.n,
                                    ps, P, k, pu.key, pu.marker, pu.ext_off, pu.ext, pu.left, pu.right, ops);     /* :594 */
            orc_free_records(&cur);
            sort_records_w(&pe, kw); sort_records_w(&pu, kw);                 /* :601-602 */
            orc_records *two[2] = { &pe, &pu };
            for (int t = 0; t < 2; t++) {                                     /* :604-605 */
                orc_records *src = two[t], dst;
                orc_partition_starts_w(src->key, src->n, kw, P, ps);
                rec_alloc(&dst, src->n + 1, src->ext_off[src->n] + 1, kw);
                dst.n = orc_key_filter_w(ORC_OP_FIRST_OF_KEY, src->key, src->marker, src->ext_off, src->ext, src->left, src->right,
                                         src->n, ps, P, k, dst.key, dst.marker, dst.ext_off, dst.ext, dst.left, dst.right, ops);
                orc_free_records(src); *src = dst;
            }
            free(ps); free(ops);
            cur = pe; unext = pu; have_split = 1;                             /* ExtendableReflexivKmer / UnExtendableReflexivKmer */
        }
        if (iterations >= q.min_iter + 3 && iterations % 3 == 0) {            /* :621-622 */
            int64_t current = cur.n;                                          /* :633-635 */
            if (contigNumber == current) {                                    /* :639 */
                if (scramble == 2) { scramble = 3; contigNumber = current; }  /* :640-642 */
                else break;                                                   /* :644 */
            } else contigNumber = current;                                    /* :647 */
        }
        sort_and_extend(&cur, P, k, ORC_TWIN_DS, scramble == 3 ? 1 : 2); TRACE();   /* :659-660 / :667-669 */
    }
    if (have_split) {                                                         /* :672-712 */
        /* union: the extendable set's partitions, then the unextendable set's (:678) */
        orc_records u; rec_alloc(&u, cur.n + unext.n + 1, cur.ext_off[cur.n] + unext.ext_off[unext.n] + 1, kw);
        int64_t m = 0, w = 0;
        orc_records *two[2] = { &cur, &unext };
        for (int t = 0; t < 2; t++) {
            orc_records *r = two[t];
            for (int64_t i = 0; i < r->n; i++) {
                KEYCPY(KEY(u.key, m), KEY(r->key, i)); u.marker[m] = r->marker[i]; u.left[m] = r->left[i]; u.right[m] = r->right[i];
                const int64_t nw = r->ext_off[i + 1] - r->ext_off[i];
                memcpy(u.ext + w, r->ext + r->ext_off[i], (size_t)nw * 8);
                w += nw; m++; u.ext_off[m] = w;
            }
            orc_free_records(r);
        }
        u.n = m; cur = u;
        int64_t *ps = (int64_t *)xmalloc((size_t)(P + 1) * 8), *ops = (int64_t *)xmalloc((size_t)(P + 1) * 8);
        for (int side = 1; side <= 2; side++) {                               /* left ends (:689-697), then right ends (:699-707) */
            orc_records f; rec_alloc(&f, cur.n + 1, cur.ext_off[cur.n] + 1, kw);
            f.n = orc_flip_all_w(cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right, cur.n, k, side,
                                 f.key, f.marker, f.ext_off, f.ext, f.left, f.right);
            orc_free_records(&cur); cur = f;
            sort_records_w(&cur, kw);
            orc_partition_starts_w(cur.key, cur.n, kw, P, ps);
            orc_records g; rec_alloc(&g, cur.n + 1, cur.ext_off[cur.n] + 1, kw);
            g.n = orc_key_filter_w(ORC_OP_LONGER_OF_KEY, cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right, cur.n,
                                   ps, P, k, g.key, g.marker, g.ext_off, g.ext, g.left, g.right, ops);
            orc_free_records(&cur); cur = g;
        }
        free(ps); free(ops);
    }
#undef TRACE
    if (n_trace) *n_trace = nt;
    sort_records_w(&cur, kw);                                                 /* :714 */
    int64_t len = orc_contigs_text_w(cur.key, cur.marker, cur.ext_off, cur.ext, cur.left, cur.right,
                                     cur.n, k, q.min_contig, out, cap, n_contigs);
    if (rec_out) *rec_out = cur; else orc_free_records(&cur);
    return len;
}

/* --------------------------------------------------------- synthetic reads */

uint64_t orc_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define SYNTH_TAG_GENOME 0x47454E4F4D45ULL
#define SYNTH_TAG_PAIRS  0x5041495253ULL
#define SYNTH_TAG_ERRORS 0x4552524F5253ULL

void orc_synth_genome(uint64_t seed, int64_t genome_len, uint64_t *packed) {
    uint64_t sg = orc_splitmix64(seed ^ SYNTH_TAG_GENOME);
    int64_t nw = (genome_len + 31) / 32;
    for (int64_t j = 0; j < nw; j++) packed[j] = orc_splitmix64(sg + (uint64_t)j);
}

static inline unsigned genome_base(const uint64_t *g, int64_t i) {
    return (unsigned)((g[i >> 5] >> (62 - 2 * (i & 31))) & 3);
}

void orc_synth_reads(uint64_t seed, const uint64_t *genome, int64_t genome_len,
                     int64_t first_read, int64_t n_reads, int read_len, uint32_t err_per_2_32,
                     char *bases) {
    static const char NUC[4] = { 'A', 'C', 'G', 'T' };
    uint64_t sp = orc_splitmix64(seed ^ SYNTH_TAG_PAIRS);
    uint64_t se = orc_splitmix64(seed ^ SYNTH_TAG_ERRORS);
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
#endif
    for (int64_t t = 0; t < n_reads; t++) {
        int64_t r = first_read + t;
        uint64_t pair = (uint64_t)r >> 1;
        int mate = (int)(r & 1);
        uint64_t u = orc_splitmix64(sp + pair);
        int64_t s = (int64_t)(u & 0xFFFF) + (int64_t)((u >> 16) & 0xFFFF)
                  + (int64_t)((u >> 32) & 0xFFFF) + (int64_t)((u >> 48) & 0xFFFF);
        int64_t frag = 350 + ((s - 131070) * 35) / 37837;   /* ~N(350,35); truncating division */
        if (frag < read_len) frag = read_len;
        if (frag > genome_len) frag = genome_len;
        uint64_t v = orc_splitmix64(u);
        int64_t start = (int64_t)((v >> 1) % (uint64_t)(genome_len - frag + 1));
        int strand = (int)(v & 1);
        int is_rc = mate ^ strand;
        int64_t pos = is_rc ? start + frag - read_len : start;
        char *dst = bases + t * (int64_t)read_len;
        for (int j = 0; j < read_len; j++) {
            unsigned b = is_rc ? 3u - genome_base(genome, pos + read_len - 1 - j)
                               : genome_base(genome, pos + j);
            uint64_t e = orc_splitmix64(se + (uint64_t)r * (uint64_t)read_len + (uint64_t)j);
            if ((uint32_t)e < err_per_2_32) b = (b + 1u + (unsigned)((e >> 32) % 3u)) & 3u;
            dst[j] = NUC[b];
        }
    }
}
