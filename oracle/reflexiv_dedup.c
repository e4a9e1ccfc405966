/*
 * reflexiv_dedup.c -- CPU restatement of the contig RC de-duplication of P/ReflexivDSDynamicKmerDedup.java.
 * TEST INFRASTRUCTURE (part of liborc.so): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * Driver `assemblyFromKmer` (:138-339): three rounds of
 *     marker k-mers of every contig >= 300 bases (forward seeds at the 31-base block starts, probes at a few windows:
 *     ReverseComplementKmerMarkerExtraction :2674-3096 in round 1, ForwardAndReverseComplementKmerMarkerExtraction
 *     :2206-2673 in rounds 2 and 3)
 *  -> sort("kmerBinary")                                                              (:180, :242, :295)
 *  -> DSMarkerKmerSelection (:1788-1870): a probe of a shorter contig (ties: the later id) meeting a seed of the same
 *     31-mer names the pair (shorter id << 32 | longer id)
 *  -> groupBy().count() >= 2 (:186-194)  -> DSMarkerKmerShorterID (:3186-3207): a row {-1, longer id} under the shorter id
 *  -> union with the contigs, sort("count"), DSShorterRCContigSeqAndTargetExtraction (:3097-3132): the shorter contig
 *     takes its target's id;  sort("count") again
 *  -> the removal class of the round (DSShorterRCContigRemoval :1405-1558, DSShorterForwardAndRCContigRemoval[Array]
 *     :508-729 / :959-1175): the contigs that share an id are merged into the longest by 15-mer seed voting
 *     (`merge2RCContigs`), an unmatched short contig goes back to the pool;
 * zipWithIndex renumbers the survivors between rounds; TagRowContigDSID (:3397-3443) writes the text.
 *
 * Contigs are handled at SEQUENCE level (one byte per base): the reference's block helpers `leftShiftArray` (:1647-1683),
 * `leftShiftOutFromArray` (:1685-1713), `combineTwoLongBlocks` (:1715-1786) and `binaryBlockReverseComplementary`
 * (:1559-1591) are suffix / prefix / concatenation / reverse complement of left-aligned 31-base blocks with a trailing 01
 * terminator; a 15-mer read past a contig's end sees that terminator as one C followed by A's (`seed_at` below), which is
 * what `(int)(block[0] >>> 2*(32-15))` yields there.  tests/test_oracle_dedup.py checks every stage against vectors made
 * by the reference's own classes (tests/golden/make_dedup_vectors.py).
 *
 * What sits between two operator classes is Spark's and follows the order contract: ONE logical
 * partition, stable sorts on the SIGNED 64-bit column, union = left rows then right rows, groupBy().count() in ascending
 * key order, zipWithIndex = position.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *xm(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle dedup: out of memory\n"); abort(); }
    return p;
}

typedef struct { uint8_t *s; int64_t n; int64_t id; int is_marker; int64_t target; } drow;    /* a row of (blocks, count) */

static drow row_seq(const uint8_t *s, int64_t n, int64_t id) {
    drow r; r.s = (uint8_t *)xm((size_t)n); memcpy(r.s, s, (size_t)n); r.n = n; r.id = id; r.is_marker = 0; r.target = 0;
    return r;
}
static drow row_copy(const drow *a) {
    drow r = *a; r.s = (uint8_t *)xm((size_t)a->n); memcpy(r.s, a->s, (size_t)a->n);
    return r;
}

/* a marker row {-1L, target} read as blocks by the removal classes: 31 T's, then the bases currentKmerSizeFromBinaryBlockArray
 * (:1636-1645) finds in `target` (numberOfTrailingZeros(0) = 64 gives -1 there) */
static drow marker_as_seq(int64_t target, int64_t id) {
    const uint64_t t = (uint64_t)target;
    const int tz = t ? __builtin_ctzll(t) : 64;
    const int last = 32 - tz / 2 - 1;
    const int64_t n = 31 + last;
    drow r; r.s = (uint8_t *)xm((size_t)(n > 0 ? n : 1)); r.n = n > 0 ? n : 0; r.id = id; r.is_marker = 0; r.target = 0;
    for (int64_t i = 0; i < r.n; i++) r.s[i] = i < 31 ? 3 : (uint8_t)((t >> (2 * (31 - (i - 31)))) & 3);
    return r;
}

typedef struct { drow *v; int64_t n, cap; } dlist;
static void dl_push(dlist *l, drow r) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 16; l->v = (drow *)realloc(l->v, (size_t)l->cap * sizeof(drow)); }
    l->v[l->n++] = r;
}
static void dl_free(dlist *l) { for (int64_t i = 0; i < l->n; i++) free(l->v[i].s); free(l->v); l->v = NULL; l->n = l->cap = 0; }

/* 31-mer at position p, left aligned in the top 62 bits, 01 terminator below (a one-block seed / probe) */
static uint64_t mer31(const uint8_t *s, int64_t p) {
    uint64_t x = 0;
    for (int j = 0; j < 31; j++) x = (x << 2) | s[p + j];
    return (x << 2) | 1;
}
/* binaryLongReverseComplementary (:2877-2906) of such a block */
static uint64_t mer31_rc(uint64_t m) {
    uint64_t x = 0;
    for (int j = 0; j < 31; j++) x = (x << 2) | (((m >> (2 * (j + 1))) & 3) ^ 3);
    return (x << 2) | 1;
}

/* buildingAlongFromThreeInt (:2908-2934) / getLeftMarker, getRightMarker, getReflexivMarker (:1882-1902, :2089-2092) */
static int64_t attr3(int marker, int64_t left, int64_t right) {
    if (left >= 500000000) left = 500000000; else if (left <= -500000000) left = 1000000000; else if (left < 0) left = 500000000 - left;
    if (right >= 1000000000) right = 1000000000; else if (right <= -1000000000) right = 2000000000; else if (right < 0) right = 1000000000 - right;
    return (int64_t)(((uint64_t)marker << 62) | ((uint64_t)(uint32_t)left << 32) | (uint64_t)(uint32_t)right);
}
static int attr_marker(int64_t a) { return (int)((uint64_t)a >> 62); }
static int32_t attr_left(int64_t a) {
    int32_t l = (int32_t)((uint64_t)a >> 32) & ~(3 << 30);
    if (l > 500000000) l = 500000000 - l;
    return l;
}
static int32_t attr_right(int64_t a) {
    int32_t r = (int32_t)a;
    if (r > 1000000000) r = 1000000000 - r;
    return r;
}

typedef struct { int64_t kmer, attr; } mrow;
typedef struct { mrow *v; int64_t n, cap; } mlist;
static void ml_push(mlist *l, int64_t k, int64_t a) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 256; l->v = (mrow *)realloc(l->v, (size_t)l->cap * sizeof(mrow)); }
    l->v[l->n].kmer = k; l->v[l->n].attr = a; l->n++;
}

/* [Forward]AndReverseComplementKmerMarkerExtraction.call (:2213-2227 / :2681-2695): seeds and probes of one contig */
static void markers_of(const drow *c, int both, mlist *out) {
    const int64_t L = c->n;
    if (L < 300) return;
    const int M = 31;
    const int64_t nb = (L - 1) / 31 + 1;                              /* blocks of the contig */
    const int64_t fa = attr3(1, L, (int32_t)c->id), pa = attr3(2, L, (int32_t)c->id);
    for (int64_t i = 0; i < nb - 1; i++) ml_push(out, (int64_t)mer31(c->s, 31 * i), fa);      /* getForwardKmerBinary :2697-2711 */
    if (L % 31 == 0) ml_push(out, (int64_t)mer31(c->s, 31 * (nb - 1)), fa);
    int64_t w[5][2]; int nw = 0;                                      /* getRCKmerProbBinary :2713-2875: the probe windows */
#define WIN(a, b) do { w[nw][0] = (a); w[nw][1] = (b); nw++; } while (0)
    if (L >= 4000) {
        WIN(0, M); WIN(1000 - M + 1, 1000); WIN((L - 2 * M) / 2, (L - 2 * M) / 2 + M); WIN(L - 1000 - M + 1, L - 1000); WIN(L - 2 * M, L - M);
    } else if (L >= 2000) {
        WIN(0, M); WIN(600 - M + 1, 600); WIN((L - 2 * M) / 2, (L - 2 * M) / 2 + M); WIN(L - 600 - M + 1, L - 600); WIN(L - 2 * M, L - M);
    } else {
        WIN(0, M); WIN((L - 2 * M) / 3, (L - 2 * M) / 3 + M); WIN((L - 2 * M) * 2 / 3, (L - 2 * M) * 2 / 3 + M); WIN(L - 2 * M, L - M);
    }
#undef WIN
    for (int q = 0; q < nw; q++)
        for (int64_t i = w[q][0]; i < w[q][1]; i++) {
            const uint64_t f = mer31(c->s, i);
            if (both) ml_push(out, (int64_t)f, pa);
            ml_push(out, (int64_t)mer31_rc(f), pa);
        }
}

static int cmp_mrow(const void *a, const void *b) {                   /* (merge sort below keeps ties in order) */
    const int64_t x = ((const mrow *)a)->kmer, y = ((const mrow *)b)->kmer;
    return x < y ? -1 : x > y;
}
static void stable_sort_mrows(mrow *v, int64_t n) {
    if (n < 2) return;
    mrow *t = (mrow *)xm((size_t)n * sizeof(mrow));
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
            while (i < mid && j < hi) t[k++] = cmp_mrow(&v[j], &v[i]) < 0 ? v[j++] : v[i++];
            while (i < mid) t[k++] = v[i++];
            while (j < hi) t[k++] = v[j++];
        }
        memcpy(v, t, (size_t)n * sizeof(mrow));
    }
    free(t);
}

/* DSMarkerKmerSelection.call (:1796-1868) over the sorted markers -> pair ids (shorter << 32 | longer), in emission order */
typedef struct { int64_t *v; int64_t n, cap; } ilist;
static void il_push(ilist *l, int64_t x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 256; l->v = (int64_t *)realloc(l->v, (size_t)l->cap * 8); }
    l->v[l->n++] = x;
}
static int64_t pair_id(int64_t a_short, int64_t a_long) {             /* buildingAlongFromTwoInt (:1870-1880) */
    return (int64_t)(((uint64_t)(uint32_t)attr_right(a_short) << 32) | (uint64_t)(uint32_t)attr_right(a_long));
}
static void probe_against(const mrow *p, const mrow *seed, ilist *out) {
    if (attr_left(p->attr) < attr_left(seed->attr)) il_push(out, pair_id(p->attr, seed->attr));
    else if (attr_left(p->attr) == attr_left(seed->attr) && attr_right(p->attr) > attr_right(seed->attr))
        il_push(out, pair_id(p->attr, seed->attr));
}
static void marker_selection(const mrow *m, int64_t n, ilist *out) {
    mrow longest; longest.kmer = 1; longest.attr = 1;                 /* Row LongestKmer = (1L, 1L)  :1789 */
    mlist shorter = {0};
    for (int64_t q = 0; q < n; q++) {
        const mrow *s = &m[q];
        if (attr_marker(s->attr) == 1) {
            if (s->kmer == longest.kmer) {
                if (attr_left(s->attr) > attr_left(longest.attr)) longest = *s;
            } else {
                for (int64_t i = 0; i < shorter.n; i++) {
                    if (shorter.v[i].kmer == longest.kmer) probe_against(&shorter.v[i], &longest, out);
                    else if (shorter.v[i].kmer == s->kmer) probe_against(&shorter.v[i], s, out);
                }
                longest = *s;
                shorter.n = 0;
            }
        } else ml_push(&shorter, s->kmer, s->attr);
    }
    for (int64_t i = 0; i < shorter.n; i++)
        if (shorter.v[i].kmer == longest.kmer) probe_against(&shorter.v[i], &longest, out);
    free(shorter.v);
}

static int cmp_i64(const void *a, const void *b) {
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

static void stable_sort_rows(drow *v, int64_t n) {                    /* sort("count"): signed, stable */
    if (n < 2) return;
    drow *t = (drow *)xm((size_t)n * sizeof(drow));
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
            while (i < mid && j < hi) t[k++] = v[j].id < v[i].id ? v[j++] : v[i++];
            while (i < mid) t[k++] = v[i++];
            while (j < hi) t[k++] = v[j++];
        }
        memcpy(v, t, (size_t)n * sizeof(drow));
    }
    free(t);
}

/* ---- merge2RCContigs -------------------------------------------------------------------------------------------- */

/* (int)(leftShiftOutFromArray(leftShiftArray(c, p), 15)[0] >>> 2*(32-15)): the 15-mer at p; past the end the block's 01
 * terminator reads as one C, then A's */
static uint32_t seed_at(const uint8_t *s, int64_t n, int64_t p) {
    uint32_t x = 0;
    for (int j = 0; j < 15; j++) {
        const int64_t q = p + j;
        x = (x << 2) | (q < n ? s[q] : q == n ? 1u : 0u);
    }
    return x;
}

typedef struct { uint32_t *key; int32_t *pos; uint64_t mask; } seedmap;
static void sm_init(seedmap *m, int64_t n_seeds) {
    uint64_t cap = 64;
    while (cap < (uint64_t)n_seeds * 2 + 8) cap *= 2;
    m->key = (uint32_t *)xm(cap * 4); m->pos = (int32_t *)xm(cap * 4); m->mask = cap - 1;
    memset(m->pos, 0xFF, cap * 4);                                    /* pos -1 = empty */
}
static void sm_put(seedmap *m, uint32_t k, int32_t pos) {             /* HashMap.put: a later position replaces an earlier one */
    uint64_t h = ((uint64_t)k * 0x9E3779B97F4A7C15ULL >> 20) & m->mask;
    while (m->pos[h] >= 0 && m->key[h] != k) h = (h + 1) & m->mask;
    m->key[h] = k; m->pos[h] = pos;
}
static int32_t sm_get(const seedmap *m, uint32_t k) {
    uint64_t h = ((uint64_t)k * 0x9E3779B97F4A7C15ULL >> 20) & m->mask;
    while (m->pos[h] >= 0) { if (m->key[h] == k) return m->pos[h]; h = (h + 1) & m->mask; }
    return -1;
}

typedef struct { int32_t *v; int64_t n, cap; } i32list;
static void i32_push(i32list *l, int32_t x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1024; l->v = (int32_t *)realloc(l->v, (size_t)l->cap * 4); }
    l->v[l->n++] = x;
}
static int cmp_i32(const void *a, const void *b) { const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return x < y ? -1 : x > y; }

static void query_into(const seedmap *m, const uint8_t *q, int64_t qn, i32list *dist) {
    for (int64_t i = 0; i < qn; i++) {
        const int32_t locus = sm_get(m, seed_at(q, qn, i));
        if (locus >= 0) i32_push(dist, (int32_t)(i + 1 - locus));
    }
}
/* the vote over the sorted distance list (:1478-1497 with min_votes 3, :578-597 / :617-636 with 4) */
static int32_t vote(i32list *dist, int min_votes) {
    qsort(dist->v, (size_t)dist->n, 4, cmp_i32);
    const int64_t total = dist->n;
    int32_t lastDistance = 0, lastFrequency = 0;
    for (int64_t i = 0; i < total; i++) {
        const int32_t d = dist->v[i];
        if (d - lastDistance >= -1 && d - lastDistance <= 1) {
            lastFrequency++;
            if ((double)lastFrequency / (double)total >= 0.3 && lastFrequency >= min_votes) return d;
        } else { lastFrequency = 1; lastDistance = d; }
    }
    return -1;
}
static void rc_seq(const uint8_t *s, int64_t n, uint8_t *out) { for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)(3 - s[n - 1 - i]); }

/* leftShiftOutFromArray (:1685-1713) at sequence level: the first p bases; p > n returns everything; p = 0 returns the first
 * block (never asked for here: p is a positive distance) */
static int64_t prefix_len(int64_t n, int64_t p) { return p > n ? n : p; }

typedef struct { uint8_t *s; int64_t n; } dseq;
static dseq concat(const uint8_t *a, int64_t an, const uint8_t *b, int64_t bn) {      /* combineTwoLongBlocks :1715-1786 */
    dseq r; r.n = an + bn; r.s = (uint8_t *)xm((size_t)r.n);
    memcpy(r.s, a, (size_t)an); memcpy(r.s + an, b, (size_t)bn);
    return r;
}

/* variant 0: DSShorterRCContigRemoval.merge2RCContigs (:1462-1557); variant 1: the forward-then-RC form of
 * DSShorterForwardAndRCContigRemoval[Array] (:565-728).  Returns the (possibly extended) long contig; an unmatched short contig
 * is appended to `pool`. */
static dseq merge2(dseq lng, const drow *sh, int variant, dlist *pool) {
    const int64_t LN = lng.n;
    seedmap m; sm_init(&m, LN / 15 + 2);
    for (int64_t i = 0; i <= LN; i += 15) sm_put(&m, seed_at(lng.s, LN, i), (int32_t)(i + 1));
    i32list dist = {0};
    dseq out = lng;
    int done = 0;
    if (variant == 1) {
        const int64_t FN = sh->n;
        query_into(&m, sh->s, FN, &dist);
        const int32_t fd = vote(&dist, 4);
        if (fd == -1 || fd == 0) {
        } else if (fd < 0) {
            int64_t flank = FN - (LN + fd);
            if (flank > FN) flank = FN;                                /* (the reference would throw on a negative shift) */
            if (flank > 0) { out = concat(lng.s, LN, sh->s + (FN - flank), flank); free(lng.s); done = 1; }
        } else {
            const int64_t p = prefix_len(FN, fd);
            out = concat(sh->s, p, lng.s, LN); free(lng.s); done = 1;
        }
    }
    if (!done) {
        const int64_t SN = sh->n;
        uint8_t *rc = (uint8_t *)xm((size_t)SN);
        rc_seq(sh->s, SN, rc);
        query_into(&m, rc, SN, &dist);                                 /* (variant 1: the forward distances stay in the list) */
        const int32_t fd = vote(&dist, variant == 0 ? 3 : 4);
        if (fd == -1) dl_push(pool, row_copy(sh));
        else if (fd == 0) {
        } else if (fd < 0) {
            int64_t flank = SN - (LN + fd);
            if (flank > SN) flank = SN;
            if (flank > 0) { out = concat(lng.s, LN, rc + (SN - flank), flank); free(lng.s); }
        } else {
            const int64_t p = prefix_len(SN, fd);
            out = concat(rc, p, lng.s, LN); free(lng.s);
        }
        free(rc);
    }
    free(m.key); free(m.pos); free(dist.v);
    return out;
}

/* the removal classes' call() (:1413-1460 / :516-563): rows sorted by id -> surviving contigs, in emission order */
static void removal(const dlist *rows, int variant, dlist *out) {
    const drow *longest = NULL;
    const drow **shorts = (const drow **)xm((size_t)(rows->n + 1) * sizeof(drow *));
    int64_t ns = 0;
    for (int64_t q = 0; q <= rows->n; q++) {
        const drow *s = q < rows->n ? &rows->v[q] : NULL;
        if (s && !longest) { longest = s; continue; }
        if (s && s->id == longest->id) {
            if (s->n > longest->n) { shorts[ns++] = longest; longest = s; } else shorts[ns++] = s;
            continue;
        }
        if (longest) {                                                 /* a new id, or the end of the partition: flush the group */
            dseq l; l.n = longest->n; l.s = (uint8_t *)xm((size_t)l.n); memcpy(l.s, longest->s, (size_t)l.n);
            for (int64_t i = 0; i < ns; i++) l = merge2(l, shorts[i], variant, out);
            drow r; r.s = l.s; r.n = l.n; r.id = 0; r.is_marker = 0; r.target = 0;
            dl_push(out, r);
            ns = 0;
        }
        longest = s;
    }
    free(shorts);
}

/* one round: contigs with ids -> surviving contigs */
static void dedup_round(const dlist *contigs, int rnd, dlist *out, int64_t *n_pairs, int64_t *n_cand) {
    mlist mk = {0};
    for (int64_t i = 0; i < contigs->n; i++) markers_of(&contigs->v[i], rnd > 1, &mk);
    stable_sort_mrows(mk.v, mk.n);
    ilist pairs = {0};
    marker_selection(mk.v, mk.n, &pairs);
    if (n_pairs) *n_pairs = pairs.n;
    qsort(pairs.v, (size_t)pairs.n, 8, cmp_i64);
    /* union: the contigs, then a marker row per pair id seen at least twice (ascending pair id) */
    dlist u = {0};
    for (int64_t i = 0; i < contigs->n; i++) dl_push(&u, row_copy(&contigs->v[i]));
    int64_t nc = 0;
    for (int64_t i = 0; i < pairs.n;) {
        int64_t j = i;
        while (j < pairs.n && pairs.v[j] == pairs.v[i]) j++;
        if (j - i >= 2) {
            drow r; r.s = (uint8_t *)xm(1); r.n = 0; r.is_marker = 1;
            r.id = (int64_t)(int32_t)((uint64_t)pairs.v[i] >> 32);     /* DSMarkerKmerShorterID :3192-3199 */
            r.target = (int64_t)(int32_t)pairs.v[i];
            dl_push(&u, r);
            nc++;
        }
        i = j;
    }
    if (n_cand) *n_cand = nc;
    free(pairs.v); free(mk.v);
    stable_sort_rows(u.v, u.n);
    /* DSShorterRCContigSeqAndTargetExtraction.call :3102-3131 */
    dlist st = {0};
    const drow *last = NULL;
    for (int64_t q = 0; q < u.n; q++) {
        const drow *s = &u.v[q];
        if (!last) { last = s; continue; }
        if (s->id == last->id) {
            /* (a leftover marker row in the contig's place is read as blocks by the next class: marker_as_seq) */
            if (s->is_marker) { drow r = last->is_marker ? marker_as_seq(last->target, 0) : row_copy(last); r.id = s->target; dl_push(&st, r); }
            else if (last->is_marker) { drow r = row_copy(s); r.id = last->target; dl_push(&st, r); }
            last = NULL;                                               /* (two contigs under one id: both are dropped, as written) */
        } else {
            dl_push(&st, last->is_marker ? marker_as_seq(last->target, last->id) : row_copy(last));
            last = s;
        }
    }
    if (last) dl_push(&st, last->is_marker ? marker_as_seq(last->target, last->id) : row_copy(last));
    dl_free(&u);
    stable_sort_rows(st.v, st.n);
    removal(&st, rnd == 1 ? 0 : 1, out);
    dl_free(&st);
}

/* Contigs (bases 0..3, contig i = bases[off[i] .. off[i+1]), ids = positions) -> the three rounds.
 * round_n[r] / round_bases[r]: survivors after round r+1 and their total length; the survivors of the last round are written
 * to out_bases / out_off (capacity cap_bases / cap_contigs); returns the number of contigs after round 3, or -1 - needed
 * bases if a capacity is short. */
int64_t orc_dedup_contigs(const uint8_t *bases, const int64_t *off, int64_t n, uint8_t *out_bases, int64_t cap_bases, int64_t *out_off,
                          int64_t cap_contigs, int64_t *round_n, int64_t *round_bases, int64_t *round_pairs, int64_t *round_cand,
                          uint8_t *r1_bases, int64_t *r1_off, uint8_t *r2_bases, int64_t *r2_off) {
    dlist cur = {0};
    for (int64_t i = 0; i < n; i++) dl_push(&cur, row_seq(bases + off[i], off[i + 1] - off[i], i));
    for (int rnd = 1; rnd <= 3; rnd++) {
        dlist nxt = {0};
        dedup_round(&cur, rnd, &nxt, round_pairs ? &round_pairs[rnd - 1] : NULL, round_cand ? &round_cand[rnd - 1] : NULL);
        dl_free(&cur);
        cur = nxt;
        int64_t tb = 0;
        for (int64_t i = 0; i < cur.n; i++) { cur.v[i].id = i; tb += cur.v[i].n; }             /* zipWithIndex */
        if (round_n) round_n[rnd - 1] = cur.n;
        if (round_bases) round_bases[rnd - 1] = tb;
        uint8_t *rb = rnd == 1 ? r1_bases : rnd == 2 ? r2_bases : NULL;
        int64_t *ro = rnd == 1 ? r1_off : rnd == 2 ? r2_off : NULL;
        if (rb && ro) {
            int64_t p = 0;
            for (int64_t i = 0; i < cur.n; i++) { ro[i] = p; memcpy(rb + p, cur.v[i].s, (size_t)cur.v[i].n); p += cur.v[i].n; }
            ro[cur.n] = p;
        }
    }
    int64_t tb = 0;
    for (int64_t i = 0; i < cur.n; i++) tb += cur.v[i].n;
    if (tb > cap_bases || cur.n > cap_contigs) { dl_free(&cur); return -1 - tb; }
    int64_t p = 0;
    for (int64_t i = 0; i < cur.n; i++) { out_off[i] = p; memcpy(out_bases + p, cur.v[i].s, (size_t)cur.v[i].n); p += cur.v[i].n; }
    out_off[cur.n] = p;
    const int64_t m = cur.n;
    dl_free(&cur);
    return m;
}

/* TagRowContigDSID.call + changeLine (:3397-3443): ">Contig-<len>-<idx>\n" + the sequence in lines of 10,000,000, one text row
 * per contig of at least min_contig bases (idx = position among ALL contigs); returns the length (writes up to cap) */
int64_t orc_dedup_text(const uint8_t *bases, const int64_t *off, int64_t n, int min_contig, char *out, int64_t cap) {
    int64_t pos = 0;
    const int64_t LIM = 10000000;
    for (int64_t i = 0; i < n; i++) {
        const int64_t L = off[i + 1] - off[i];
        if (L < min_contig) continue;
        char hdr[64];
        const int hl = snprintf(hdr, sizeof hdr, ">Contig-%lld-%lld\n", (long long)L, (long long)i);
        for (int j = 0; j < hl; j++) { if (pos < cap) out[pos] = hdr[j]; pos++; }
        for (int64_t j = 0; j < L; j++) {
            if (j > 0 && j % LIM == 0) { if (pos < cap) out[pos] = '\n'; pos++; }
            if (pos < cap) out[pos] = "ACGT"[bases[off[i] + j]];
            pos++;
        }
        if (pos < cap) out[pos] = '\n';
        pos++;
    }
    return pos;
}
