/*
 * reflexiv_dynamic.c -- CPU restatement of the dynamic-k record format and passes:
 * P/ReflexivDSDynamicKmerFirstFour.java (DSkmerRandomReflection :2509-2762, DSExtendReflexivKmer :1581-2373) and
 * P/ReflexivDSDynamicKmerIteration.java (DSExtendReflexivKmerToArrayLoop :465-1249).  TEST INFRASTRUCTURE (part of
 * liborc.so): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * The third record layout: a key of ANY length (a (k-1)-mer of whichever k the reduction kept) as left-aligned 31-base
 * blocks with a 01 terminator after the last base, an attribute long (marker << 62 | left << 32 | right, negatives stored
 * as 30000 - v: buildingAlongFromThreeInt FirstFour:2340-2366, getLeftMarker / getRightMarker :2318-2338) and an extension
 * in the same left-aligned form (one long in the first four passes, an array afterwards).  The classes are written on
 * four block helpers -- leftShiftArray, leftShiftOutFromArray, combineTwoLongBlocks, currentKmerSizeFromBinaryBlockArray --
 * which are suffix, prefix, concatenation and length of a base string; this file works on base strings (one byte per
 * base) and forms blocks only where the reference's behaviour depends on them: the ORDER of sort("k-1") (Spark orders
 * array<long> element by element as signed longs, a proper prefix first).
 *
 * One pass (DSExtendReflexivKmer.call FirstFour:1603-1763, DSExtendReflexivKmerToArrayLoop.call Iteration:487-...): the
 * scan with a one-row holder and the toggling orientation of SURVEY.md B.5, except that a row meets the holder when their
 * keys are EQUAL OR ONE IS A PREFIX OF THE OTHER (dynamicSubKmerComparator), the forward row must not be the shorter one,
 * the merged key keeps the longer key's length, and left / right follow the dynamic formulas (reflexivExtend
 * FirstFour:1957-2120).  tests/test_oracle_dynamic.py checks every operator against vectors made by the reference's own
 * classes (tests/golden/make_dynamic_vectors.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *key; int32_t klen;
    const uint8_t *ext; int32_t elen;
    int32_t marker, left, right;
} drec;

/* buildingAlongFromThreeInt followed by getLeftMarker / getRightMarker: what a value reads back as */
static int32_t attr_clamp(int32_t v) {
    if (v >= 30000) return 30000;
    if (v <= -30000) return -30000;
    return v;
}

/* block j of a base string as Spark sees it (signed) */
static int64_t block_of(const uint8_t *s, int32_t n, int32_t j) {
    uint64_t x = 0;
    const int32_t b0 = 31 * j;
    int32_t m = n - b0;
    if (m > 31) m = 31;
    for (int32_t i = 0; i < m; i++) x |= (uint64_t)s[b0 + i] << (2 * (31 - i));
    if (b0 + 31 >= n) x |= 1ULL << (2 * (31 - m));            /* the last block carries the terminator */
    return (int64_t)x;
}
static int32_t blocks_of_len(int32_t n) { return n <= 0 ? 1 : (n - 1) / 31 + 1; }

/* ordering of two keys as array<long>: element by element (signed), a proper prefix first */
static int key_cmp(const uint8_t *a, int32_t an, const uint8_t *b, int32_t bn) {
    const int32_t na = blocks_of_len(an), nb = blocks_of_len(bn);
    const int32_t n = na < nb ? na : nb;
    for (int32_t j = 0; j < n; j++) {
        const int64_t x = block_of(a, an, j), y = block_of(b, bn, j);
        if (x != y) return x < y ? -1 : 1;
    }
    return (na > nb) - (na < nb);
}

/* stable merge sort of row indices by key */
void orc_dyn_sort_perm(const uint8_t *key_bases, const int64_t *key_off, int64_t n, int64_t *perm) {
    int64_t *tmp = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * 8);
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                const int64_t a = perm[i], b = perm[j];
                const int c = key_cmp(key_bases + key_off[b], (int32_t)(key_off[b + 1] - key_off[b]), key_bases + key_off[a],
                                      (int32_t)(key_off[a + 1] - key_off[a]));
                if (c < 0) tmp[k++] = perm[j++]; else tmp[k++] = perm[i++];
            }
            while (i < mid) tmp[k++] = perm[i++];
            while (j < hi) tmp[k++] = perm[j++];
        }
        memcpy(perm, tmp, (size_t)n * 8);
    }
    free(tmp);
}

int orc_dyn_keys_equal(const uint8_t *a, int32_t an, const uint8_t *b, int32_t bn) { return an == bn && memcmp(a, b, (size_t)an) == 0; }

/* ---- output set ---------------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *key; int64_t *key_off; uint8_t *ext; int64_t *ext_off; int32_t *marker, *left, *right;
    int64_t n, cap_n, cap_key, cap_ext, need_key, need_ext;
} dout;

static void put(dout *o, const uint8_t *k1, int32_t k1n, const uint8_t *k2, int32_t k2n, const uint8_t *e1, int32_t e1n, const uint8_t *e2,
                int32_t e2n, int32_t marker, int32_t left, int32_t right) {
    const int64_t kn = (int64_t)k1n + k2n, en = (int64_t)e1n + e2n;
    if (o->n < o->cap_n && o->need_key + kn <= o->cap_key && o->need_ext + en <= o->cap_ext) {
        uint8_t *kd = o->key + o->need_key, *ed = o->ext + o->need_ext;
        memcpy(kd, k1, (size_t)k1n); memcpy(kd + k1n, k2, (size_t)k2n);
        memcpy(ed, e1, (size_t)e1n); memcpy(ed + e1n, e2, (size_t)e2n);
        o->key_off[o->n] = o->need_key; o->ext_off[o->n] = o->need_ext;
        o->marker[o->n] = marker; o->left[o->n] = left; o->right[o->n] = right;
    }
    o->need_key += kn; o->need_ext += en; o->n++;
}

/* singleKmerRandomizer (FirstFour:1857-1930 / Iteration singleKmerRandomizer): the record in orientation m */
static void flip_emit(dout *o, const drec *r, int m) {
    /* (written for any lengths as Iteration's array form does it: combined = key + ext, key' = combined[|ext|:],
     * ext' = combined[:|ext|]; FirstFour's single-long form is the same whenever |ext| <= |key|, which its four passes keep) */
    if ((r->marker == 1 && m == 2) || (r->marker == 2 && m == 1)) {
        const int32_t kl = r->klen, el = r->elen;
        uint8_t *c = (uint8_t *)malloc((size_t)(kl + el + 1));
        if (r->marker == 1) {
            memcpy(c, r->key, (size_t)kl); memcpy(c + kl, r->ext, (size_t)el);
            put(o, c + el, kl, NULL, 0, c, el, NULL, 0, 2, r->left, r->right);
        } else {
            memcpy(c, r->ext, (size_t)el); memcpy(c + el, r->key, (size_t)kl);
            put(o, c, kl, NULL, 0, c + kl, el, NULL, 0, 1, r->left, r->right);
        }
        free(c);
    } else {
        put(o, r->key, r->klen, NULL, 0, r->ext, r->elen, NULL, 0, r->marker, r->left, r->right);
    }
}

/* reflexivExtend (FirstFour:1957-2120, Iteration likewise): forward f + reflected r -> one record in orientation m */
static void merge_emit(dout *o, const drec *f, const drec *r, int32_t bubble, int m) {
    const int32_t extra = f->klen > r->klen ? f->klen - r->klen : 0;
    const drec *lg = f->klen >= r->klen ? f : r;                  /* the longer key is kept */
    int32_t left, right;
    if (bubble < 0) {
        left = r->left >= 0 ? r->left : f->left - r->elen;
        right = f->right >= 0 ? f->right : r->right - f->elen - extra;
    } else if (f->left > 0) {
        left = bubble;
        right = f->right >= 0 ? f->right : r->right - f->elen - extra;
    } else {
        left = r->left >= 0 ? r->left : f->left - r->elen;
        right = bubble - extra;
    }
    left = attr_clamp(left); right = attr_clamp(right);
    if (m == 2) {
        /* combined = L + S; ext' = P + combined[:|S|]; key' = combined[|S|:] */
        const int32_t S = f->elen, L = lg->klen;
        if (S <= L) {
            put(o, lg->key + S, L - S, f->ext, S, r->ext, r->elen, lg->key, S, 2, left, right);
        } else {                                                   /* (an extension longer than the key: not reached by the passes) */
            uint8_t *c = (uint8_t *)malloc((size_t)(L + S));
            memcpy(c, lg->key, (size_t)L); memcpy(c + L, f->ext, (size_t)S);
            put(o, c + S, L, NULL, 0, r->ext, r->elen, c, S, 2, left, right);
            free(c);
        }
    } else {
        /* combined = P + L; key' = combined[:|L|]; ext' = combined[|L|:] + S */
        const int32_t Pn = r->elen, L = lg->klen;
        if (Pn <= L) {
            put(o, r->ext, Pn, lg->key, L - Pn, lg->key + (L - Pn), Pn, f->ext, f->elen, 1, left, right);
        } else {
            uint8_t *c = (uint8_t *)malloc((size_t)(L + Pn));
            memcpy(c, r->ext, (size_t)Pn); memcpy(c + Pn, lg->key, (size_t)L);
            put(o, c, L, NULL, 0, c + L, Pn, f->ext, f->elen, 1, left, right);
            free(c);
        }
    }
}

static int related(const drec *a, const drec *b) {              /* subKmerSlotComparator || dynamicSubKmerComparator */
    const int32_t n = a->klen < b->klen ? a->klen : b->klen;
    return memcmp(a->key, b->key, (size_t)n) == 0;
}

/* One pass over rows sorted by key.  stage 0: DSExtendReflexivKmer (FirstFour); 1: DSExtendReflexivKmerToArrayLoop
 * (Iteration; start_iteration = param.startIteration selects the >= 61 rules, start_marker 1 when param.scramble == 3).
 * Returns the number of output rows; need_key / need_ext report the base capacities used or needed. */
int64_t orc_dyn_extend_pass(const uint8_t *key_bases, const int64_t *key_off, const int32_t *marker, const uint8_t *ext_bases,
                            const int64_t *ext_off, const int32_t *left, const int32_t *right, int64_t n, const int64_t *part_start, int P,
                            int stage, int start_iteration, int start_marker, uint8_t *o_key, int64_t cap_key, int64_t *o_key_off,
                            uint8_t *o_ext, int64_t cap_ext, int64_t *o_ext_off, int32_t *o_marker, int32_t *o_left, int32_t *o_right,
                            int64_t cap_n, int64_t *out_part_start, int64_t *need_key, int64_t *need_ext) {
    dout o = {o_key, o_key_off, o_ext, o_ext_off, o_marker, o_left, o_right, 0, cap_n, cap_key, cap_ext, 0, 0};
    for (int p = 0; p < P; p++) {
        if (out_part_start) out_part_start[p] = o.n;
        int m = start_marker;
        int have = 0;
        drec h; memset(&h, 0, sizeof h);
        for (int64_t q = part_start[p]; q < part_start[p + 1]; q++) {
            drec s;
            s.key = key_bases + key_off[q]; s.klen = (int32_t)(key_off[q + 1] - key_off[q]);
            s.ext = ext_bases + ext_off[q]; s.elen = (int32_t)(ext_off[q + 1] - ext_off[q]);
            s.marker = marker[q]; s.left = left[q]; s.right = right[q];
            if (!have) { h = s; have = 1; continue; }
            if (!related(&s, &h)) {                                /* a new group: the holder goes out, s takes its place */
                flip_emit(&o, &h, m); m = 3 - m;
                h = s;
                continue;
            }
            if (s.marker == h.marker) { flip_emit(&o, &s, m); m = 3 - m; continue; }
            const drec *f = s.marker == 1 ? &s : &h, *r = s.marker == 1 ? &h : &s;
            if (s.marker == 1) {
                const int32_t extra = h.klen < s.klen ? s.klen - h.klen : 0;
                if (s.klen < h.klen) {                             /* the forward row is the shorter one: no merge */
                    if (stage == 0 || start_iteration < 61) { flip_emit(&o, &s, m); m = 3 - m; }
                    continue;                                      /* (from iteration 61 on the row is dropped) */
                }
                int32_t d;
                int ok = 1;
                if (s.left < 0 && h.right < 0) d = -1;
                else if (s.left >= 0 && h.right >= 0) d = -1;
                else if (s.left >= 0 && s.left - h.elen >= 0) d = s.left - h.elen;
                else if (h.right >= 0 && h.right - s.elen - extra >= 0) d = h.right - s.elen;
                else { ok = 0; d = 0; }
                if (!ok) { flip_emit(&o, &s, m); m = 3 - m; continue; }
                merge_emit(&o, f, r, d, m); m = 3 - m; have = 0;
            } else {
                const int32_t extra = s.klen < h.klen ? h.klen - s.klen : 0;
                if (h.klen < s.klen) {                             /* the forward HOLDER is the shorter one */
                    if (stage == 1 && start_iteration >= 61) have = 0;      /* (from iteration 61 on the holder is dropped) */
                    flip_emit(&o, &s, m); m = 3 - m;
                    continue;
                }
                int32_t d;
                int ok = 1;
                if (s.right < 0 && h.left < 0) d = -1;
                else if (s.right >= 0 && h.left >= 0) d = -1;
                else if (s.right >= 0 && s.right - h.elen - extra >= 0) d = s.right - h.elen;
                else if (h.left >= 0 && h.left - s.elen >= 0) d = h.left - s.elen;
                else { ok = 0; d = 0; }
                if (!ok) { flip_emit(&o, &s, m); m = 3 - m; continue; }
                merge_emit(&o, f, r, d, m); m = 3 - m; have = 0;
            }
        }
        if (have) { flip_emit(&o, &h, m); m = 3 - m; }
    }
    if (out_part_start) out_part_start[P] = o.n;
    if (need_key) *need_key = o.need_key;
    if (need_ext) *need_ext = o.need_ext;
    return o.n;
}

/* DSkmerRandomReflection.call (FirstFour:2518-2524): every row of a partition in the toggling orientation, starting at 2 */
int64_t orc_dyn_random_reflection(const uint8_t *key_bases, const int64_t *key_off, const int32_t *marker, const uint8_t *ext_bases,
                                  const int64_t *ext_off, const int32_t *left, const int32_t *right, int64_t n, const int64_t *part_start,
                                  int P, uint8_t *o_key, int64_t cap_key, int64_t *o_key_off, uint8_t *o_ext, int64_t cap_ext,
                                  int64_t *o_ext_off, int32_t *o_marker, int32_t *o_left, int32_t *o_right, int64_t cap_n) {
    dout o = {o_key, o_key_off, o_ext, o_ext_off, o_marker, o_left, o_right, 0, cap_n, cap_key, cap_ext, 0, 0};
    for (int p = 0; p < P; p++) {
        int m = 2;
        for (int64_t q = part_start[p]; q < part_start[p + 1]; q++) {
            drec s;
            s.key = key_bases + key_off[q]; s.klen = (int32_t)(key_off[q + 1] - key_off[q]);
            s.ext = ext_bases + ext_off[q]; s.elen = (int32_t)(ext_off[q + 1] - ext_off[q]);
            s.marker = marker[q]; s.left = left[q]; s.right = right[q];
            flip_emit(&o, &s, m); m = 3 - m;
        }
    }
    (void)n;
    return o.n;
}

/* block form of a base string (the wire form a Row carries): writes blocks_of_len(n) longs */
int32_t orc_dyn_blocks(const uint8_t *s, int32_t n, int64_t *out) {
    const int32_t nb = blocks_of_len(n);
    for (int32_t j = 0; j < nb; j++) out[j] = block_of(s, n, j);
    return nb;
}
int32_t orc_dyn_attr_clamp(int32_t v) { return attr_clamp(v); }
