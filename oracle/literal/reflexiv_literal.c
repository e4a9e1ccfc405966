/*
 * reflexiv_literal.c -- the reference's OWN bit arithmetic for the flips and merges of the single-word
 * and first-array extend stages (k <= 31) and for the k > 31 counter's extraction (further down), written out
 * statement by statement from
 *   P/ReflexivDSMain.java:3153-3226 (DSExtendReflexivKmer.singleKmerRandomizer)
 *   P/ReflexivDSMain.java:3241-3325 (DSExtendReflexivKmer.reflexivExtend)
 *   P/ReflexivDSMain.java:2702-2810 (DSExtendReflexivKmerToArrayFirstTime.singleKmerRandomizer)
 *   P/ReflexivDSMain.java:2830-2967 (DSExtendReflexivKmerToArrayFirstTime.reflexivExtend)
 * with Java's integer semantics (64-bit two's complement, shift counts taken mod 64, >>> logical,
 * Long.numberOfLeadingZeros(0) = 64).  P = src/main/java/uni/bielefeld/cmg/reflexiv/pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Purpose: the oracle restates flips and
 * merges at SEQUENCE level (unpack, concatenate, repack); this file is the mechanical counterpart, so that
 * tests/test_oracle_literal.py can fuzz one against the other -- a pin of the sequence model to the reference's
 * bit code that does not go through anybody's reading of what the code "means".  The array-loop stage
 * (P/ReflexivDSMain.java:1894-2514, ~620 lines of word shuffling) is NOT transliterated here; SURVEY.md C.9
 * records the survey's own fuzz of it.
 */
#include <stdint.h>

typedef int64_t jlong;
static inline jlong jshl(jlong a, int n) { return (jlong)((uint64_t)a << (n & 63)); }
static inline jlong jushr(jlong a, int n) { return (jlong)((uint64_t)a >> (n & 63)); }
static inline int jnlz(jlong a) { return a ? __builtin_clzll((uint64_t)a) : 64; }

typedef struct {
    jlong key; int32_t marker; jlong ext[2]; int32_t n_ext; int32_t left, right;
} lit_rec;

/* maxSubKmerBinary = ~((~0L) << 2*param.subKmerSize)  (:3026);  maxBlockBinary = ~((~0L) << 2*31)  (:2575) */
static inline jlong max_sub(int sub) { return ~jshl(~(jlong)0, 2 * sub); }
static inline jlong max_block(void) { return ~jshl(~(jlong)0, 2 * 31); }

/* DSExtendReflexivKmer.singleKmerRandomizer :3153-3226; m = randomReflexivMarker on entry.
 * Returns 0, or -1 where the reference prints "what? not possible" (and emits zeros). */
int lit_flip_single(const lit_rec *cur, int m, int sub, lit_rec *out) {
    int bad = 0;
    if (cur->marker == 1) {                                                                  /* :3156 */
        int currentSuffixLength = 64 / 2 - (jnlz(cur->ext[0]) / 2 + 1);                      /* :3157 */
        jlong maxSuffixLengthBinary = ~jshl(~(jlong)0, 2 * currentSuffixLength);             /* :3158 */
        jlong newReflexivSubKmer = 0, newReflexivLong = 0;
        if (m == 2) {                                                                        /* :3162 */
            if (currentSuffixLength >= sub) bad = -1;                                        /* :3163-3168 */
            else {
                newReflexivSubKmer = jshl(cur->key, currentSuffixLength * 2);                /* :3170 */
                newReflexivSubKmer &= max_sub(sub);                                          /* :3171 */
                newReflexivSubKmer |= (cur->ext[0] & maxSuffixLengthBinary);                 /* :3172 */
                newReflexivLong = jushr(cur->key, 2 * (sub - currentSuffixLength));          /* :3175 */
                newReflexivLong |= jshl(1, 2 * currentSuffixLength);                         /* :3176 */
            }
            out->key = newReflexivSubKmer; out->marker = m; out->ext[0] = newReflexivLong; out->n_ext = 1;   /* :3181-3183 */
            out->left = cur->left; out->right = cur->right;
        } else { *out = *cur; out->n_ext = 1; }                                              /* :3185 */
    } else {                                                                                 /* :3187 */
        int currentPrefixLength = 64 / 2 - (jnlz(cur->ext[0]) / 2 + 1);                      /* :3188 */
        jlong maxSuffixLengthBinary = ~jshl(~(jlong)0, 2 * currentPrefixLength);             /* :3189 */
        jlong newReflexivSubKmer = 0, newReflexivLong = 0;
        if (m == 2) { *out = *cur; out->n_ext = 1; }                                         /* :3192-3193 */
        else {
            if (currentPrefixLength >= sub) bad = -1;                                        /* :3195-3200 */
            else {
                newReflexivSubKmer = jshl(cur->ext[0] & maxSuffixLengthBinary, 2 * (sub - currentPrefixLength));   /* :3202 */
                newReflexivSubKmer |= jushr(cur->key, 2 * currentPrefixLength);              /* :3204 */
                newReflexivLong = cur->key & maxSuffixLengthBinary;                          /* :3206 */
                newReflexivLong |= jshl(1, 2 * currentPrefixLength);                         /* :3207 */
            }
            out->key = newReflexivSubKmer; out->marker = m; out->ext[0] = newReflexivLong; out->n_ext = 1;   /* :3210-3212 */
            out->left = cur->left; out->right = cur->right;
        }
    }
    return bad;
}

/* left / right of a merged record: :3272-3288 (and :3308-3324, :2880-2900, :2946-2966) */
static void merged_ends(const lit_rec *f, const lit_rec *r, int bubbleDistance, lit_rec *out) {
    if (bubbleDistance < 0) { out->left = r->left; out->right = f->right; }
    else if (f->left > 0) { out->left = bubbleDistance; out->right = f->right; }
    else { out->left = r->left; out->right = bubbleDistance; }
}

/* DSExtendReflexivKmer.reflexivExtend :3241-3325 */
int lit_merge_single(const lit_rec *f, const lit_rec *r, int bubbleDistance, int m, int sub, lit_rec *out) {
    int bad = 0;
    int forwardSuffixLength = 64 / 2 - (jnlz(f->ext[0]) / 2 + 1);                            /* :3246 */
    int reflexedPrefixLength = 64 / 2 - (jnlz(r->ext[0]) / 2 + 1);                           /* :3247 */
    jlong maxSuffixLengthBinary = ~jshl(~(jlong)0, 2 * forwardSuffixLength);                 /* :3248 */
    jlong maxPrefixLengthBinary = ~jshl(~(jlong)0, 2 * reflexedPrefixLength);                /* :3249 */
    if (m == 2) {                                                                            /* :3252 */
        jlong newReflexivSubKmer = 0, newReflexivLong = 0;
        if (forwardSuffixLength >= sub) bad = -1;                                            /* :3256-3259 */
        else {
            newReflexivSubKmer = jshl(f->key, 2 * forwardSuffixLength);                      /* :3261 */
            newReflexivSubKmer &= max_sub(sub);                                              /* :3262 */
            newReflexivSubKmer |= (f->ext[0] & maxSuffixLengthBinary);                       /* :3263 */
            newReflexivLong = jshl(r->ext[0], 2 * forwardSuffixLength);                      /* :3265 */
            newReflexivLong |= jushr(f->key, 2 * (sub - forwardSuffixLength));               /* :3266 */
        }
        out->key = newReflexivSubKmer; out->marker = 2; out->ext[0] = newReflexivLong; out->n_ext = 1;
    } else {                                                                                 /* :3291 */
        jlong newForwardSubKmer = 0, newForwardLong = 0;
        if (reflexedPrefixLength >= sub) bad = -1;                                           /* :3295-3298 */
        else {
            newForwardSubKmer = jshl(r->ext[0] & maxPrefixLengthBinary, 2 * (sub - reflexedPrefixLength));   /* :3300 */
            newForwardSubKmer |= jushr(r->key, 2 * reflexedPrefixLength);                    /* :3301 */
            newForwardLong = r->key & maxPrefixLengthBinary;                                 /* :3303 */
            newForwardLong |= jshl(1, 2 * reflexedPrefixLength);                             /* :3304 */
            newForwardLong = jshl(newForwardLong, 2 * forwardSuffixLength);                  /* :3305 */
            newForwardLong |= (f->ext[0] & maxSuffixLengthBinary);                           /* :3306 */
        }
        out->key = newForwardSubKmer; out->marker = 1; out->ext[0] = newForwardLong; out->n_ext = 1;
    }
    merged_ends(f, r, bubbleDistance, out);
    return bad;
}

/* DSExtendReflexivKmerToArrayFirstTime.singleKmerRandomizer :2702-2810: the same flips, output as a one-element array */
int lit_flip_first(const lit_rec *cur, int m, int sub, lit_rec *out) {
    return lit_flip_single(cur, m, sub, out);        /* :2717-2727 = :3170-3176, :2771-2781 = :3202-3207 word for word */
}

/* DSExtendReflexivKmerToArrayFirstTime.reflexivExtend :2830-2967 */
int lit_merge_first(const lit_rec *f, const lit_rec *r, int bubbleDistance, int m, int sub, lit_rec *out) {
    int bad = 0;
    int forwardSuffixLength = 64 / 2 - (jnlz(f->ext[0]) / 2 + 1);                            /* :2835 */
    int reflexedPrefixLength = 64 / 2 - (jnlz(r->ext[0]) / 2 + 1);                           /* :2836 */
    int concatenateLength = forwardSuffixLength + reflexedPrefixLength;                      /* :2837 */
    jlong maxSuffixLengthBinary = ~jshl(~(jlong)0, 2 * forwardSuffixLength);                 /* :2838 */
    jlong maxPrefixLengthBinary = ~jshl(~(jlong)0, 2 * reflexedPrefixLength);                /* :2839 */
    out->ext[0] = out->ext[1] = 0; out->n_ext = 1;
    if (m == 2) {                                                                            /* :2842 */
        jlong newReflexivSubKmer = 0, newReflexivLong = 0;
        if (forwardSuffixLength >= sub) bad = -1;                                            /* :2847-2852 */
        else {
            newReflexivSubKmer = jshl(f->key, 2 * forwardSuffixLength);                      /* :2855 */
            newReflexivSubKmer &= max_sub(sub);                                              /* :2856 */
            newReflexivSubKmer |= (f->ext[0] & maxSuffixLengthBinary);                       /* :2857 */
            if (concatenateLength > 31) {                                                    /* :2858 */
                jlong newReflexivLonghead = jushr(r->ext[0], 2 * (31 - forwardSuffixLength));    /* :2859 */
                newReflexivLong = jshl(r->ext[0], 2 * forwardSuffixLength);                  /* :2860 */
                newReflexivLong &= max_block();                                              /* :2861 */
                newReflexivLong |= jushr(f->key, 2 * (sub - forwardSuffixLength));           /* :2862 */
                out->n_ext = concatenateLength / 31 + 1;                                     /* :2864 */
                out->ext[0] = newReflexivLonghead; out->ext[1] = newReflexivLong;            /* :2865-2866 */
            } else {
                newReflexivLong = jshl(r->ext[0], 2 * forwardSuffixLength);                  /* :2868 */
                newReflexivLong |= jushr(f->key, 2 * (sub - forwardSuffixLength));           /* :2869 */
                out->ext[0] = newReflexivLong;                                               /* :2872-2873 */
            }
        }
        out->key = newReflexivSubKmer; out->marker = 2;
    } else {                                                                                 /* :2902 */
        jlong newForwardSubKmer = 0, newForwardLong = 0;
        if (reflexedPrefixLength >= sub) bad = -1;                                           /* :2907-2912 */
        else {
            newForwardSubKmer = jshl(r->ext[0] & maxPrefixLengthBinary, 2 * (sub - reflexedPrefixLength));   /* :2914 */
            newForwardSubKmer |= jushr(r->key, 2 * reflexedPrefixLength);                    /* :2915 */
            if (concatenateLength > 31) {                                                    /* :2917 */
                jlong newForwardLonghead = f->key & maxPrefixLengthBinary;                   /* :2918 */
                newForwardLonghead = jushr(newForwardLonghead, 2 * (31 - forwardSuffixLength));  /* :2919 */
                newForwardLonghead |= jshl(1, 2 * (concatenateLength - 31));                 /* :2920 */
                newForwardLong = jshl(f->key, 2 * forwardSuffixLength);                      /* :2922 */
                newForwardLong |= (f->ext[0] & maxSuffixLengthBinary);                       /* :2923 */
                newForwardLong &= max_block();                                               /* :2924 */
                out->n_ext = concatenateLength / 31 + 1;                                     /* :2926 */
                out->ext[0] = newForwardLonghead; out->ext[1] = newForwardLong;              /* :2927-2928 */
            } else {
                newForwardLong = r->key & maxPrefixLengthBinary;                             /* :2930 */
                newForwardLong |= jshl(1, 2 * reflexedPrefixLength);                         /* :2931 */
                newForwardLong = jshl(newForwardLong, 2 * forwardSuffixLength);              /* :2932 */
                newForwardLong |= (f->ext[0] & maxSuffixLengthBinary);                       /* :2933 */
                out->ext[0] = newForwardLong;                                                /* :2936-2937 */
            }
        }
        out->key = newForwardSubKmer; out->marker = 1;
    }
    merged_ends(f, r, bubbleDistance, out);
    return bad;
}

/* ------------------------------------------------------------------------------------------------------------------
 * k > 31 counter: ReflexivDataFrameCounter64.ReverseComplementKmerBinaryExtractionFromDataset64, the rolling
 * multi-word forward / reverse-complement arrays and the choice between them, statement by statement from
 *   P/ReflexivDataFrameCounter64.java:391 (maxKmerBits), :403-647 (the read loop), :652-687 (compareLongArrayBlocks),
 *   :704-716 (nucleotideValue), U/DefaultParam.java:80-81 (kmerSizeResidue = k % 32, kmerBinarySlots = k / 32 + 1).
 * One read (ASCII, `len` characters) -> the canonical k-mers of its windows, `slots` words each, appended to out
 * (at most cap k-mers are written); returns the number of k-mers the read emits.  The oracle's sequence-level
 * orc_extract_canon_w is fuzzed against this in tests/test_oracle_literal.py.
 */
static jlong lit_nucleotide_value(int a) {                   /* :704-716 */
    if (a == 'A') return 0;
    if (a == 'C') return 1;
    if (a == 'G') return 2;
    return 3;
}

static int lit_compare_blocks(const jlong *forward, const jlong *reverse, int slots, int residue) {   /* :652-687 */
    for (int i = 0; i < slots; i++) {
        if (i < slots - 1) {
            for (int j = 0; j < 32; j++) {
                jlong s1 = jushr(forward[i], 2 * (31 - j)) & 3;
                jlong s2 = jushr(reverse[i], 2 * (31 - j)) & 3;
                if (s1 < s2) return 1;
                else if (s1 > s2) return 0;
            }
        } else {
            for (int j = 0; j < residue; j++) {
                jlong s1 = jushr(forward[i], 2 * (residue - 1 - j)) & 3;
                jlong s2 = jushr(reverse[i], 2 * (residue - 1 - j)) & 3;
                if (s1 < s2) return 1;
                else if (s1 > s2) return 0;
            }
        }
    }
    return 1;                                                 /* "should not happen": a palindrome keeps the forward strand */
}

#define LIT_MAX_SLOTS 8
int64_t lit_counter64_extract(const uint8_t *read, int len, int k, int frontClip, int endClip, uint64_t *out, int64_t cap) {
    const int residue = k % 32, slots = k / 32 + 1;           /* DefaultParam :80-81 */
    if (slots > LIT_MAX_SLOTS || slots < 2) return -1;
    const jlong maxKmerBits = ~jshl(~(jlong)0, 2 * residue);   /* :391 */
    int64_t emitted = 0;
    if (len - k - endClip + 1 <= 0 || frontClip > len) return 0;           /* :410-412 */
    jlong nb = 0, nbrc = 0;
    jlong fs[LIT_MAX_SLOTS] = {0}, rs[LIT_MAX_SLOTS] = {0};
    for (int i = frontClip; i < len - endClip; i++) {                      /* :419 */
        const jlong v = lit_nucleotide_value(read[i]);
        /* forward k-mer in bits  :424-463 */
        if (i - frontClip <= k - 1) {
            nb = jshl(nb, 2);
            nb |= v;
            if ((i - frontClip + 1) % 32 == 0) {
                fs[(i - frontClip + 1) / 32 - 1] = nb;
                nb = 0;
            }
            if (i - frontClip == k - 1) {
                nb &= maxKmerBits;
                fs[(i - frontClip + 1) / 32] = nb;
                nb = 0;
            }
        } else {
            jlong t1 = jushr(fs[slots - 1], 2 * (residue - 1));
            jlong t2;
            fs[slots - 1] = jshl(fs[slots - 1], 2);
            fs[slots - 1] |= v;
            fs[slots - 1] &= maxKmerBits;
            for (int j = slots - 2; j >= 0; j--) {
                t2 = jushr(fs[j], 2 * 31);
                fs[j] = jshl(fs[j], 2);
                fs[j] |= t1;
                t1 = t2;
            }
        }
        /* reverse complement  :466-515 */
        jlong c = v ^ 3;
        if (i - frontClip <= k - 1) {
            if (i - frontClip < residue - 1) {
                c = jshl(c, 2 * (i - frontClip));
                nbrc |= c;
            } else if (i - frontClip == residue - 1) {
                c = jshl(c, 2 * (i - frontClip));
                nbrc |= c;
                rs[slots - 1] = nbrc;
                nbrc = 0;
            } else if ((i - frontClip - residue + 1) % 32 == 0) {
                c = jshl(c, 2 * ((i - frontClip - residue) % 32));
                nbrc |= c;
                rs[slots - ((i - frontClip - residue + 1) / 32) - 1] = nbrc;
                nbrc = 0;
            } else {
                c = jshl(c, 2 * ((i - frontClip - residue) % 32));
                nbrc |= c;
            }
        } else {
            jlong t1 = jshl(rs[0], 2 * 31);
            jlong t2;
            rs[0] = jushr(rs[0], 2);
            c = jshl(c, 2 * 31);
            rs[0] |= c;
            for (int j = 1; j < slots - 1; j++) {
                t2 = jshl(rs[j], 2 * 31);
                rs[j] = jushr(rs[j], 2);
                rs[j] |= t1;
                t1 = t2;
            }
            rs[slots - 1] = jushr(rs[slots - 1], 2);
            t1 = jushr(t1, 2 * (31 - residue + 1));
            rs[slots - 1] |= t1;
        }
        /* the first complete k-mer and every one after it  :526, :626-645 */
        if (i - frontClip >= k - 1) {
            const jlong *pick = lit_compare_blocks(fs, rs, slots, residue) ? fs : rs;
            if (emitted < cap) for (int j = 0; j < slots; j++) out[emitted * slots + j] = (uint64_t)pick[j];
            emitted++;
        }
    }
    return emitted;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Fork filters of the DS twin (min_error_cov off), P/ReflexivDSMain.java:3369-3417 (DSFilterForkSubKmer) and
 * :3486-3538 (DSFilterForkReflectedSubKmer): one partition of rows (key, marker, ext, left, right) in, the kept rows
 * out (n_out returned).  Row fields as the reference's getLong(0), getInt(1), getLong(2), getInt(3), getInt(4).
 */
typedef struct { jlong key; int32_t marker; jlong ext; int32_t left, right; } lit_row;

int64_t lit_fork_forward(const lit_row *in, int64_t n, int subKmerSize, lit_row *out) {      /* :3375-3416 */
    int64_t m = 0;
    for (int64_t q = 0; q < n; q++) {
        lit_row s = in[q];
        if (m == 0) {
            out[m++] = (lit_row){s.key, s.marker, s.ext, s.left, -1};
        } else {
            if (s.key == out[m - 1].key) {
                if (s.left > out[m - 1].left) {
                    out[m - 1] = (lit_row){s.key, s.marker, s.ext, s.left, subKmerSize};
                } else if (s.left == out[m - 1].left) {
                    if (s.ext > out[m - 1].ext) {
                        out[m - 1] = (lit_row){s.key, s.marker, s.ext, s.left, subKmerSize};
                    } else {
                        s = out[m - 1];
                        out[m - 1] = (lit_row){s.key, s.marker, s.ext, s.left, subKmerSize};
                    }
                } else {
                    s = out[m - 1];
                    out[m - 1] = (lit_row){s.key, s.marker, s.ext, s.left, subKmerSize};
                }
            } else {
                out[m++] = (lit_row){s.key, s.marker, s.ext, s.left, -1};
            }
        }
    }
    return m;
}

int64_t lit_fork_reflected(const lit_row *in, int64_t n, int subKmerSize, lit_row *out) {    /* :3493-3537 */
    int64_t m = 0;
    int32_t lastCoverage = 0;
    for (int64_t q = 0; q < n; q++) {
        lit_row s = in[q];
        if (m == 0) {
            lastCoverage = s.left;
            out[m++] = (lit_row){s.key, s.marker, s.ext, -1, s.right};
        } else {
            if (s.key == out[m - 1].key) {
                if (s.left > lastCoverage) {
                    lastCoverage = s.left;
                    out[m - 1] = (lit_row){s.key, s.marker, s.ext, subKmerSize, s.right};
                } else if (s.left == lastCoverage) {
                    int l1 = 64 / 2 - (jnlz(s.ext) / 2 + 1);
                    int l2 = 64 / 2 - (jnlz(out[m - 1].ext) / 2 + 1);
                    jlong f1 = jushr(s.ext, 2 * (l1 - 1));
                    jlong f2 = jushr(out[m - 1].ext, 2 * (l2));          /* (as written: the kept row's sentinel, not its first base) */
                    if (f1 > f2) {
                        out[m - 1] = (lit_row){s.key, s.marker, s.ext, subKmerSize, s.right};
                    } else {
                        s = out[m - 1];
                        out[m - 1] = (lit_row){s.key, s.marker, s.ext, subKmerSize, s.right};
                    }
                } else {
                    s = out[m - 1];
                    out[m - 1] = (lit_row){s.key, s.marker, s.ext, subKmerSize, s.right};
                }
            } else {
                lastCoverage = s.left;
                out[m++] = (lit_row){s.key, s.marker, s.ext, -1, s.right};
            }
        }
    }
    return m;
}
