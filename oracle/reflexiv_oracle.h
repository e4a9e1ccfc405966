/*
 * reflexiv_oracle.h -- CPU restatement of Reflexiv's fixed-k assembly hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under reflexiv_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / the reported CPU baseline.
 *
 * Pinning: the reference's tests hold no golden vectors for this path
 * (src/test/.../ReflexivMainTest.java:36-45 asserts 1 == 1), and no JVM exists
 * in the build container, so the reference itself cannot run.  The oracle is
 * pinned by the single known answer the reference documents
 * (docs/example.html:303,320-343: one 4558-base contig per strand, first 1200
 * bases printed) -- see tests/test_oracle_example.py.  Everything else is
 * "parity unpinned by reference tests" and anchored on the Java source cited
 * per function below.  P = src/main/java/uni/bielefeld/cmg/reflexiv/pipeline.
 *
 * All record arrays are flat struct-of-arrays in the reference's own record
 * layout:
 *   key    (k-1)-mer, 2 bits/base, first base in the highest used bit pair
 *   marker 1 = forward  (sequence = key || ext), 2 = reflected (ext || key)
 *   ext    words [ext_off[i], ext_off[i+1]): word 0 = first f (1..31) bases with
 *          a 1-bit sentinel at bit 2f, every further word exactly 31 bases
 *   left,right  bubble-distance markers (< 0: free end)
 */
#ifndef REFLEXIV_ORACLE_H
#define REFLEXIV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TWIN_DS  0   /* P/ReflexivDSMain.java arithmetic (normative)   */
#define ORC_TWIN_RDD 1   /* P/ReflexivMain.java arithmetic (operator twin)  */

typedef struct {
    int32_t k;                /* kmerSize                 DefaultParam.java:74  */
    int32_t min_cov;          /* minKmerCoverage          DefaultParam.java:103 */
    int32_t max_cov;          /* maxKmerCoverage          DefaultParam.java:104 */
    int32_t min_error_cov;    /* minErrorCoverage         DefaultParam.java:105 */
    int32_t min_contig;       /* minContig                DefaultParam.java:107 */
    int32_t min_iter;         /* minimumIteration         DefaultParam.java:115 */
    int32_t max_iter;         /* maximumIteration         DefaultParam.java:114 */
    int32_t front_clip;       /* frontClip                DefaultParam.java:119 */
    int32_t end_clip;         /* endClip                  DefaultParam.java:120 */
    int32_t partitions;       /* logical partitions P of the order contract    */
    int32_t twin;             /* ORC_TWIN_DS / ORC_TWIN_RDD                     */
    int32_t coalesce;         /* apply the partition coalesce rule (:277-281)   */
    int32_t extras;           /* k > 31 only: the from-counts extras of P/ReflexivDSMain64.java:584-619, 672-712
                               * (orientation doubling, extendable / unextendable split, end filters); default 1 */
} orc_params;

void orc_default_params(orc_params *p);

/* a-1  FastqFilterWithQual  P/ReflexivMain.java:3089-3113.  Walks '\n'-separated
 * lines of text[0..len) with the reference's 4-line state machine and records the
 * (offset,length) of the sequence line of every emitted record.  Returns the
 * number of records (may exceed cap; only the first cap are written). */
int64_t orc_fastq_group(const char *text, int64_t len,
                        int64_t *seq_off, int32_t *seq_len, int64_t cap);

/* DSFastqFilterOnlySeq.call  P/ReflexivDataFrameCounter.java:238-290 (the counter's line filter,
 * same in ReflexivDataFrameCounter64.java:236-288): keeps every line longer than 20 characters
 * that starts with neither '@' nor '+' and has one of A T C G N at positions 0, 4, 9, 14, 19
 * (so a quality line that happens to pass is kept too).  Same output convention as above. */
int64_t orc_fastq_only_seq(const char *text, int64_t len,
                           int64_t *seq_off, int32_t *seq_len, int64_t cap);

/* a-2  ReverseComplementKmerBinaryExtraction.call  P/ReflexivMain.java:3013-3075.
 * Reads are ASCII, read i = bases[read_off[i] .. read_off[i+1]).  Returns the
 * number of canonical k-mers (written to out if it fits in cap). */
int64_t orc_extract_canon(const char *bases, const int64_t *read_off, int64_t n_reads,
                          int k, int front_clip, int end_clip,
                          uint64_t *out, int64_t cap);

/* a-3/a-4  reduceByKey(KmerCounting) + KmerCoverageFilter
 * P/ReflexivMain.java:155,160-163,2895-2899,3115-3119.  Sorts kmers in place,
 * emits (key,count) with min<=count<=max in ascending key order.  Returns the
 * number of survivors; *n_distinct (optional) = distinct keys before filter. */
int64_t orc_count_filter(uint64_t *kmers, int64_t n, int min_cov, int max_cov, int twin,
                         uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                         int64_t *n_distinct);

/* a-5/a-6  KmerReverseComplement + ForwardSubKmerExtraction
 * P/ReflexivMain.java:2910-2930, 2709-2730.  n (kmer,count) -> 2n records
 * (key = kmer>>>2, marker 1, ext = kmer&3 [no sentinel], left = right = count). */
void orc_rc_expand_subkmer(const uint64_t *kmers, const int32_t *counts, int64_t n, int k,
                           uint64_t *key, int32_t *marker, uint64_t *ext,
                           int32_t *left, int32_t *right);

uint64_t orc_revcomp(uint64_t kmer, int k);

/* Order contract B.0: stable sort by key; writes the permutation (perm[i] =
 * source index of the record at sorted position i). */
void orc_sort_perm(const uint64_t *key, int64_t n, int64_t *perm);

/* Order contract B.0: logical partition starts after a sort: start[p] =
 * floor(p*n/P) moved forward so that equal keys never split; start[P] = n. */
void orc_partition_starts(const uint64_t *sorted_key, int64_t n, int P, int64_t *start);

/* a-7  FilterForkSubKmer[WithErrorCorrection]  P/ReflexivMain.java:2412-2540,
 * DS P/ReflexivDSMain.java:3375-3483.  Single-word records sorted by key.
 * part_start[P+1] in, out_part_start[P+1] out.  Returns survivors. */
int64_t orc_fork_filter_forward(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                const int32_t *left, const int32_t *right, int64_t n,
                                const int64_t *part_start, int P,
                                int k, int min_error_cov, int twin,
                                uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                int32_t *oleft, int32_t *oright, int64_t *out_part_start);

/* a-8  ReflectedSubKmerExtractionFromForward  P/ReflexivMain.java:2742-2768 */
void orc_reflect_from_forward(const uint64_t *key, const uint64_t *ext, int64_t n, int k,
                              uint64_t *okey, int32_t *omarker, uint64_t *oext);

/* a-9  FilterForkReflectedSubKmer[WithErrorCorrection]  P/ReflexivMain.java:2550-2696,
 * DS P/ReflexivDSMain.java:3493-3616 */
int64_t orc_fork_filter_reflected(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                  const int32_t *left, const int32_t *right, int64_t n,
                                  const int64_t *part_start, int P,
                                  int k, int min_error_cov, int twin,
                                  uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                  int32_t *oleft, int32_t *oright, int64_t *out_part_start);

/* a-10 kmerRandomReflection  P/ReflexivMain.java:2783-2885: record j of a partition
 * takes orientation 2 if j is even, 1 if odd (in place). */
void orc_random_reflection(uint64_t *key, int32_t *marker, uint64_t *ext, int64_t n,
                           const int64_t *part_start, int P, int k);

/* a-11..a-13  ExtendReflexivKmer / ...ToArrayFirstTime / ...ToArrayLoop
 * P/ReflexivMain.java:2019-2401, 1564-2013, 762-1558 (DS: 3011-3367, 2559-3009,
 * 1746-2557).  One extend pass over records already sorted by key.  Outputs must
 * have room for n records and ext_off[n] words.  Returns the output record count;
 * out_part_start[P+1] receives the per-partition output offsets. */
int64_t orc_extend_pass(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                        const uint64_t *ext, const int32_t *left, const int32_t *right,
                        int64_t n, const int64_t *part_start, int P, int k, int twin,
                        uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                        int32_t *oleft, int32_t *oright, int64_t *out_part_start);

/* gather records (with variable-length ext) by permutation */
void orc_gather(const int64_t *perm, int64_t n,
                const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                const uint64_t *ext, const int32_t *left, const int32_t *right,
                uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                int32_t *oleft, int32_t *oright);

/* a-15  BinaryReflexivKmerArrayToString + KmerToContig + TagContigID
 * P/ReflexivMain.java:696-741, 590-637, 573-581 (DS 855-900, 743-795, 717-725).
 * Writes the text saveAsTextFile would write (every element followed by '\n').
 * Returns the text length (written only if it fits in cap). */
int64_t orc_contigs_text(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                         const uint64_t *ext, const int32_t *left, const int32_t *right,
                         int64_t n, int k, int min_contig, int twin,
                         char *out, int64_t cap, int64_t *n_contigs);

/* a-14 whole driver (P/ReflexivMain.java:147-310 / P/ReflexivDSMain.java:204-352)
 * from sorted filtered (kmer,count) to contig text.  trace (optional, cap entries)
 * receives the record count after every extend pass; *n_trace the passes run.
 * If rec_out != NULL it is filled with a malloc'd copy of the final records
 * (free with orc_free_records). */
typedef struct {
    int64_t n;
    uint64_t *key; int32_t *marker; int64_t *ext_off; uint64_t *ext;
    int32_t *left; int32_t *right;
} orc_records;

int64_t orc_assemble_from_counts(const uint64_t *kmers, const int32_t *counts, int64_t n,
                                 const orc_params *prm,
                                 char *out, int64_t cap, int64_t *n_contigs,
                                 int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                                 orc_records *rec_out);
void orc_free_records(orc_records *r);

/* ---- a-2w: k > 31 (the counter's multi-word k-mers), P/ReflexivDataFrameCounter64.java
 * W = k/32+1 words per k-mer: words 0..W-2 hold 32 bases each (all 64 bits), the last word the
 * k%32 remaining bases right-aligned (:429-437).  k % 32 != 0, k > 32.  AoS: k-mer i = out[i*W ..]. */
int orc_words_w(int k);                                   /* k/32+1 (U/DefaultParam.java:81) */
/* ReverseComplementKmerBinaryExtractionFromDataset64.call :401-650, compareLongArrayBlocks :652-687 */
int64_t orc_extract_canon_w(const char *bases, const int64_t *read_off, int64_t n_reads,
                            int k, int front_clip, int end_clip, uint64_t *out, int64_t cap);
/* groupBy("kmerBlocks").count() + the two filters (:197-209).  Output ascending by base string
 * (the order contract's count-stage order).  kmers (n*W words) is sorted in place. */
int64_t orc_count_filter_w(uint64_t *kmers, int64_t n, int k, int min_cov, int max_cov,
                           uint64_t *out_keys, int64_t *out_counts, int64_t cap, int64_t *n_distinct);
/* DSBinaryKmerToString.call :340-369: the k characters of one k-mer (no terminator). */
void orc_kmer_text_w(const uint64_t *kmer, int k, char *out);


/* ---- k > 31: the assembler's twin  P/ReflexivDSMain64.java:374-826 (assemblyFromKmer).
 * Keys are (k-1)-mers in orc_sub_words(k) = (k-2)/31+1 words of 31 bases, the last word holding the
 * remaining (k-2)%31+1 bases right-aligned (U/DefaultParam.java:93-94); key arrays are AoS, kw
 * consecutive words per record, as the reference's Row(long[]).  k-mers entering the path hold k
 * bases in orc_asm_words(k) = (k-1)/31+1 such words (KmerBinarizer :10812-10819) -- NOT the counter's
 * 32-bases-per-word layout; the two meet through CSV text only (orc_counter_to_asm_w restates that
 * round trip).  Extensions, markers and left/right are as for k <= 31.  The _w operators accept any
 * k >= 3 (kw = 1 up to k = 32) and are what the k <= 31 entry points above call with kw = 1.
 * Flips and merges are restated at sequence level; SURVEY.md C.9 records the one case where the
 * reference's bit code differs (first-array stage, forward output, 16 + 16 bases: P/ReflexivDSMain64.java
 * :9377-9379 leaves 15 junk bases above the length marker) -- the oracle keeps the sequence model.
 * Parity pin: NONE held by the reference for k > 31 (the documented example is k = 31): "parity
 * unpinned" beyond the cross-check against tests/pymodel.py and the k = 31 agreement of the shared code. */
int orc_sub_words(int k);
int orc_asm_words(int k);
/* KmerBinarizer.call :10772-10836, one CSV row */
int orc_kmer_binarize_w(const char *kmer_text, const char *count_text, int k, uint64_t *words, int32_t *cover);
/* counter layout (k/32+1 words of 32 bases) -> assembler layout, as the CSV round trip does */
void orc_counter_to_asm_w(const uint64_t *kmers32, int64_t n, int k, uint64_t *kmers31);
/* DSKmerReverseComplement :10706-10755 + DSForwardSubKmerExtraction :10363-10403 */
void orc_rc_expand_subkmer_w(const uint64_t *kmers, const int32_t *counts, int64_t n, int k,
                             uint64_t *key, int32_t *marker, uint64_t *ext,
                             int32_t *left, int32_t *right);
void orc_sort_perm_w(const uint64_t *key, int64_t n, int kw, int64_t *perm);
void orc_partition_starts_w(const uint64_t *sorted_key, int64_t n, int kw, int P, int64_t *start);
/* DSFilterForkSubKmer[WithErrorCorrection] :10072-10208 */
int64_t orc_fork_filter_forward_w(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                  const int32_t *left, const int32_t *right, int64_t n,
                                  const int64_t *part_start, int P,
                                  int k, int min_error_cov, int twin,
                                  uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                  int32_t *oleft, int32_t *oright, int64_t *out_part_start);
/* DSReflectedSubKmerExtractionFromForward :10426-10475 */
void orc_reflect_from_forward_w(const uint64_t *key, const uint64_t *ext, int64_t n, int k,
                                uint64_t *okey, int32_t *omarker, uint64_t *oext);
/* DSFilterForkReflectedSubKmer[WithErrorCorrection] :10210-10360 */
int64_t orc_fork_filter_reflected_w(const uint64_t *key, const int32_t *marker, const uint64_t *ext,
                                    const int32_t *left, const int32_t *right, int64_t n,
                                    const int64_t *part_start, int P,
                                    int k, int min_error_cov, int twin,
                                    uint64_t *okey, int32_t *omarker, uint64_t *oext,
                                    int32_t *oleft, int32_t *oright, int64_t *out_part_start);
/* DSkmerRandomReflection :10491-10690 */
void orc_random_reflection_w(uint64_t *key, int32_t *marker, uint64_t *ext, int64_t n,
                             const int64_t *part_start, int P, int k);
/* DSExtendReflexivKmer :9465-10070, ...ToArrayFirstTime :8733-9463, ...ToArrayLoop :7446-8731.
 * start_marker: the value randomReflexivMarker has when a task starts (2; 1 once param.scramble == 3,
 * :7484-7486). */
int64_t orc_extend_pass_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                          const uint64_t *ext, const int32_t *left, const int32_t *right,
                          int64_t n, const int64_t *part_start, int P, int k, int twin, int start_marker,
                          uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                          int32_t *oleft, int32_t *oright, int64_t *out_part_start);
void orc_gather_w(const int64_t *perm, int64_t n, int kw,
                  const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                  const uint64_t *ext, const int32_t *left, const int32_t *right,
                  uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext,
                  int32_t *oleft, int32_t *oright);
/* DSBinaryReflexivKmerArrayToString :1913-1975 + DSKmerToContig :842-892 + TagRowContigID :830-841 */
int64_t orc_contigs_text_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off,
                           const uint64_t *ext, const int32_t *left, const int32_t *right,
                           int64_t n, int k, int min_contig,
                           char *out, int64_t cap, int64_t *n_contigs);
/* The from-counts extras, one function per operator class of P/ReflexivDSMain64.java, on records
 * sorted by key with their partition starts (outputs need room for 2n records / 2 * words where noted):
 *   orc_double_w            DSReflexivAndForwardKmer :2126-3042          every record, then its other orientation (2n out)
 *   orc_extendable_pairs_w  DSFilterExtendableKmerPairs :5305-6375       both members of every mergeable (forward,
 *                           reflected) pair on one key, both as forward records; the task's last holder too
 *   orc_unextendable_w      DSFilterUnExtendableKmer :6377-7444          what is left when the mergeable pairs are dropped
 *   orc_first_of_key_w      DSFilterStillExtendableKmerFromPairs :3228-3390   one record per key (the first)
 *   orc_longer_of_key_w     DSFilterStillExtendableKmerEnds :3044-3226   of two records on one key the one with the
 *                           larger (length * 31 + first word length)
 *   orc_flip_all_w          DSFilterUnExtendableKmerLeftEnds :3392-4347 (m = 1) / ...RightEnds :4349-5303 (m = 2) */
int64_t orc_double_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                     const int32_t *left, const int32_t *right, int64_t n, int k,
                     uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft, int32_t *oright);
int64_t orc_key_filter_w(int op, const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                         const int32_t *left, const int32_t *right, int64_t n, const int64_t *part_start, int P, int k,
                         uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft,
                         int32_t *oright, int64_t *out_part_start);
#define ORC_OP_EXTENDABLE_PAIRS 1
#define ORC_OP_UNEXTENDABLE     2
#define ORC_OP_FIRST_OF_KEY     3
#define ORC_OP_LONGER_OF_KEY    4
int64_t orc_flip_all_w(const uint64_t *key, const int32_t *marker, const int64_t *ext_off, const uint64_t *ext,
                       const int32_t *left, const int32_t *right, int64_t n, int k, int m,
                       uint64_t *okey, int32_t *omarker, int64_t *oext_off, uint64_t *oext, int32_t *oleft, int32_t *oright);

/* assemblyFromKmer :374-826 from the filtered (k-mer, count) list in ascending order, k-mers in the assembler
 * layout; prm->extras selects whether :584-619 / :672-712 run (see the .c file). */
int64_t orc_assemble_from_counts_w(const uint64_t *kmers, const int32_t *counts, int64_t n,
                                   const orc_params *prm,
                                   char *out, int64_t cap, int64_t *n_contigs,
                                   int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                                   orc_records *rec_out);

/* ---- CPU baseline over all host cores (bench.py cpu_baseline; SURVEY.md 8d "OpenMP, one logical partition
 * per thread ~ local[N]").  orc_set_threads(t > 1) makes sorts and gathers split their index range and the
 * partition-wise operators (fork filters, random reflection, extend pass) run one logical partition per task,
 * each task being the serial function on that partition alone; results are identical to t = 1
 * (tests/test_oracle_omp.py).  The parity tests leave it at 1. */
void orc_set_threads(int t);
int orc_get_threads(void);
int orc_host_cores(void);
/* a-2 + a-3 + a-4 fused over the threads set above: map side = per-thread slices of the reads into range
 * buckets of the k-mer space, reduce side = per-bucket sort + count + filter.  Same output as
 * orc_extract_canon + orc_count_filter (ascending).  Returns the survivors (written if <= cap). */
int64_t orc_count_reads_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                            int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                            uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                            int64_t *n_distinct, int64_t *n_instances);

/* The same restricted to the range buckets [b_lo, b_hi) of 4096 (top 12 bits of the k-mer): several passes over
 * disjoint shares of the k-mer space count a read set whose instances do not fit in memory at once; outputs
 * concatenate to the full ascending list.  And its k = 33..63 twin (two-word k-mers, int64 counts, the filters of
 * P/ReflexivDataFrameCounter64.java:197-205; buckets = top 12 bits of word 0). */
int64_t orc_count_reads_range_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                                  int k, int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                                  int b_lo, int b_hi,
                                  uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                                  int64_t *n_distinct, int64_t *n_instances);
int64_t orc_count_reads_w2_range_omp(const char *bases, const int64_t *read_off, int64_t n_reads,
                                     int k, int front_clip, int end_clip, int min_cov, int max_cov,
                                     int b_lo, int b_hi,
                                     uint64_t *out_keys, int64_t *out_counts, int64_t cap,
                                     int64_t *n_distinct, int64_t *n_instances);

/* Synthetic reads, integer-only counter-based generator shared
 * bit-for-bit with reflexiv_amd/csrc (rfx_synth_*). */
uint64_t orc_splitmix64(uint64_t x);
void orc_synth_genome(uint64_t seed, int64_t genome_len, uint64_t *packed /* ceil(len/32) words */);
/* read r (0..n_reads): ASCII into bases[r*read_len ...] */
void orc_synth_reads(uint64_t seed, const uint64_t *genome, int64_t genome_len,
                     int64_t first_read, int64_t n_reads, int read_len, uint32_t err_per_2_32,
                     char *bases);

#ifdef __cplusplus
}
#endif
#endif
