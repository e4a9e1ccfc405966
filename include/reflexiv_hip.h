/*
 * reflexiv_hip.h -- C ABI of libreflexiv_hip.so, the MI355X (gfx950) implementation of
 * Reflexiv's k-mer counting + reflexible extend-and-merge hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI seam of its own:
 * the path sits behind Spark's Java functional interfaces, one inner class per operator,
 * called once per partition.  Every "operator" entry point below replaces the body of one
 * of those classes (cited as P/<file>:<lines>, P = src/main/java/uni/bielefeld/cmg/
 * reflexiv/pipeline) and takes / returns the same records as flat arrays; INTEGRATION.md
 * shows the JNI stub that binds each of them.  Conventions:
 *
 *  - plain pointers and sizes only; the caller owns every buffer (Java: direct
 *    ByteBuffers or arrays pinned with GetPrimitiveArrayCritical);
 *  - host entry points (rfx_*) take HOST pointers, stage through HBM and run the HIP
 *    kernels; device entry points (rfx_dev_*) take DEVICE pointers and run on the
 *    context's stream, for callers that keep the data resident (the Python/torch
 *    harness, the C++ driver, the multi-GPU path);
 *  - return value: RFX_OK or a negative rfx_status; never aborts, never falls back to a
 *    CPU path.  RFX_E_CAP means an output buffer was too small: *out_n (and *out_words)
 *    hold the needed size, nothing else was written;
 *  - re-entrant per context; one context = one device + one HIP stream.
 *
 * Record layout = the reference's: key is the (k-1)-mer, 2 bits per
 * base, first base in the highest used pair; marker 1 = forward (sequence = key||ext),
 * 2 = reflected (ext||key); ext words [ext_off[i], ext_off[i+1]): word 0 holds the first
 * f (1..31) bases under a 1-bit sentinel at bit 2f, every further word exactly 31 bases;
 * left/right are the bubble-distance markers (< 0: free end).
 */
#ifndef REFLEXIV_HIP_H
#define REFLEXIV_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RFX_OK        =  0,
    RFX_E_ARG     = -1,   /* bad argument (k out of range, null pointer, ...)              */
    RFX_E_CAP     = -2,   /* output capacity too small; needed size reported              */
    RFX_E_HIP     = -3,   /* a HIP runtime call failed; see rfx_last_error()               */
    RFX_E_NOGPU   = -4,   /* no usable gfx950 device                                       */
    RFX_E_STATE   = -5,   /* "impossible" record state (the reference prints and goes on)  */
    RFX_E_LIMIT   = -6,   /* size beyond this build's limits (e.g. > 2^32-1 records)       */
    RFX_E_HOST    = -7    /* a C++ exception (std::bad_alloc, std::system_error ...) was caught at this boundary; what() in
                           * rfx_last_error().  Every entry point is a function-try-block: nothing unwinds into the caller  */
} rfx_status;

#define RFX_TWIN_DS  0    /* arithmetic of P/ReflexivDSMain.java (the wired, fixed twin)   */
#define RFX_TWIN_RDD 1    /* arithmetic of P/ReflexivMain.java (the RDD-surface twin)      */

typedef struct rfx_ctx rfx_ctx;

/* U/DefaultParam.java:74-120 -- the fields that reach the hot path. */
typedef struct {
    int32_t k;               /* kmerSize (<= 31: rfx_dev_assemble; 32..125: rfx_dev_assemble_w)  :74  */
    int32_t min_cov;         /* minKmerCoverage                                :103 */
    int32_t max_cov;         /* maxKmerCoverage                                :104 */
    int32_t min_error_cov;   /* minErrorCoverage (4 * 2)                       :105 */
    int32_t min_contig;      /* minContig                                      :107 */
    int32_t min_iter;        /* minimumIteration                               :115 */
    int32_t max_iter;        /* maximumIteration                               :114 */
    int32_t front_clip;      /* frontClip                                      :119 */
    int32_t end_clip;        /* endClip                                        :120 */
    int32_t partitions;      /* logical partitions P of the order contract */
    int32_t twin;            /* RFX_TWIN_DS / RFX_TWIN_RDD                      */
    int32_t coalesce;        /* apply P/ReflexivMain.java:277-281               */
    int32_t extras;          /* k > 31 only (rfx_dev_assemble_w): the from-counts extras of P/ReflexivDSMain64.java:584-619,
                              * 672-712 -- orientation doubling, extendable / unextendable split, end filters.  Default 1
                              * (the reference's driver always runs them); 0 iterates every record to the end. */
} rfx_params;

/* Flat record set.  n and the pointers are filled by the caller on input; on output the
 * callee sets n (and ext_off[n] words of ext).  cap_n / cap_words are the capacities of
 * the caller's output buffers (ext_off needs cap_n + 1 entries); on RFX_E_CAP nothing is
 * written except need_n / need_words. */
typedef struct {
    int64_t   n;
    uint64_t *key;
    int32_t  *marker;
    int64_t  *ext_off;
    uint64_t *ext;
    int32_t  *left;
    int32_t  *right;
    int64_t   cap_n;
    int64_t   cap_words;
    int64_t   need_n;      /* set by the callee: records / words the output needs */
    int64_t   need_words;
    int32_t   key_words;   /* words per key: 0 or 1 for k <= 32; (k-2)/31+1 for k > 32, key then holds n * key_words
                            * words, record i at key[i * key_words ..] (31 bases per word, the last word the remaining
                            * (k-2)%31+1 bases right-aligned: P/ReflexivDSMain64.java, U/DefaultParam.java:93-94).
                            * The callee sets it on output records. */
    int32_t   reserved_;
} rfx_records;

/* ------------------------------------------------------------------ context */

int  rfx_version(void);
void rfx_default_params(rfx_params *p);                 /* U/DefaultParam.java defaults   */
/* device < 0: current device.  Fails with RFX_E_NOGPU when there is no gfx950 GPU.       */
int  rfx_ctx_create(int device, rfx_ctx **out);
void rfx_ctx_destroy(rfx_ctx *ctx);
int  rfx_ctx_sync(rfx_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own. */
int  rfx_ctx_set_stream(rfx_ctx *ctx, void *hip_stream);
void *rfx_ctx_stream(rfx_ctx *ctx);
const char *rfx_last_error(rfx_ctx *ctx);               /* text of the last RFX_E_HIP / RFX_E_HOST / RFX_E_STATE */
/* The context keeps its large buffers between calls (grow-only workspaces: the count stage's record buffers ~ 3 B per
 * k-mer instance, the extend stage's two arenas, and -- rfx_assemble_reads -- the packed reads plus, up to
 * RFX_KEEP_STAGING_BYTES = 8 GiB, the ASCII staging).  rfx_ctx_workspace_bytes reports what is held; rfx_ctx_trim waits
 * for the context's stream and hands everything back to the driver (the next call allocates again). */
int  rfx_ctx_trim(rfx_ctx *ctx);
int64_t rfx_ctx_workspace_bytes(rfx_ctx *ctx);

/* ------------------------------------------------- operators, host buffers */

/* ReverseComplementKmerBinaryExtraction.call  P/ReflexivMain.java:3013-3075
 * (DS: P/ReflexivDSMain.java:3961-4024, P/ReflexivDataFrameCounter.java:459-526).
 * ASCII reads, read i = bases[read_off[i] .. read_off[i+1]); emits the canonical k-mers
 * in read order, window order. */
int rfx_extract_canon(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off,
                      int64_t n_reads, int k, int front_clip, int end_clip,
                      uint64_t *out_kmers, int64_t cap, int64_t *out_n);

/* reduceByKey(KmerCounting) + filter(KmerCoverageFilter)
 * P/ReflexivMain.java:155,160-163,2895-2899,3115-3119 (DS :207-216).
 * Output ascending by k-mer (the order contract's count-stage order). */
int rfx_count_filter(rfx_ctx *ctx, const uint64_t *kmers, int64_t n,
                     int min_cov, int max_cov, int twin,
                     uint64_t *out_keys, int32_t *out_counts, int64_t cap,
                     int64_t *out_n, int64_t *out_distinct);

/* ---- k > 31: the counter's multi-word k-mers ----
 * W = k/32+1 words per k-mer, words 0..W-2 hold 32 bases each, the last word the k%32 remaining
 * bases right-aligned (P/ReflexivDataFrameCounter64.java:429-437); k > 32, k % 32 != 0, W <= 8.
 * Arrays hold W consecutive words per k-mer, as the reference's Row(long[]) does. */

/* ReverseComplementKmerBinaryExtractionFromDataset64.call
 * P/ReflexivDataFrameCounter64.java:401-650 (canonical by compareLongArrayBlocks :652-687:
 * base-wise fwd < rc, ties -> fwd).  Skip rule :410.  cap/out_n count k-mers, not words. */
int rfx_extract_canon_w(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off,
                        int64_t n_reads, int k, int front_clip, int end_clip,
                        uint64_t *out_kmers, int64_t cap, int64_t *out_n);

/* groupBy("kmerBlocks").count() + filter(count >= min if min > 1) + filter(count <= max if
 * max < 10000000)  P/ReflexivDataFrameCounter64.java:191-205.  Output ascending by base string. */
int rfx_count_filter_w(rfx_ctx *ctx, const uint64_t *kmers, int64_t n, int k,
                       int min_cov, int max_cov,
                       uint64_t *out_keys, int64_t *out_counts, int64_t cap,
                       int64_t *out_n, int64_t *out_distinct);

/* KmerReverseComplement.call + ForwardSubKmerExtraction.call
 * P/ReflexivMain.java:2910-2930, 2709-2730 (DS :3849-3868, :3625-3644).
 * n (kmer,count) -> 2n single-word records; ext carries no sentinel yet. */
int rfx_rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *kmers, const int32_t *counts,
                          int64_t n, int k, rfx_records *out);

/* sortByKey() / sort("k-1")  P/ReflexivMain.java:179,191,211,235,247,286: global stable
 * sort by key; also returns the logical partition starts part_start[P+1]. */
int rfx_sort_records(rfx_ctx *ctx, const rfx_records *in, int P,
                     rfx_records *out, int64_t *part_start);

/* FilterForkSubKmer[WithErrorCorrection].call  P/ReflexivMain.java:2412-2540
 * (DS :3375-3483).  Input sorted by key with its partition starts. */
int rfx_fork_filter_forward(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start,
                            int P, int k, int min_error_cov, int twin,
                            rfx_records *out, int64_t *out_part_start);

/* ReflectedSubKmerExtractionFromForward.call  P/ReflexivMain.java:2742-2768 (DS :3661-3685) */
int rfx_reflect_from_forward(rfx_ctx *ctx, const rfx_records *in, int k, rfx_records *out);

/* FilterForkReflectedSubKmer[WithErrorCorrection].call  P/ReflexivMain.java:2550-2696
 * (DS :3493-3616) */
int rfx_fork_filter_reflected(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start,
                              int P, int k, int min_error_cov, int twin,
                              rfx_records *out, int64_t *out_part_start);

/* kmerRandomReflection.call  P/ReflexivMain.java:2783-2885 (DS :3697-3805) */
int rfx_random_reflection(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start,
                          int P, int k, rfx_records *out);

/* ExtendReflexivKmer / ExtendReflexivKmerToArrayFirstTime / ExtendReflexivKmerToArrayLoop
 * .call  P/ReflexivMain.java:2048-2362, 1594-1974, 792-1519 (DS :3040-3329, :2589-2971,
 * :1776-2518).  One extend pass over records sorted by key.  stage: 0 single word,
 * 1 first array pass, 2 array loop -- the three classes share one algorithm and one
 * word layout; stage 0 additionally checks that every output fits one word. */
int rfx_extend_pass(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P,
                    int k, int twin, int stage, rfx_records *out, int64_t *out_part_start);

/* ---- k > 31: the assembler's twin P/ReflexivDSMain64.java (assemblyFromKmer :374-826) ----
 * Every record operator above also takes k = 32 .. 125: keys are then (k-1)-mers of key_words = (k-2)/31+1 words
 * (rfx_records.key_words; 31 bases per word), the class bodies replaced are DSKmerReverseComplement +
 * DSForwardSubKmerExtraction (:10706-10755, :10363-10403; rfx_rc_expand_subkmer: k-mers of (k-1)/31+1 words each in
 * the layout KmerBinarizer writes, :10812-10819 -- NOT the counter's 32-bases-per-word layout), sort("k-1"),
 * DSFilterForkSubKmer[WithErrorCorrection] (:10072-10208), DSReflectedSubKmerExtractionFromForward (:10426-10475),
 * DSFilterForkReflectedSubKmer[WithErrorCorrection] (:10210-10360), DSkmerRandomReflection (:10491-10690),
 * DSExtendReflexivKmer / ...ToArrayFirstTime / ...ToArrayLoop (:9465-10070, :8733-9463, :7446-8731; twin is ignored,
 * this class has the DS arithmetic only) and DSBinaryReflexivKmerArrayToString + DSKmerToContig + TagRowContigID
 * (:1913-1975, :842-892, :830-841; header ">Contig-<len>-<idx>").  Flips and merges follow the sequence model
 * (SURVEY.md B.7); the one case where the reference's bit code differs is recorded in SURVEY.md C.9. */

/* DSExtendReflexivKmerToArrayLoop.call with param.scramble (2 or 3) as the class reads it (:7484-7486: the task's
 * emission marker starts at 1 instead of 2 once scramble == 3).  Any k the record operators take. */
int rfx_extend_pass_w(rfx_ctx *ctx, const rfx_records *in, const int64_t *part_start, int P, int k, int stage,
                      int scramble, rfx_records *out, int64_t *out_part_start);

/* The operator classes of the k > 31 from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712), one `op` each, on
 * records sorted by key with their partition starts (any k the record operators take):
 *   RFX_OP_DOUBLE            DSReflexivAndForwardKmer :2126-3042: every record, then its other orientation (out: 2n
 *                            records, 2 x words; out_part_start = 2 x part_start)
 *   RFX_OP_EXTENDABLE_PAIRS  DSFilterExtendableKmerPairs :5305-6375: both members of every (forward, reflected) pair on
 *                            one key that the extend pass would merge, both as forward records; a task's last holder too
 *   RFX_OP_UNEXTENDABLE      DSFilterUnExtendableKmer :6377-7444: the rest (reflected holders come out forward)
 *   RFX_OP_FIRST_OF_KEY      DSFilterStillExtendableKmerFromPairs :3228-3390: of two records on one key the first
 *   RFX_OP_LONGER_OF_KEY     DSFilterStillExtendableKmerEnds :3044-3226: ... the one with the larger length*31+first word
 *   RFX_OP_ALL_FORWARD / RFX_OP_ALL_REFLECTED   DSFilterUnExtendableKmerLeftEnds :3392-4347 / ...RightEnds :4349-5303 */
#define RFX_OP_DOUBLE           0
#define RFX_OP_EXTENDABLE_PAIRS 1
#define RFX_OP_UNEXTENDABLE     2
#define RFX_OP_FIRST_OF_KEY     3
#define RFX_OP_LONGER_OF_KEY    4
#define RFX_OP_ALL_FORWARD      5
#define RFX_OP_ALL_REFLECTED    6
int rfx_extras_operator(rfx_ctx *ctx, int op, const rfx_records *in, const int64_t *part_start, int P, int k,
                        rfx_records *out, int64_t *out_part_start);

/* BinaryReflexivKmerArrayToString + KmerToContig + TagContigID
 * P/ReflexivMain.java:696-741, 590-637, 573-581 (DS :855-900, :743-795, :717-725):
 * the text saveAsTextFile writes (host-side formatting). */
int rfx_contigs_text(rfx_ctx *ctx, const rfx_records *in, int k, int min_contig, int twin,
                     char *out, int64_t cap, int64_t *out_len, int64_t *out_contigs);

/* ------------------------------------------- resident pipeline, device buffers */

/* 2-bit read store: read i occupies words [i*words_per_read, (i+1)*words_per_read), base j
 * in word j/32 at bits 63-2*(j%32)..62-2*(j%32) (code A0 C1 G2, anything else 3:
 * P/ReflexivMain.java:3062-3074), read_len[i] bases.  All pointers are device pointers. */
int rfx_dev_encode_reads(rfx_ctx *ctx, const uint8_t *d_bases, const int64_t *d_read_off,
                         int64_t n_reads, int words_per_read,
                         uint64_t *d_words, uint32_t *d_read_len);

/* Number of k-mer instances the extraction emits for uniform reads (host arithmetic). */
int64_t rfx_kmers_per_read(int read_len, int k, int front_clip, int end_clip);

/* Workspace bytes rfx_dev_count_reads needs for n_kmers instances. */
int64_t rfx_count_workspace_bytes(int64_t n_kmers);

/* extract + reduceByKey + filter fused, reads of one uniform length already packed in HBM.
 * d_out_keys/d_out_counts (cap entries) receive the survivors ascending by k-mer.
 * d_workspace: rfx_count_workspace_bytes(n_reads * kmers_per_read) bytes.
 * shard_lo/shard_hi select the radix shard [lo, hi) of 2^16 of the k-mer hash space that
 * this call counts (0, 65536 = everything); used by the multi-GPU path. */
int rfx_dev_count_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads,
                        int words_per_read, int read_len, int k, int front_clip, int end_clip,
                        int min_cov, int max_cov, int twin,
                        void *d_workspace, int64_t workspace_bytes,
                        uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                        int64_t *out_n, int64_t *out_distinct, int64_t *out_instances);

/* k > 31 twin of rfx_dev_count_reads (all device pointers; d_out_keys: cap*W words). */
int64_t rfx_kmers_per_read_w(int read_len, int k, int front_clip, int end_clip);
int rfx_dev_count_reads_w(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads,
                          int words_per_read, int read_len, int k, int front_clip, int end_clip,
                          int min_cov, int max_cov,
                          uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap,
                          int64_t *out_n, int64_t *out_distinct, int64_t *out_instances);

/* The same for reads of different lengths (real FASTQ): d_read_len[i] bases in read i, as
 * rfx_dev_encode_reads writes them; max_read_len <= 32 * words_per_read bounds them. */
int rfx_dev_count_reads_ragged(rfx_ctx *ctx, const uint64_t *d_words, const uint32_t *d_read_len,
                               int64_t n_reads, int words_per_read, int max_read_len, int k,
                               int front_clip, int end_clip, int min_cov, int max_cov, int twin,
                               uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                               int64_t *out_n, int64_t *out_distinct, int64_t *out_instances);

/* Multi-GPU exchange support for k = 33..63: the canonical two-word k-mers of packed reads (16-byte
 * elements {word0, word1}) written contiguously per owner (owner = mulhi(hash(k-mer), n_owners)),
 * d_owner_off[n_owners+1] element offsets; and the count of such elements after the exchange
 * (any order) -> ascending (2 words per key, int64 counts), as rfx_dev_count_reads_w returns them. */
int rfx_dev_bucket_wide_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads,
                                 int words_per_read, int read_len, int k, int front_clip, int end_clip,
                                 int n_owners, void *d_out_elems, int64_t cap_elems,
                                 int64_t *d_owner_off, int64_t *h_owner_off);
int rfx_dev_count_wide_elems(rfx_ctx *ctx, const void *d_elems, int64_t n_elems, int k,
                             int min_cov, int max_cov,
                             uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap,
                             int64_t *out_n, int64_t *out_distinct);

/* The same two steps on 32-byte super-k-mer RECORDS of two-word k-mers (k = 33..63): a run of <= 16
 * consecutive windows that share the minimiser of their CENTRAL 31 (k odd) / 30 (k even) bases, carried
 * as the run's k + windows - 1 bases + a header; ~5 B per instance across the exchange instead of 16.
 * Bucketing: d_out_records = NULL / cap_records = 0 reports *out_n_records (RFX_E_CAP), as for k <= 31. */
int rfx_dev_bucket_wide_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read,
                                         int read_len, int k, int front_clip, int end_clip, int n_owners,
                                         void *d_out_records, int64_t cap_records, int64_t *d_owner_off,
                                         int64_t *h_owner_off, int64_t *out_n_records);
int rfx_dev_count_wide_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint, int k,
                               int min_cov, int max_cov, uint64_t *d_out_keys, int64_t *d_out_counts, int64_t cap,
                               int64_t *out_n, int64_t *out_distinct);

/* Same, from an explicit k-mer array (the reduceByKey input) in HBM. */
int rfx_dev_count_kmers(rfx_ctx *ctx, const uint64_t *d_kmers, int64_t n,
                        int min_cov, int max_cov, int twin,
                        void *d_workspace, int64_t workspace_bytes,
                        uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap,
                        int64_t *out_n, int64_t *out_distinct);

/* Multi-GPU exchange support: bucket the canonical k-mers of packed reads by owner
 * (owner = hash(kmer) * n_owners >> 64, the radix shard of the k-mer space), writing each
 * owner's k-mers contiguously into d_out with d_owner_off[n_owners+1] element offsets. */
int rfx_dev_bucket_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads,
                            int words_per_read, int read_len, int k, int front_clip,
                            int end_clip, int n_owners, uint64_t *d_out, int64_t cap,
                            int64_t *d_owner_off, int64_t *h_owner_off);

/* The same two steps on super-k-mer RECORDS (16 bytes per run of <= 16 consecutive windows that
 * share a minimiser; ~2.6 B per instance instead of 8), for k = 21..31.  Bucketing: call with
 * d_out = NULL / cap_records = 0 to learn *out_n_records (returns RFX_E_CAP), then with a buffer.
 * Counting: n_instances_hint (k-mer instances the records hold, 0 = unknown) sizes the radix plan. */
int rfx_dev_bucket_records_by_owner(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read,
                                    int read_len, int k, int front_clip, int end_clip, int n_owners,
                                    void *d_out_records, int64_t cap_records, int64_t *d_owner_off,
                                    int64_t *h_owner_off, int64_t *out_n_records);
int rfx_dev_count_records(rfx_ctx *ctx, const void *d_records, int64_t n_records, int64_t n_instances_hint,
                          int k, int min_cov, int max_cov, int twin, uint64_t *d_out_keys,
                          int32_t *d_out_counts, int64_t cap, int64_t *out_n, int64_t *out_distinct);

/* The same exchange AFTER a local combine -- the map-side combine of `reduceByKey`
 * (P/ReflexivMain.java:155; Spark sums per key inside every map task before the shuffle): a rank
 * counts its own reads first and ships (k-mer, partial count) PAIRS, 16 bytes {k-mer, count}, one per
 * distinct k-mer of the rank instead of one unit per instance (k <= 31).
 *   rfx_dev_combine_reads         reads -> every distinct canonical k-mer with its local count, no filter,
 *                                 grouped by owner = mulhi(kmer_hash(k-mer), n_owners) (bucket o =
 *                                 d_out_pairs[owner_off[o] .. owner_off[o+1]), *out_n pairs in all).  Both
 *                                 d_scratch_pairs and d_out_pairs hold cap_pairs pairs; RFX_E_CAP with
 *                                 *out_n = the capacity needed when that is short
 *   rfx_dev_bucket_pairs_by_owner the grouping step alone (pairs with count 0 are dropped)
 *   rfx_dev_merge_pairs           the owner's half: sum the partial counts per k-mer, then
 *                                 KmerCoverageFilter (P/ReflexivMain.java:3115-3119), ascending order */
int rfx_dev_combine_reads(rfx_ctx *ctx, const uint64_t *d_words, int64_t n_reads, int words_per_read, int read_len,
                          int k, int front_clip, int end_clip, int n_owners, void *d_scratch_pairs, void *d_out_pairs,
                          int64_t cap_pairs, int64_t *d_owner_off, int64_t *h_owner_off, int64_t *out_n,
                          int64_t *out_instances);
int rfx_dev_bucket_pairs_by_owner(rfx_ctx *ctx, const void *d_pairs, int64_t n_pairs, int n_owners, void *d_out_pairs,
                                  int64_t *d_owner_off, int64_t *h_owner_off);
int rfx_dev_merge_pairs(rfx_ctx *ctx, const void *d_pairs, int64_t n_pairs, int k, int min_cov, int max_cov, int twin,
                        uint64_t *d_out_keys, int32_t *d_out_counts, int64_t cap, int64_t *out_n,
                        int64_t *out_distinct);

/* ---- the count stage on several GPUs of one node: the shuffle of `reduceByKey` (P/ReflexivMain.java:155, map-side
 * combine included) / `groupBy("value").count()` (P/ReflexivDSMain.java:207-209) as an all-to-all(v) over RCCL / xGMI.
 * One process or thread per GPU, each with its own rfx_ctx and one rfx_comm.  The k-mer space is radix-sharded by the
 * owner of each k-mer's minimiser; what crosses the links are super-k-mer records (16 B per run of windows for
 * k = 21..31, 32 B for k = 33..63: 2.6 / 5.4 bytes per k-mer instance).  RCCL is bound at run time (dlopen): the host
 * process's own RCCL when it has one, /opt/rocm/lib/librccl.so.1 otherwise; RFX_E_NOGPU when there is none.
 *   rfx_comm_unique_id      rank 0 makes the 128-byte id; the host hands it to every rank (Spark: a broadcast variable)
 *   rfx_comm_init           collective: every rank calls it with the same id, its rank and the world size (<= 64)
 *   rfx_comm_all_reduce_i64 sum (op 0) / max (op 1) of up to 8 host int64 over the ranks, in place (count() of the
 *                           stop rule, totals, a barrier)
 *   rfx_dev_sharded_count   collective: this rank's packed reads in HBM (d_read_len: per-read lengths for ragged reads,
 *                           k = 21..31, or NULL = every read has read_len bases) -> its shard of the filtered (k-mer, count) list,
 *                           ascending (d_out_counts: int32 for k <= 31, int64 beyond, as the fused calls; k / 32 + 1 words per
 *                           key).  k = 21..31 and 33..63 exchange super-k-mer records in generations; every other k of the
 *                           counters (3..20, 65..125; not a multiple of 32) exchanges its k-mer instances in one go;
 *                           out_totals[3] = instances, distinct, survivors over ALL ranks.  `generations` (1..8) cuts
 *                           the hash space so that generation g is counted while g+1.. travel.  RFX_E_CAP -- on
 *                           EVERY rank when the shard of ANY rank did not fit -- with *out_n = what this rank needs.
 *   rfx_dev_gather_shards   collective: every rank's shard to rank `root`, shard after shard in rank order
 * Environment: RFX_COMM_LIMIT_BYTES (per peer and call, default 512 MiB), RFX_COMM_SELF_VIA_RCCL=1 (tests: the rank's own
 * bucket through ncclSend / ncclRecv instead of a device copy). */
typedef struct rfx_comm rfx_comm;
int rfx_comm_unique_id(uint8_t *id128);
int rfx_comm_init(rfx_ctx *ctx, const uint8_t *id128, int rank, int world, rfx_comm **out);
void rfx_comm_destroy(rfx_comm *comm);
int rfx_comm_rank(const rfx_comm *comm);
int rfx_comm_world(const rfx_comm *comm);
int64_t rfx_comm_last_bytes_bucketed(const rfx_comm *comm);
int rfx_comm_all_reduce_i64(rfx_comm *comm, int64_t *h_vals, int n, int op);
int rfx_dev_sharded_count(rfx_ctx *ctx, rfx_comm *comm, const uint64_t *d_words, const uint32_t *d_read_len, int64_t n_reads,
                          int words_per_read, int read_len, int k, int front_clip, int end_clip, int generations, int min_cov,
                          int max_cov, int twin, uint64_t *d_out_keys, void *d_out_counts, int64_t cap, int64_t *out_n,
                          int64_t *out_totals);
/* The extend stage on several GPUs -- every sortByKey of the reference's driver (P/ReflexivMain.java:179,191,211,235,247,286;
 * k > 31: P/ReflexivDSMain64.java:504-563) as a range shuffle of whole records over RCCL: local stable sort, rank splitters
 * from ONE all-gather of a device-built sample, one count matrix, one all-to-all(v), local stable sort; the order contract's
 * logical partitions are cut out of the GLOBAL sorted sequence (any prm->partitions with any number of ranks: a partition may
 * go on on the next rank, the parity it brings along travels in one small all-gather); count() of the stop
 * rule and the trace are one all-reduce per pass.  d_keys / d_counts: this rank's shard of the filtered (k-mer, count) list
 * in HBM, any order (k <= 31: one word per k-mer, as rfx_dev_assemble; k = 32..124: (k-1)/31+1 words of 31 bases, as
 * rfx_dev_assemble_w).  The record set stays sharded while it has more than `gather_below` records over all ranks
 * (< 0: RFX_SHARD_GATHER_BELOW or 32 Mi; 0: to the end of the loop); then -- and before the k > 31 from-counts extras -- it is
 * gathered on rank 0, where the one-GPU driver takes the loop up.  The contig text (identical to rfx_dev_assemble[_w] on one
 * GPU for the same prm->partitions), *out_contigs and the trace arrive on rank 0 (*out_len = 0 elsewhere).  Collective: a
 * failure of any rank fails the call on every rank (RFX_E_STATE on the others), and RFX_E_CAP -- the text buffer of rank 0
 * is too short -- comes back on EVERY rank with *out_len = the length needed, so that retries re-enter together. */
int rfx_dev_sharded_assemble(rfx_ctx *ctx, rfx_comm *comm, const uint64_t *d_keys, const int32_t *d_counts, int64_t n,
                             const rfx_params *prm, int64_t gather_below, char *out, int64_t cap, int64_t *out_len,
                             int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace);
/* The whole resident path on several GPUs from ASCII reads in host memory (the multi-GPU rfx_assemble_reads; what one
 * Spark executor per GPU calls with ITS partition of the reads): upload + 2-bit encode (any read lengths), the sharded
 * count above, then rfx_dev_sharded_assemble (gather_below as there: a bacterial genome's survivors go to rank 0 at once,
 * a record set that does not fit one GPU stays sharded).  The contig text arrives on rank 0 (*out_len = 0 on the others;
 * RFX_E_CAP on every rank, see there).  k = 21..31.  Collective. */
int rfx_sharded_assemble_reads(rfx_ctx *ctx, rfx_comm *comm, const uint8_t *bases, const int64_t *read_off, int64_t n_reads,
                               const rfx_params *prm, int generations, int64_t gather_below, char *out, int64_t cap,
                               int64_t *out_len, int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                               int64_t *out_totals);
int rfx_dev_gather_shards(rfx_ctx *ctx, rfx_comm *comm, const uint64_t *d_keys, const void *d_counts, int64_t n,
                          int key_words, int count_bytes, int root, uint64_t *d_out_keys, void *d_out_counts, int64_t cap,
                          int64_t *out_n);

/* ---- the record operators on sets that STAY in HBM (the multi-GPU extend stage: reflexiv_amd/dist.py puts the RCCL
 * all-to-all of whole records between them).  Same operators, same reference classes as the host
 * entry points above; here every pointer inside rfx_records, every part_start and the k-mer / count arrays are DEVICE
 * pointers.  Inputs: in->n records, in->need_words = ext_off[n] (the value the producing call reported), key_words
 * as above.  Outputs: caller-allocated arrays of cap_n records / cap_words words (ext_off: cap_n + 1); on return
 * n / need_n / need_words are set (host fields); RFX_E_CAP when a capacity is short.  All run on the context's stream
 * and return after it has drained. */
int rfx_dev_rc_expand_subkmer(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n, int k,
                              rfx_records *d_out);
int rfx_dev_sort_records(rfx_ctx *ctx, const rfx_records *d_in, int P, int k, rfx_records *d_out, int64_t *d_part_start);
int rfx_dev_fork_filter(rfx_ctx *ctx, int reflected, const rfx_records *d_in, const int64_t *d_part_start, int P, int k,
                        int min_error_cov, int twin, rfx_records *d_out, int64_t *d_out_part_start);
int rfx_dev_reflect_from_forward(rfx_ctx *ctx, const rfx_records *d_in, int k, rfx_records *d_out);
int rfx_dev_random_reflection(rfx_ctx *ctx, const rfx_records *d_in, const int64_t *d_part_start, int P, int k,
                              rfx_records *d_out);
int rfx_dev_extend_pass(rfx_ctx *ctx, const rfx_records *d_in, const int64_t *d_part_start, int P, int k, int twin,
                        int stage, int scramble, rfx_records *d_out, int64_t *d_out_part_start);
/* positions of m values in n ascending one-word keys: d_out[j] = number of keys < values[j] (upper = 0) or <= (upper = 1)
 * -- the splitter search of the range shuffle that stands in for sortByKey's range partitioner */
int rfx_dev_lower_bound(rfx_ctx *ctx, const uint64_t *d_sorted_keys, int64_t n, const uint64_t *d_values, int64_t m,
                        int upper, int64_t *d_out);

/* Whole driver  P/ReflexivMain.java:168-310 (DS :221-352) from the filtered, ascending
 * (kmer,count) list in HBM to the contig text in host memory.  trace (optional) receives
 * the record count after every extend pass. */
int rfx_dev_assemble(rfx_ctx *ctx, const uint64_t *d_keys, const int32_t *d_counts, int64_t n,
                     const rfx_params *prm, char *out, int64_t cap, int64_t *out_len,
                     int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace);

/* k = 33..63: put (two-word k-mer, count) pairs that are in any order (e.g. the shards of the hash-partitioned count
 * gathered from several GPUs) into ascending k-mer order in place -- the order rfx_dev_count_reads_w returns and the
 * from-counts driver expects (order contract: ascending by base string). */
int rfx_dev_order_kmers_w(rfx_ctx *ctx, uint64_t *d_keys, int64_t *d_counts, int64_t n, int k);

/* KmerBinarizer.call + the count filter of the from-counts driver (P/ReflexivDSMain64.java:10772-10836, :473-478) for
 * k-mers that never left HBM: the counter's output (k/32+1 words of 32 bases per k-mer, int64 counts, ascending) ->
 * the assembler's input ((k-1)/31+1 words of 31 bases, int32 counts read as the CSV text would be: 10 digits or more
 * = 1000000000), keeping min_cov <= count <= max_cov.  Output buffers hold n entries.  All device pointers. */
int rfx_dev_counter_to_asm(rfx_ctx *ctx, const uint64_t *d_keys32, const int64_t *d_counts64, int64_t n, int k,
                           int min_cov, int max_cov, uint64_t *d_out_kmers, int32_t *d_out_counts, int64_t *out_n);

/* Whole k > 31 driver ReflexivDSMain64.assemblyFromKmer (P/ReflexivDSMain64.java:458-826) from the filtered, ascending
 * (k-mer, count) list in HBM (assembler layout) to the contig text.  prm->extras = 1 (default): at iteration
 * minimumIteration + 3 the records are doubled into both orientations and split into the members of mergeable pairs
 * and the rest (:584-619), only the first set is iterated further (:621-661), and after the loop the two are united and
 * records that share an end with a longer one are dropped (:672-712).  prm->extras = 0: the loop iterates all records.
 * Stop rule: checked from minimumIteration + 3 on, the first repeat of the count sets param.scramble = 3 (later passes
 * start their emission marker at 1), the second stops. */
int rfx_dev_assemble_w(rfx_ctx *ctx, const uint64_t *d_kmers, const int32_t *d_counts, int64_t n,
                       const rfx_params *prm, char *out, int64_t cap, int64_t *out_len,
                       int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace);

/* The same driver from HOST arrays (what `run -kmerc COUNTS -kmer 63` holds after KmerBinarizer and the count filter,
 * P/ReflexivDSMain64.java:458-478): n k-mers of (k-1)/31+1 words each, ascending, with their int32 counts. */
int rfx_assemble_counts_w(rfx_ctx *ctx, const uint64_t *kmers, const int32_t *counts, int64_t n,
                          const rfx_params *prm, char *out, int64_t cap, int64_t *out_len,
                          int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace);

/* The whole resident path from ASCII reads in host memory (any lengths) to the contig text:
 * upload, 2-bit encode, extract + count + filter (prm->min_cov .. max_cov), the driver above --
 * nothing but the reads goes up and nothing but the text comes back.  k <= 31.
 * out_kept (optional) = number of k-mers that passed the coverage filter. */
int rfx_assemble_reads(rfx_ctx *ctx, const uint8_t *bases, const int64_t *read_off, int64_t n_reads,
                       const rfx_params *prm, char *out, int64_t cap, int64_t *out_len,
                       int64_t *out_contigs, int64_t *trace, int64_t trace_cap, int64_t *n_trace,
                       int64_t *out_kept);

/* Contig RC de-duplication: P/ReflexivDSDynamicKmerDedup.java `assemblyFromKmer` (:138-339) -- three rounds of
 * marker 31-mers (ReverseComplementKmerMarkerExtraction :2674-3096, ForwardAndReverseComplementKmerMarkerExtraction :2206-2673)
 * -> sort -> DSMarkerKmerSelection (:1788-1870) -> groupBy().count() >= 2 -> DSMarkerKmerShorterID (:3186-3207) ->
 * DSShorterRCContigSeqAndTargetExtraction (:3097-3132) -> DSShorterRCContigRemoval (:1405-1558) /
 * DSShorterForwardAndRCContigRemoval[Array] (:508-729, :959-1175), then TagRowContigDSID (:3397-3443).  The fixed-k path
 * emits every contig on both strands; this reports each once (and merges overlapping pieces as the reference does).
 *   rfx_dedup_contigs      contigs as ASCII bases + offsets (ids = positions, as zipWithIndex numbers them) -> the survivors
 *                          (ASCII + offsets; RFX_E_CAP when a capacity is short) and / or the text: ">Contig-<len>-<idx>" +
 *                          the sequence in lines of 10,000,000, contigs of at least min_contig bases; round_n[3] (optional)
 *                          = contigs left after each round
 *   rfx_dedup_contig_text  the same from the contig text the path writes (either twin's headers, 100-column lines) */
int rfx_dedup_contigs(rfx_ctx *ctx, const uint8_t *bases_ascii, const int64_t *contig_off, int64_t n_contigs, int min_contig,
                      uint8_t *out_bases_ascii, int64_t cap_bases, int64_t *out_off, int64_t cap_contigs, int64_t *out_n,
                      char *text, int64_t text_cap, int64_t *text_len, int64_t *round_n);
int rfx_dedup_contig_text(rfx_ctx *ctx, const char *contig_text, int64_t len, int min_contig, char *out, int64_t cap,
                          int64_t *out_len, int64_t *out_contigs, int64_t *round_n);

/* The dynamic-k record format and passes: P/ReflexivDSDynamicKmerFirstFour.java (DSkmerRandomReflection
 * :2509-2762, DSExtendReflexivKmer :1581-2373) and P/ReflexivDSDynamicKmerIteration.java (DSExtendReflexivKmerToArrayLoop
 * :465-1249) -- the "meta" assembler's passes on keys of ANY length (a (k-1)-mer of whichever k the reduction kept).  In
 * a Row a key is left-aligned 31-base blocks with a 01 terminator, the attribute one long (marker << 62 | left << 32 |
 * right, negatives as 30000 - v), the extension in the same left-aligned form; across this boundary a record set is base
 * CODES (A0 C1 G2 T3, one byte per base) with offsets + marker / left / right (left / right as the reference reads them
 * back: clamped to +-30000).  Two keys meet when they are equal OR one is a prefix of the other; the forward record must
 * not be the shorter one; the merged key keeps the longer key's length.
 *   rfx_dyn_sort               sort("k-1") as Spark orders array<long> (element by element as signed longs, a proper prefix
 *                              first), stable, + the cut into P logical partitions (floor(p*n/P) moved past equal keys)
 *   rfx_dyn_random_reflection  DSkmerRandomReflection.call on the given partitions
 *   rfx_dyn_extend_pass        one pass over sorted records: stage 0 DSExtendReflexivKmer (extension in one long),
 *                              1 DSExtendReflexivKmerToArrayLoop (start_iteration = param.startIteration: from 61 on a
 *                              shorter forward record is dropped); start_marker 2, or 1 when param.scramble == 3
 *   rfx_dyn_run                the drivers with the records resident in HBM between the operators:
 *                              FirstFour.assemblyFromKmer (:137-224) = random_reflection 1, passes_first_four 4, no
 *                              iterations (end < start); Iteration.assemblyFromKmer (:134-205) = iterations start..end
 * Outputs: caller-allocated arrays of cap_n records / cap_key / cap_ext bases; n / need_key / need_ext are set; RFX_E_CAP
 * when a capacity is short.  Keys of more than 124 bases: RFX_E_LIMIT (the reference's k-mer list ends at 95). */
typedef struct {
    int64_t n;
    uint8_t *key; int64_t *key_off;      /* key i = key[key_off[i] .. key_off[i+1]) */
    uint8_t *ext; int64_t *ext_off;
    int32_t *marker, *left, *right;
    int64_t cap_n, cap_key, cap_ext, need_key, need_ext;
} rfx_dyn_records;
/* DynamicKmerBinarizerFromReducedToSubKmer (P/ReflexivDSDynamicKmerFirstFour.java:2931-3016 and Iteration's twin) on the device: the
 * text rows of the hand-over files -> records.  text + row_off[n_rows + 1]: row i = text[row_off[i], row_off[i+1]), its fields joined
 * by ',' (a trailing newline is ignored).  form 0: (k-mer, "m|l|r") -> key = the k-mer without its last base, extension = that base,
 * orientation 1; form 1: (sub-k-mer, "m|l|r", extension).  A leading '(' / trailing ')' of the tuple text is dropped; left / right
 * clamped to +-30000 as the reference reads its attribute long back (:2340-2366).  Outputs as for the other rfx_dyn_* calls. */
int rfx_dyn_binarize(rfx_ctx *ctx, const char *text, const int64_t *row_off, int64_t n_rows, int form, rfx_dyn_records *out);
int rfx_dyn_sort(rfx_ctx *ctx, const rfx_dyn_records *in, int P, rfx_dyn_records *out, int64_t *part_start);
int rfx_dyn_random_reflection(rfx_ctx *ctx, const rfx_dyn_records *in, const int64_t *part_start, int P, rfx_dyn_records *out);
int rfx_dyn_extend_pass(rfx_ctx *ctx, const rfx_dyn_records *in, const int64_t *part_start, int P, int stage, int start_iteration,
                        int start_marker, rfx_dyn_records *out, int64_t *out_part_start);
int rfx_dyn_run(rfx_ctx *ctx, const rfx_dyn_records *in, int P, int random_reflection, int passes_first_four, int start_iteration,
                int end_iteration, rfx_dyn_records *out, int64_t *trace, int64_t trace_cap, int64_t *n_trace);
/* the Row form of the layout, for the shim that converts: blocks <-> base codes, the attribute long <-> its three ints */
int rfx_dyn_blocks_to_bases(const int64_t *blocks, int n_blocks, uint8_t *out, int cap);
int rfx_dyn_bases_to_blocks(const uint8_t *bases, int n, int64_t *out, int cap);
int64_t rfx_dyn_attribute(int marker, int left, int right);
void rfx_dyn_attribute_unpack(int64_t attribute, int *marker, int *left, int *right);

/* Synthetic reads: integer-only counter-based generator, bit-identical to
 * oracle/reflexiv_oracle.c orc_synth_*.  Writes packed reads straight into HBM. */
int rfx_dev_synth_genome(rfx_ctx *ctx, uint64_t seed, int64_t genome_len, uint64_t *d_genome);
int rfx_dev_synth_reads(rfx_ctx *ctx, uint64_t seed, const uint64_t *d_genome, int64_t genome_len,
                        int64_t first_read, int64_t n_reads, int read_len, uint32_t err_per_2_32,
                        int words_per_read, uint64_t *d_words);

/* Plain stable radix sort of (key, value) pairs in HBM -- exposed for tests and for the
 * driver; d_tmp_* are same-sized scratch arrays. key_bits = significant low bits. */
int rfx_dev_sort_pairs(rfx_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int64_t n,
                       int key_bits, uint64_t *d_tmp_keys, uint32_t *d_tmp_vals);

/* Timing of the last rfx_dev_count_* call, per kernel family, from HIP events recorded on
 * the context's stream (ms).  names: "hist1","part1","hist2","part2","leaf","sort". */
int rfx_last_count_timing(rfx_ctx *ctx, const char *name, float *ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
