/*
 * reflexiv_jni.c -- the JNI layer between Reflexiv's Java driver and libreflexiv_hip.so
 * (include/reflexiv_hip.h).  One native method of uni.bielefeld.cmg.reflexiv.gpu.Rfx per C-ABI entry
 * point of the hot path; each replaces the body of one Spark operator class of the reference (cited in the
 * header next to the entry point it forwards to).
 *
 * Build (on a machine with a JDK; none exists in the build container, so this file is compiled there by
 * nobody -- tests/test_jni_sources.py checks it against the header and Rfx.java instead):
 *   gcc -O2 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       jni/reflexiv_jni.c -Lreflexiv_amd -lreflexiv_hip -Wl,-rpath,'$ORIGIN' -o reflexiv_amd/libreflexiv_jni.so
 *
 * Conventions: a context handle is the rfx_ctx pointer as a jlong; record sets travel as
 * uni.bielefeld.cmg.reflexiv.gpu.RfxRecords (six primitive arrays + n + keyWords), pinned with
 * Get<Type>ArrayElements for the duration of the call (no critical regions around GPU work); output record sets are allocated by the Java
 * side at their upper bound and trimmed there (n is written back).  A negative rfx_status becomes a
 * RuntimeException, so Spark's task retry / job abort semantics are those of the reference.
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "reflexiv_hip.h"

#define RFX_CLASS(name) Java_uni_bielefeld_cmg_reflexiv_gpu_Rfx_##name
#define RFX_N_PARAMS ((int)(sizeof(rfx_params) / sizeof(int32_t)))        /* 13 */

static void throw_rfx(JNIEnv *env, rfx_ctx *ctx, int st, const char *where) {
    char msg[640];
    snprintf(msg, sizeof msg, "%s: rfx status %d%s%s", where, st, ctx ? ": " : "", ctx ? rfx_last_error(ctx) : "");
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

static rfx_ctx *ctx_of(jlong h) { return (rfx_ctx *)(intptr_t)h; }

/* ------------------------------------------------------------------ RfxRecords <-> rfx_records */

typedef struct {
    jobject obj;
    jlongArray key, ext_off, ext;
    jintArray marker, left, right;
    rfx_records r;
} pinned_records;

static jfieldID F_n, F_keyWords, F_key, F_marker, F_extOff, F_ext, F_left, F_right;

static int records_ids(JNIEnv *env) {
    if (F_n) return 1;
    jclass c = (*env)->FindClass(env, "uni/bielefeld/cmg/reflexiv/gpu/RfxRecords");
    if (!c) return 0;
    F_n = (*env)->GetFieldID(env, c, "n", "J");
    F_keyWords = (*env)->GetFieldID(env, c, "keyWords", "I");
    F_key = (*env)->GetFieldID(env, c, "key", "[J");
    F_marker = (*env)->GetFieldID(env, c, "marker", "[I");
    F_extOff = (*env)->GetFieldID(env, c, "extOff", "[J");
    F_ext = (*env)->GetFieldID(env, c, "ext", "[J");
    F_left = (*env)->GetFieldID(env, c, "left", "[I");
    F_right = (*env)->GetFieldID(env, c, "right", "[I");
    return F_n && F_keyWords && F_key && F_marker && F_extOff && F_ext && F_left && F_right;
}

/* the six arrays through Get<Type>ArrayElements -- NOT GetPrimitiveArrayCritical: field reads of the second record set, the
 * write-back of n / keyWords and the GPU call itself all happen while they are held, none of which a critical region
 * allows (no other JNI call, no blocking); capacities come from the array lengths */
static int records_pin(JNIEnv *env, jobject o, pinned_records *p) {
    memset(p, 0, sizeof *p);
    if (!records_ids(env)) return 0;
    p->obj = o;
    p->key = (jlongArray)(*env)->GetObjectField(env, o, F_key);
    p->marker = (jintArray)(*env)->GetObjectField(env, o, F_marker);
    p->ext_off = (jlongArray)(*env)->GetObjectField(env, o, F_extOff);
    p->ext = (jlongArray)(*env)->GetObjectField(env, o, F_ext);
    p->left = (jintArray)(*env)->GetObjectField(env, o, F_left);
    p->right = (jintArray)(*env)->GetObjectField(env, o, F_right);
    const int kw = (*env)->GetIntField(env, o, F_keyWords);
    p->r.n = (*env)->GetLongField(env, o, F_n);
    p->r.key_words = kw;
    p->r.cap_n = (*env)->GetArrayLength(env, p->marker);
    p->r.cap_words = (*env)->GetArrayLength(env, p->ext);
    p->r.key = (uint64_t *)(*env)->GetLongArrayElements(env, p->key, NULL);
    p->r.marker = (int32_t *)(*env)->GetIntArrayElements(env, p->marker, NULL);
    p->r.ext_off = (int64_t *)(*env)->GetLongArrayElements(env, p->ext_off, NULL);
    p->r.ext = (uint64_t *)(*env)->GetLongArrayElements(env, p->ext, NULL);
    p->r.left = (int32_t *)(*env)->GetIntArrayElements(env, p->left, NULL);
    p->r.right = (int32_t *)(*env)->GetIntArrayElements(env, p->right, NULL);
    return p->r.key && p->r.marker && p->r.ext_off && p->r.ext && p->r.left && p->r.right;
}

/* small long[] arguments (partition starts): a native copy in, a region write out -- nothing is pinned during the GPU call */
static int64_t *longs_in(JNIEnv *env, jlongArray a) {
    const jsize n = (*env)->GetArrayLength(env, a);
    int64_t *buf = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    if (buf && n > 0) (*env)->GetLongArrayRegion(env, a, 0, n, (jlong *)buf);
    return buf;
}
static void longs_out(JNIEnv *env, jlongArray a, int64_t *buf, int write_back) {
    if (!buf) return;
    if (write_back) (*env)->SetLongArrayRegion(env, a, 0, (*env)->GetArrayLength(env, a), (const jlong *)buf);
    free(buf);
}

/* mode 0: copy back (outputs), JNI_ABORT: inputs */
static void records_unpin(JNIEnv *env, pinned_records *p, jint mode) {
    if (p->r.right) (*env)->ReleaseIntArrayElements(env, p->right, (jint *)p->r.right, mode);
    if (p->r.left) (*env)->ReleaseIntArrayElements(env, p->left, (jint *)p->r.left, mode);
    if (p->r.ext) (*env)->ReleaseLongArrayElements(env, p->ext, (jlong *)p->r.ext, mode);
    if (p->r.ext_off) (*env)->ReleaseLongArrayElements(env, p->ext_off, (jlong *)p->r.ext_off, mode);
    if (p->r.marker) (*env)->ReleaseIntArrayElements(env, p->marker, (jint *)p->r.marker, mode);
    if (p->r.key) (*env)->ReleaseLongArrayElements(env, p->key, (jlong *)p->r.key, mode);
    if (mode == 0 && p->obj) {
        (*env)->SetLongField(env, p->obj, F_n, (jlong)p->r.n);
        (*env)->SetIntField(env, p->obj, F_keyWords, (jint)(p->r.key_words > 1 ? p->r.key_words : 1));
    }
}

/* ------------------------------------------------------------------------------------ context */

JNIEXPORT jint JNICALL RFX_CLASS(version)(JNIEnv *env, jclass c) {
    (void)env; (void)c;
    return rfx_version();
}

JNIEXPORT jlong JNICALL RFX_CLASS(ctxCreate)(JNIEnv *env, jclass c, jint device) {
    (void)c;
    rfx_ctx *ctx = NULL;
    const int st = rfx_ctx_create(device, &ctx);
    if (st != RFX_OK) { throw_rfx(env, NULL, st, "rfx_ctx_create (a gfx950 GPU is required; there is no CPU fallback)"); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL RFX_CLASS(ctxDestroy)(JNIEnv *env, jclass c, jlong h) {
    (void)env; (void)c;
    rfx_ctx_destroy(ctx_of(h));
}

/* U/DefaultParam.java defaults as the library sees them: int[13] in rfx_params field order */
JNIEXPORT jintArray JNICALL RFX_CLASS(defaultParams)(JNIEnv *env, jclass c) {
    (void)c;
    rfx_params p;
    rfx_default_params(&p);
    jintArray a = (*env)->NewIntArray(env, RFX_N_PARAMS);
    if (a) (*env)->SetIntArrayRegion(env, a, 0, RFX_N_PARAMS, (const jint *)&p);
    return a;
}

/* ---------------------------------------------------------------------- extraction and counting */

/* ReverseComplementKmerBinaryExtraction.call (P/ReflexivMain.java:3013-3075) and, wide != 0,
 * ReverseComplementKmerBinaryExtractionFromDataset64.call (P/ReflexivDataFrameCounter64.java:401-650):
 * -> long[n * W] canonical k-mers in read / window order (W = 1, or k/32+1) */
static jlongArray extract_common(JNIEnv *env, jlong h, jbyteArray bases, jlongArray readOff, jint k, jint fc, jint ec, int wide) {
    rfx_ctx *ctx = ctx_of(h);
    const jsize nOff = (*env)->GetArrayLength(env, readOff);
    const int W = wide ? k / 32 + 1 : 1;
    int64_t n = 0;
    jlongArray out = NULL;
    for (int pass = 0; pass < 2; pass++) {                 /* size query, then the real call */
        if (pass == 1) { out = (*env)->NewLongArray(env, (jsize)(n * W)); if (!out) return NULL; }
        jbyte *b = (*env)->GetByteArrayElements(env, bases, NULL);
        jlong *o = (*env)->GetLongArrayElements(env, readOff, NULL);
        jlong *dst = pass ? (*env)->GetLongArrayElements(env, out, NULL) : NULL;
        int st = wide ? rfx_extract_canon_w(ctx, (const uint8_t *)b, (const int64_t *)o, nOff - 1, k, fc, ec, (uint64_t *)dst, pass ? n : 0, &n)
                      : rfx_extract_canon(ctx, (const uint8_t *)b, (const int64_t *)o, nOff - 1, k, fc, ec, (uint64_t *)dst, pass ? n : 0, &n);
        if (dst) (*env)->ReleaseLongArrayElements(env, out, dst, 0);
        (*env)->ReleaseLongArrayElements(env, readOff, o, JNI_ABORT);
        (*env)->ReleaseByteArrayElements(env, bases, b, JNI_ABORT);
        if (st != RFX_OK && !(pass == 0 && st == RFX_E_CAP)) { throw_rfx(env, ctx, st, wide ? "rfx_extract_canon_w" : "rfx_extract_canon"); return NULL; }
    }
    return out;
}

JNIEXPORT jlongArray JNICALL RFX_CLASS(extractCanon)(JNIEnv *env, jclass c, jlong h, jbyteArray bases, jlongArray readOff,
                                                    jint k, jint frontClip, jint endClip) {
    (void)c;
    return extract_common(env, h, bases, readOff, k, frontClip, endClip, 0);
}

JNIEXPORT jlongArray JNICALL RFX_CLASS(extractCanonW)(JNIEnv *env, jclass c, jlong h, jbyteArray bases, jlongArray readOff,
                                                     jint k, jint frontClip, jint endClip) {
    (void)c;
    return extract_common(env, h, bases, readOff, k, frontClip, endClip, 1);
}

/* reduceByKey(KmerCounting) + filter(KmerCoverageFilter) (P/ReflexivMain.java:155,160-163,2895-2899,3115-3119):
 * outKeys / outCounts hold kmers.length entries; returns the number of survivors (ascending by k-mer) */
JNIEXPORT jlong JNICALL RFX_CLASS(countFilter)(JNIEnv *env, jclass c, jlong h, jlongArray kmers, jint minCov, jint maxCov, jint twin,
                                              jlongArray outKeys, jintArray outCounts) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const jsize n = (*env)->GetArrayLength(env, kmers);
    int64_t m = 0, d = 0;
    jlong *in = (*env)->GetLongArrayElements(env, kmers, NULL);
    jlong *ok = (*env)->GetLongArrayElements(env, outKeys, NULL);
    jint *oc = (*env)->GetIntArrayElements(env, outCounts, NULL);
    const int st = rfx_count_filter(ctx, (const uint64_t *)in, n, minCov, maxCov, twin, (uint64_t *)ok, (int32_t *)oc,
                                    (*env)->GetArrayLength(env, outCounts), &m, &d);
    (*env)->ReleaseIntArrayElements(env, outCounts, oc, 0);
    (*env)->ReleaseLongArrayElements(env, outKeys, ok, 0);
    (*env)->ReleaseLongArrayElements(env, kmers, in, JNI_ABORT);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_count_filter"); return -1; }
    return (jlong)m;
}

/* groupBy("kmerBlocks").count() + the two filters (P/ReflexivDataFrameCounter64.java:191-205), k > 31 */
JNIEXPORT jlong JNICALL RFX_CLASS(countFilterW)(JNIEnv *env, jclass c, jlong h, jlongArray kmers, jint k, jint minCov, jint maxCov,
                                               jlongArray outKeys, jlongArray outCounts) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const int W = k / 32 + 1;
    const jsize n = (*env)->GetArrayLength(env, kmers) / W;
    int64_t m = 0, d = 0;
    jlong *in = (*env)->GetLongArrayElements(env, kmers, NULL);
    jlong *ok = (*env)->GetLongArrayElements(env, outKeys, NULL);
    jlong *oc = (*env)->GetLongArrayElements(env, outCounts, NULL);
    const int st = rfx_count_filter_w(ctx, (const uint64_t *)in, n, k, minCov, maxCov, (uint64_t *)ok, (int64_t *)oc,
                                      (*env)->GetArrayLength(env, outCounts), &m, &d);
    (*env)->ReleaseLongArrayElements(env, outCounts, oc, 0);
    (*env)->ReleaseLongArrayElements(env, outKeys, ok, 0);
    (*env)->ReleaseLongArrayElements(env, kmers, in, JNI_ABORT);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_count_filter_w"); return -1; }
    return (jlong)m;
}

/* ---------------------------------------------------------------------------- record operators */

/* KmerReverseComplement.call + ForwardSubKmerExtraction.call (P/ReflexivMain.java:2910-2930, 2709-2730;
 * k > 31: DSKmerReverseComplement + DSForwardSubKmerExtraction, P/ReflexivDSMain64.java:10706-10755, 10363-10403) */
JNIEXPORT void JNICALL RFX_CLASS(rcExpandSubkmer)(JNIEnv *env, jclass c, jlong h, jlongArray kmers, jintArray counts, jint k, jobject out) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const jsize n = (*env)->GetArrayLength(env, counts);
    pinned_records po;
    int st = RFX_E_ARG;
    if (records_pin(env, out, &po)) {
        jlong *km = (*env)->GetLongArrayElements(env, kmers, NULL);
        jint *cn = (*env)->GetIntArrayElements(env, counts, NULL);
        st = rfx_rc_expand_subkmer(ctx, (const uint64_t *)km, (const int32_t *)cn, n, k, &po.r);
        (*env)->ReleaseIntArrayElements(env, counts, cn, JNI_ABORT);
        (*env)->ReleaseLongArrayElements(env, kmers, km, JNI_ABORT);
    }
    records_unpin(env, &po, 0);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_rc_expand_subkmer");
}

/* sortByKey() (P/ReflexivMain.java:179,191,211,235,247,286) for callers that keep a whole RDD on one GPU;
 * under Spark's own shuffle the driver keeps sortByKey() and does not call this */
JNIEXPORT void JNICALL RFX_CLASS(sortRecords)(JNIEnv *env, jclass c, jlong h, jobject in, jint P, jobject out, jlongArray partStart) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o) {
        int64_t *ps = longs_in(env, partStart);
        st = rfx_sort_records(ctx, &pi.r, P, &po.r, (int64_t *)ps);
        longs_out(env, partStart, ps, 1);
    }
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_sort_records");
}

/* which: 0 FilterForkSubKmer[WithErrorCorrection].call (P/ReflexivMain.java:2412-2540),
 *        1 FilterForkReflectedSubKmer[WithErrorCorrection].call (:2550-2696) */
static void fork_common(JNIEnv *env, jlong h, int which, jobject in, jlongArray partStart, jint k, jint minErr, jint twin,
                        jobject out, jlongArray outPartStart) {
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int P = (*env)->GetArrayLength(env, partStart) - 1;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o && P >= 1) {
        int64_t *ps = longs_in(env, partStart);
        int64_t *ops = longs_in(env, outPartStart);
        st = which ? rfx_fork_filter_reflected(ctx, &pi.r, (const int64_t *)ps, P, k, minErr, twin, &po.r, (int64_t *)ops)
                   : rfx_fork_filter_forward(ctx, &pi.r, (const int64_t *)ps, P, k, minErr, twin, &po.r, (int64_t *)ops);
        longs_out(env, outPartStart, ops, 1);
        longs_out(env, partStart, ps, 0);
    }
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, which ? "rfx_fork_filter_reflected" : "rfx_fork_filter_forward");
}

JNIEXPORT void JNICALL RFX_CLASS(forkFilterForward)(JNIEnv *env, jclass c, jlong h, jobject in, jlongArray partStart, jint k,
                                                   jint minErrorCov, jint twin, jobject out, jlongArray outPartStart) {
    (void)c;
    fork_common(env, h, 0, in, partStart, k, minErrorCov, twin, out, outPartStart);
}

JNIEXPORT void JNICALL RFX_CLASS(forkFilterReflected)(JNIEnv *env, jclass c, jlong h, jobject in, jlongArray partStart, jint k,
                                                     jint minErrorCov, jint twin, jobject out, jlongArray outPartStart) {
    (void)c;
    fork_common(env, h, 1, in, partStart, k, minErrorCov, twin, out, outPartStart);
}

/* ReflectedSubKmerExtractionFromForward.call (P/ReflexivMain.java:2742-2768) */
JNIEXPORT void JNICALL RFX_CLASS(reflectFromForward)(JNIEnv *env, jclass c, jlong h, jobject in, jint k, jobject out) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o) st = rfx_reflect_from_forward(ctx, &pi.r, k, &po.r);
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_reflect_from_forward");
}

/* kmerRandomReflection.call (P/ReflexivMain.java:2783-2885) */
JNIEXPORT void JNICALL RFX_CLASS(randomReflection)(JNIEnv *env, jclass c, jlong h, jobject in, jlongArray partStart, jint k, jobject out) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int P = (*env)->GetArrayLength(env, partStart) - 1;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o && P >= 1) {
        int64_t *ps = longs_in(env, partStart);
        st = rfx_random_reflection(ctx, &pi.r, (const int64_t *)ps, P, k, &po.r);
        longs_out(env, partStart, ps, 0);
    }
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_random_reflection");
}

/* ExtendReflexivKmer / ...ToArrayFirstTime / ...ToArrayLoop .call (P/ReflexivMain.java:2048-2362, 1594-1974, 792-1519);
 * stage 0 / 1 / 2.  scramble: param.scramble as DSExtendReflexivKmerToArrayLoop of P/ReflexivDSMain64.java reads it
 * (:7484-7486); 2 everywhere else. */
JNIEXPORT void JNICALL RFX_CLASS(extendPass)(JNIEnv *env, jclass c, jlong h, jobject in, jlongArray partStart, jint k, jint twin,
                                            jint stage, jint scramble, jobject out, jlongArray outPartStart) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int P = (*env)->GetArrayLength(env, partStart) - 1;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o && P >= 1) {
        int64_t *ps = longs_in(env, partStart);
        int64_t *ops = longs_in(env, outPartStart);
        st = scramble == 2 ? rfx_extend_pass(ctx, &pi.r, (const int64_t *)ps, P, k, twin, stage, &po.r, (int64_t *)ops)
                           : rfx_extend_pass_w(ctx, &pi.r, (const int64_t *)ps, P, k, stage, scramble, &po.r, (int64_t *)ops);
        longs_out(env, outPartStart, ops, 1);
        longs_out(env, partStart, ps, 0);
    }
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_extend_pass");
}

/* one operator class of the k > 31 from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712); op = RFX_OP_* */
JNIEXPORT void JNICALL RFX_CLASS(extrasOperator)(JNIEnv *env, jclass c, jlong h, jint op, jobject in, jlongArray partStart, jint k,
                                                jobject out, jlongArray outPartStart) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi, po;
    int st = RFX_E_ARG;
    const int P = (*env)->GetArrayLength(env, partStart) - 1;
    const int ok_i = records_pin(env, in, &pi), ok_o = records_pin(env, out, &po);
    if (ok_i && ok_o && P >= 1) {
        int64_t *ps = longs_in(env, partStart);
        int64_t *ops = longs_in(env, outPartStart);
        st = rfx_extras_operator(ctx, op, &pi.r, (const int64_t *)ps, P, k, &po.r, (int64_t *)ops);
        longs_out(env, outPartStart, ops, 1);
        longs_out(env, partStart, ps, 0);
    }
    records_unpin(env, &po, 0);
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) throw_rfx(env, ctx, st, "rfx_extras_operator");
}

/* BinaryReflexivKmerArrayToString + KmerToContig + TagContigID (P/ReflexivMain.java:696-741, 590-637, 573-581):
 * the text saveAsTextFile writes for these records, ids counted from 0 (the caller adds zipWithIndex offsets
 * when it formats partition by partition) */
JNIEXPORT jbyteArray JNICALL RFX_CLASS(contigsText)(JNIEnv *env, jclass c, jlong h, jobject in, jint k, jint minContig, jint twin) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    pinned_records pi;
    if (!records_pin(env, in, &pi)) { records_unpin(env, &pi, JNI_ABORT); throw_rfx(env, ctx, RFX_E_ARG, "rfx_contigs_text"); return NULL; }
    int64_t len = 0, nc = 0;
    int st = rfx_contigs_text(ctx, &pi.r, k, minContig, twin, NULL, 0, &len, &nc);
    jbyteArray out = NULL;
    if (st == RFX_OK || st == RFX_E_CAP) {
        records_unpin(env, &pi, JNI_ABORT);
        out = (*env)->NewByteArray(env, (jsize)len);
        if (!out || !records_pin(env, in, &pi)) { records_unpin(env, &pi, JNI_ABORT); return NULL; }
        jbyte *dst = (*env)->GetByteArrayElements(env, out, NULL);
        st = rfx_contigs_text(ctx, &pi.r, k, minContig, twin, (char *)dst, len, &len, &nc);
        (*env)->ReleaseByteArrayElements(env, out, dst, 0);
    }
    records_unpin(env, &pi, JNI_ABORT);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_contigs_text"); return NULL; }
    return out;
}

/* ---------------------------------------------------------------------------- resident pipeline */

/* The whole path (P/ReflexivMain.java:95-322) in one call: ASCII reads of any length up, contig text back.
 * params: int[13] in rfx_params field order (Rfx.defaultParams()).  k <= 31. */
JNIEXPORT jbyteArray JNICALL RFX_CLASS(assembleReads)(JNIEnv *env, jclass c, jlong h, jbyteArray bases, jlongArray readOff, jintArray params) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    if ((*env)->GetArrayLength(env, params) != RFX_N_PARAMS) { throw_rfx(env, ctx, RFX_E_ARG, "rfx_assemble_reads (params: Rfx.defaultParams())"); return NULL; }
    rfx_params prm;
    (*env)->GetIntArrayRegion(env, params, 0, RFX_N_PARAMS, (jint *)&prm);
    const jsize nOff = (*env)->GetArrayLength(env, readOff);
    int64_t cap = (int64_t)(*env)->GetArrayLength(env, bases) * 3 + (1 << 20);    /* both strands + headers + line breaks */
    for (;;) {
        char *buf = (char *)malloc((size_t)cap);             /* native staging: no JNI call is made while arrays are pinned */
        if (!buf) { throw_rfx(env, ctx, RFX_E_HIP, "rfx_assemble_reads (out of host memory)"); return NULL; }
        /* NOT a critical region: the call blocks on the GPU for the whole run, and a critical region stalls every other
         * thread's GC for that long (Get<Type>ArrayElements pins or copies; either is fine for a read-only input) */
        jbyte *b = (*env)->GetByteArrayElements(env, bases, NULL);
        jlong *o = (*env)->GetLongArrayElements(env, readOff, NULL);
        if (!b || !o) {
            if (b) (*env)->ReleaseByteArrayElements(env, bases, b, JNI_ABORT);
            if (o) (*env)->ReleaseLongArrayElements(env, readOff, o, JNI_ABORT);
            free(buf);
            return NULL;                                     /* (OutOfMemoryError is pending) */
        }
        int64_t len = 0, nc = 0, ntr = 0, kept = 0;
        const int st = rfx_assemble_reads(ctx, (const uint8_t *)b, (const int64_t *)o, nOff - 1, &prm, buf, cap, &len, &nc,
                                          NULL, 0, &ntr, &kept);
        (*env)->ReleaseLongArrayElements(env, readOff, o, JNI_ABORT);
        (*env)->ReleaseByteArrayElements(env, bases, b, JNI_ABORT);
        if (st == RFX_E_CAP && len > cap) { cap = len; free(buf); continue; }
        if (st != RFX_OK) { free(buf); throw_rfx(env, ctx, st, "rfx_assemble_reads"); return NULL; }
        jbyteArray out = (*env)->NewByteArray(env, (jsize)len);
        if (out) (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte *)buf);
        free(buf);
        return out;
    }
}


/* ---------------------------------------------------------------------------- several GPUs of one node */

/* the shuffle of reduceByKey (P/ReflexivMain.java:155) as an RCCL all-to-all inside the library: one executor task per
 * GPU (Spark barrier stage), each with its context and one communicator.  commUniqueId: task 0 makes the 128-byte id and
 * hands it round (BarrierTaskContext.allGather); commInit: collective. */
JNIEXPORT jbyteArray JNICALL RFX_CLASS(commUniqueId)(JNIEnv *env, jclass c) {
    (void)c;
    uint8_t id[128];
    const int st = rfx_comm_unique_id(id);
    if (st != RFX_OK) { throw_rfx(env, NULL, st, "rfx_comm_unique_id"); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, 128);
    if (out) (*env)->SetByteArrayRegion(env, out, 0, 128, (const jbyte *)id);
    return out;
}

JNIEXPORT jlong JNICALL RFX_CLASS(commInit)(JNIEnv *env, jclass c, jlong h, jbyteArray id, jint rank, jint world) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    uint8_t buf[128];
    if ((*env)->GetArrayLength(env, id) != 128) { throw_rfx(env, ctx, RFX_E_ARG, "rfx_comm_init (id: 128 bytes)"); return 0; }
    (*env)->GetByteArrayRegion(env, id, 0, 128, (jbyte *)buf);
    rfx_comm *comm = NULL;
    const int st = rfx_comm_init(ctx, buf, rank, world, &comm);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_comm_init"); return 0; }
    return (jlong)(intptr_t)comm;
}

JNIEXPORT void JNICALL RFX_CLASS(commDestroy)(JNIEnv *env, jclass c, jlong comm) {
    (void)env; (void)c;
    rfx_comm_destroy((rfx_comm *)(intptr_t)comm);
}

/* count() of the stop rule / a barrier over the tasks: sum (op 0) or max (op 1) of up to 8 longs, in place */
JNIEXPORT void JNICALL RFX_CLASS(commAllReduce)(JNIEnv *env, jclass c, jlong h, jlong comm, jlongArray vals, jint op) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const jsize n = (*env)->GetArrayLength(env, vals);
    int64_t v[8];
    if (n < 1 || n > 8) { throw_rfx(env, ctx, RFX_E_ARG, "rfx_comm_all_reduce_i64 (1..8 values)"); return; }
    (*env)->GetLongArrayRegion(env, vals, 0, n, (jlong *)v);
    const int st = rfx_comm_all_reduce_i64((rfx_comm *)(intptr_t)comm, v, (int)n, op);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_comm_all_reduce_i64"); return; }
    (*env)->SetLongArrayRegion(env, vals, 0, n, (const jlong *)v);
}

/* The whole path on several GPUs (P/ReflexivMain.java:95-322 with the shuffles of :155 AND of every sortByKey, :179-286,
 * over RCCL; gatherBelow as rfx_dev_sharded_assemble's: -1 = default): every task passes ITS
 * partition's reads; the contig text comes back on rank 0 (an empty array on the others).  params: int[13]; k = 21..31.
 * totals (long[3], optional): k-mer instances, distinct k-mers, k-mers kept -- over all tasks. */
JNIEXPORT jbyteArray JNICALL RFX_CLASS(shardedAssembleReads)(JNIEnv *env, jclass c, jlong h, jlong comm, jbyteArray bases, jlongArray readOff,
                                                            jintArray params, jint generations, jlong gatherBelow, jlongArray totals) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    if ((*env)->GetArrayLength(env, params) != RFX_N_PARAMS) { throw_rfx(env, ctx, RFX_E_ARG, "rfx_sharded_assemble_reads (params: Rfx.defaultParams())"); return NULL; }
    rfx_params prm;
    (*env)->GetIntArrayRegion(env, params, 0, RFX_N_PARAMS, (jint *)&prm);
    const jsize nOff = (*env)->GetArrayLength(env, readOff);
    /* the text lands on rank 0 only and holds every task's contigs: size it from the job's instances after a first try */
    int64_t cap = (int64_t)(*env)->GetArrayLength(env, bases) * 3 + (1 << 20);
    for (;;) {
        char *buf = (char *)malloc((size_t)cap);
        if (!buf) { throw_rfx(env, ctx, RFX_E_HIP, "rfx_sharded_assemble_reads (out of host memory)"); return NULL; }
        /* NOT a critical region: this is a blocking COLLECTIVE (upload, RCCL all-to-all, the whole extend loop) that waits for
         * the other tasks of the barrier stage.  A critical region blocks the collector; a task of the same JVM that has not
         * entered yet could stall in a GC waiting for this region while this task waits for it inside RCCL -- a deadlock. */
        jbyte *b = (*env)->GetByteArrayElements(env, bases, NULL);
        jlong *o = (*env)->GetLongArrayElements(env, readOff, NULL);
        if (!b || !o) {
            if (b) (*env)->ReleaseByteArrayElements(env, bases, b, JNI_ABORT);
            if (o) (*env)->ReleaseLongArrayElements(env, readOff, o, JNI_ABORT);
            free(buf);
            return NULL;                                     /* (OutOfMemoryError is pending; the peers time out in RCCL) */
        }
        int64_t len = 0, nc = 0, ntr = 0, tot[3] = {0, 0, 0};
        const int st = rfx_sharded_assemble_reads(ctx, (rfx_comm *)(intptr_t)comm, (const uint8_t *)b, (const int64_t *)o, nOff - 1, &prm,
                                                  generations, (int64_t)gatherBelow, buf, cap, &len, &nc, NULL, 0, &ntr, tot);
        (*env)->ReleaseLongArrayElements(env, readOff, o, JNI_ABORT);
        (*env)->ReleaseByteArrayElements(env, bases, b, JNI_ABORT);
        /* RFX_E_CAP is returned on EVERY rank with the length rank 0 needs, so every task repeats the collective together */
        if (st == RFX_E_CAP && len > cap) { cap = len; free(buf); continue; }
        if (st != RFX_OK) { free(buf); throw_rfx(env, ctx, st, "rfx_sharded_assemble_reads"); return NULL; }
        if (totals && (*env)->GetArrayLength(env, totals) >= 3) (*env)->SetLongArrayRegion(env, totals, 0, 3, (const jlong *)tot);
        jbyteArray out = (*env)->NewByteArray(env, (jsize)len);
        if (out) (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte *)buf);
        free(buf);
        return out;
    }
}


/* Contig RC de-duplication: ReflexivDSDynamicKmerDedup.assemblyFromKmer (P/ReflexivDSDynamicKmerDedup.java:138-339) on the
 * contig text a run wrote (rows of saveAsTextFile joined by '\n') -> the text TagRowContigDSID (:3397-3443) writes */
JNIEXPORT jbyteArray JNICALL RFX_CLASS(dedupContigText)(JNIEnv *env, jclass c, jlong h, jbyteArray contigText, jint minContig) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const jsize n = (*env)->GetArrayLength(env, contigText);
    char *src = (char *)malloc((size_t)n + 1), *dst = (char *)malloc((size_t)n + 4096);
    if (!src || !dst) { free(src); free(dst); throw_rfx(env, ctx, RFX_E_HIP, "rfx_dedup_contig_text (out of host memory)"); return NULL; }
    (*env)->GetByteArrayRegion(env, contigText, 0, n, (jbyte *)src);
    int64_t len = 0, nc = 0;
    const int st = rfx_dedup_contig_text(ctx, src, (int64_t)n, minContig, dst, (int64_t)n + 4096, &len, &nc, NULL);
    jbyteArray out = NULL;
    if (st == RFX_OK) {
        out = (*env)->NewByteArray(env, (jsize)len);
        if (out) (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte *)dst);
    }
    free(src); free(dst);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_dedup_contig_text"); return NULL; }
    return out;
}


/* ---------------------------------------------------------------------------- the dynamic-k ("meta") passes */

/* DSExtendReflexivKmer / DSExtendReflexivKmerToArrayLoop of ReflexivDSDynamicKmerFirstFour / ...Iteration on one task's rows
 * (P/ReflexivDSDynamicKmerFirstFour.java:1581-2373, P/ReflexivDSDynamicKmerIteration.java:465-1249).  Rows travel flattened:
 * keyBlocks / extBlocks = the rows' long[] blocks back to back with keyOff / extOff (in blocks), attribute = the packed long.
 * The shim converts to the library's base-code form (rfx_dyn_blocks_to_bases / rfx_dyn_attribute_unpack), runs
 * rfx_dyn_run (random_reflection, passes, iterations as for the drivers) and converts back; the result arrays come back in
 * a long[4][] = {keyBlocks, keyOff, extBlocks, extOff} plus outAttr (sized by the caller to the input row count). */
JNIEXPORT jlong JNICALL RFX_CLASS(dynRun)(JNIEnv *env, jclass c, jlong h, jlongArray keyBlocks, jlongArray keyOff, jlongArray extBlocks,
                                         jlongArray extOff, jlongArray attr, jint P, jint randomReflection, jint passesFirstFour,
                                         jint startIteration, jint endIteration, jlongArray outKeyBlocks, jlongArray outKeyOff,
                                         jlongArray outExtBlocks, jlongArray outExtOff, jlongArray outAttr) {
    (void)c;
    rfx_ctx *ctx = ctx_of(h);
    const jsize n = (*env)->GetArrayLength(env, attr);
    /* the offset arrays must hold n + 1 entries, ascending from 0, inside their block arrays: checked BEFORE anything is
     * indexed with them (a short keyOff used to be a native crash inside the JVM) */
    if ((*env)->GetArrayLength(env, keyOff) < n + 1 || (*env)->GetArrayLength(env, extOff) < n + 1) {
        throw_rfx(env, ctx, RFX_E_ARG, "rfx_dyn_run (keyOff / extOff: one entry per row + 1)");
        return 0;
    }
    int64_t *kb = longs_in(env, keyBlocks), *ko = longs_in(env, keyOff), *eb = longs_in(env, extBlocks), *eo = longs_in(env, extOff),
            *at = longs_in(env, attr);
    if (!kb || !ko || !eb || !eo || !at) {
        longs_out(env, keyBlocks, kb, 0); longs_out(env, keyOff, ko, 0); longs_out(env, extBlocks, eb, 0); longs_out(env, extOff, eo, 0);
        longs_out(env, attr, at, 0);
        return 0;                                            /* (OutOfMemoryError is pending) */
    }
    int st = RFX_OK;
    {
        const int64_t lkb = (*env)->GetArrayLength(env, keyBlocks), leb = (*env)->GetArrayLength(env, extBlocks);
        if (ko[0] != 0 || eo[0] != 0 || ko[n] > lkb || eo[n] > leb) st = RFX_E_ARG;
        for (jsize i = 0; i < n && st == RFX_OK; i++)
            if (ko[i + 1] < ko[i] || eo[i + 1] < eo[i]) st = RFX_E_ARG;
        if (st != RFX_OK) {
            longs_out(env, keyBlocks, kb, 0); longs_out(env, keyOff, ko, 0); longs_out(env, extBlocks, eb, 0); longs_out(env, extOff, eo, 0);
            longs_out(env, attr, at, 0);
            throw_rfx(env, ctx, st, "rfx_dyn_run (keyOff / extOff must ascend from 0 and stay inside keyBlocks / extBlocks)");
            return 0;
        }
    }
    const int64_t nkb = ko[n], neb = eo[n];
    /* to base codes */
    uint8_t *key = (uint8_t *)malloc((size_t)nkb * 31 + 1), *ext = (uint8_t *)malloc((size_t)neb * 31 + 1);
    int64_t *koff = (int64_t *)malloc((size_t)(n + 1) * 8), *eoff = (int64_t *)malloc((size_t)(n + 1) * 8);
    int32_t *mk = (int32_t *)malloc((size_t)(n + 1) * 4), *lf = (int32_t *)malloc((size_t)(n + 1) * 4), *rt = (int32_t *)malloc((size_t)(n + 1) * 4);
    int64_t *okoff = (int64_t *)malloc((size_t)(n + 1) * 8), *oeoff = (int64_t *)malloc((size_t)(n + 1) * 8);
    int32_t *omk = (int32_t *)malloc((size_t)(n + 1) * 4), *olf = (int32_t *)malloc((size_t)(n + 1) * 4), *ort = (int32_t *)malloc((size_t)(n + 1) * 4);
    int64_t pk = 0, pe = 0;
    if (!key || !ext || !koff || !eoff || !mk || !lf || !rt || !okoff || !oeoff || !omk || !olf || !ort) {
        free(key); free(ext); free(koff); free(eoff); free(mk); free(lf); free(rt);
        free(okoff); free(oeoff); free(omk); free(olf); free(ort);
        longs_out(env, keyBlocks, kb, 0); longs_out(env, keyOff, ko, 0); longs_out(env, extBlocks, eb, 0); longs_out(env, extOff, eo, 0);
        longs_out(env, attr, at, 0);
        throw_rfx(env, ctx, RFX_E_HOST, "rfx_dyn_run (out of host memory)");
        return 0;
    }
    for (jsize i = 0; i < n && st == RFX_OK; i++) {
        koff[i] = pk; eoff[i] = pe;
        const int lk = rfx_dyn_blocks_to_bases(kb + ko[i], (int)(ko[i + 1] - ko[i]), key + pk, (int)((ko[i + 1] - ko[i]) * 31));
        const int le = rfx_dyn_blocks_to_bases(eb + eo[i], (int)(eo[i + 1] - eo[i]), ext + pe, (int)((eo[i + 1] - eo[i]) * 31));
        if (lk < 0 || le < 0) st = RFX_E_ARG;
        pk += lk > 0 ? lk : 0; pe += le > 0 ? le : 0;
        int m, l, r;
        rfx_dyn_attribute_unpack(at[i], &m, &l, &r);
        mk[i] = m; lf[i] = l; rt[i] = r;
    }
    koff[n] = pk; eoff[n] = pe;
    rfx_dyn_records in = {n, key, koff, ext, eoff, mk, lf, rt, n, pk, pe, 0, 0};
    int64_t cap_b = pk + pe + 64, out_n = 0;
    uint8_t *okey = NULL, *oext = NULL;
    while (st == RFX_OK) {
        free(okey); free(oext);
        okey = (uint8_t *)malloc((size_t)cap_b); oext = (uint8_t *)malloc((size_t)cap_b);
        if (!okey || !oext) { st = RFX_E_HOST; break; }
        rfx_dyn_records out = {0, okey, okoff, oext, oeoff, omk, olf, ort, n, cap_b, cap_b, 0, 0};
        int64_t ntr = 0;
        st = rfx_dyn_run(ctx, &in, P, randomReflection, passesFirstFour, startIteration, endIteration, &out, NULL, 0, &ntr);
        if (st == RFX_E_CAP) { cap_b = (out.need_key > out.need_ext ? out.need_key : out.need_ext) + 64; st = RFX_OK; continue; }
        out_n = out.n;
        break;
    }
    if (st == RFX_OK) {                                              /* back to blocks, into the caller's arrays */
        const jsize capK = (*env)->GetArrayLength(env, outKeyBlocks), capE = (*env)->GetArrayLength(env, outExtBlocks);
        int64_t *okb = (int64_t *)malloc((size_t)capK * 8 + 8), *oeb = (int64_t *)malloc((size_t)capE * 8 + 8);
        int64_t *oko = (int64_t *)malloc((size_t)(n + 1) * 8), *oeo = (int64_t *)malloc((size_t)(n + 1) * 8), *oat = (int64_t *)malloc((size_t)(n + 1) * 8);
        int64_t bk = 0, be = 0;
        if (!okb || !oeb || !oko || !oeo || !oat) st = RFX_E_HOST;
        if (st == RFX_OK && ((*env)->GetArrayLength(env, outKeyOff) < out_n + 1 || (*env)->GetArrayLength(env, outExtOff) < out_n + 1 ||
                             (*env)->GetArrayLength(env, outAttr) < out_n))
            st = RFX_E_CAP;
        for (int64_t i = 0; i < out_n && st == RFX_OK; i++) {
            oko[i] = bk; oeo[i] = be;
            const int nk = rfx_dyn_bases_to_blocks(okey + okoff[i], (int)(okoff[i + 1] - okoff[i]), okb + bk, (int)(capK - bk));
            const int ne = rfx_dyn_bases_to_blocks(oext + oeoff[i], (int)(oeoff[i + 1] - oeoff[i]), oeb + be, (int)(capE - be));
            if (nk < 0 || ne < 0) { st = RFX_E_CAP; break; }
            bk += nk; be += ne;
            oat[i] = rfx_dyn_attribute(omk[i], olf[i], ort[i]);
        }
        oko[out_n] = bk; oeo[out_n] = be;
        if (st == RFX_OK) {
            (*env)->SetLongArrayRegion(env, outKeyBlocks, 0, (jsize)bk, (const jlong *)okb);
            (*env)->SetLongArrayRegion(env, outExtBlocks, 0, (jsize)be, (const jlong *)oeb);
            (*env)->SetLongArrayRegion(env, outKeyOff, 0, (jsize)(out_n + 1), (const jlong *)oko);
            (*env)->SetLongArrayRegion(env, outExtOff, 0, (jsize)(out_n + 1), (const jlong *)oeo);
            (*env)->SetLongArrayRegion(env, outAttr, 0, (jsize)out_n, (const jlong *)oat);
        }
        free(okb); free(oeb); free(oko); free(oeo); free(oat);
    }
    free(key); free(ext); free(koff); free(eoff); free(mk); free(lf); free(rt);
    free(okey); free(oext); free(okoff); free(oeoff); free(omk); free(olf); free(ort);
    longs_out(env, keyBlocks, kb, 0); longs_out(env, keyOff, ko, 0); longs_out(env, extBlocks, eb, 0); longs_out(env, extOff, eo, 0);
    longs_out(env, attr, at, 0);
    if (st != RFX_OK) { throw_rfx(env, ctx, st, "rfx_dyn_run"); return 0; }
    return (jlong)out_n;
}
