package uni.bielefeld.cmg.reflexiv.gpu;

import java.util.concurrent.ConcurrentHashMap;

/**
 * Native entry points of libreflexiv_hip.so (include/reflexiv_hip.h) through jni/reflexiv_jni.c: one method per
 * C-ABI entry point of the hot path, each replacing the body of one Spark operator class of
 * pipeline/ReflexivMain.java (cited on the C side).  There is no CPU fallback: without a gfx950 GPU
 * {@link #ctxCreate} throws.
 *
 * A context (one GPU + one HIP stream) is not shared between threads: executor threads ask
 * {@link #ctxForThisTask(int)} and get their own, on GPU partitionId % gpuCount.
 */
public final class Rfx {
    static {
        System.loadLibrary("reflexiv_jni");        // which links libreflexiv_hip.so
    }

    private Rfx() { }

    public static final int TWIN_DS = 0;            // arithmetic of pipeline/ReflexivDSMain.java
    public static final int TWIN_RDD = 1;           // arithmetic of pipeline/ReflexivMain.java

    /** index of each field in the int[13] parameter block (rfx_params, include/reflexiv_hip.h) */
    public static final int P_K = 0, P_MIN_COV = 1, P_MAX_COV = 2, P_MIN_ERROR_COV = 3, P_MIN_CONTIG = 4, P_MIN_ITER = 5,
            P_MAX_ITER = 6, P_FRONT_CLIP = 7, P_END_CLIP = 8, P_PARTITIONS = 9, P_TWIN = 10, P_COALESCE = 11, P_EXTRAS = 12;

    /** operator classes of the k > 31 from-counts extras (ReflexivDSMain64.java:584-619, 672-712) for {@link #extrasOperator} */
    public static final int OP_DOUBLE = 0, OP_EXTENDABLE_PAIRS = 1, OP_UNEXTENDABLE = 2, OP_FIRST_OF_KEY = 3, OP_LONGER_OF_KEY = 4,
            OP_ALL_FORWARD = 5, OP_ALL_REFLECTED = 6;

    private static final ConcurrentHashMap<Long, Long> CTX_OF_THREAD = new ConcurrentHashMap<Long, Long>();
    private static volatile int gpuCount = Integer.getInteger("reflexiv.gpus", 1);

    public static void setGpuCount(int n) { gpuCount = Math.max(1, n); }

    /** the calling thread's context, created on first use on GPU partitionId % gpuCount */
    public static long ctxForThisTask(int partitionId) {
        final long tid = Thread.currentThread().getId();
        Long h = CTX_OF_THREAD.get(tid);
        if (h == null) {
            h = ctxCreate(partitionId % gpuCount);
            CTX_OF_THREAD.put(tid, h);
        }
        return h;
    }

    public static native int version();
    public static native long ctxCreate(int device);
    public static native void ctxDestroy(long ctx);
    public static native int[] defaultParams();

    public static native long[] extractCanon(long ctx, byte[] bases, long[] readOff, int k, int frontClip, int endClip);
    public static native long[] extractCanonW(long ctx, byte[] bases, long[] readOff, int k, int frontClip, int endClip);
    public static native long countFilter(long ctx, long[] kmers, int minCov, int maxCov, int twin, long[] outKeys, int[] outCounts);
    public static native long countFilterW(long ctx, long[] kmers, int k, int minCov, int maxCov, long[] outKeys, long[] outCounts);

    public static native void rcExpandSubkmer(long ctx, long[] kmers, int[] counts, int k, RfxRecords out);
    public static native void sortRecords(long ctx, RfxRecords in, int P, RfxRecords out, long[] partStart);
    public static native void forkFilterForward(long ctx, RfxRecords in, long[] partStart, int k, int minErrorCov, int twin,
                                                RfxRecords out, long[] outPartStart);
    public static native void forkFilterReflected(long ctx, RfxRecords in, long[] partStart, int k, int minErrorCov, int twin,
                                                  RfxRecords out, long[] outPartStart);
    public static native void reflectFromForward(long ctx, RfxRecords in, int k, RfxRecords out);
    public static native void randomReflection(long ctx, RfxRecords in, long[] partStart, int k, RfxRecords out);
    public static native void extendPass(long ctx, RfxRecords in, long[] partStart, int k, int twin, int stage, int scramble,
                                         RfxRecords out, long[] outPartStart);
    public static native void extrasOperator(long ctx, int op, RfxRecords in, long[] partStart, int k, RfxRecords out, long[] outPartStart);
    public static native byte[] contigsText(long ctx, RfxRecords in, int k, int minContig, int twin);

    public static native byte[] assembleReads(long ctx, byte[] bases, long[] readOff, int[] params);

    /** ReflexivDSDynamicKmerDedup.assemblyFromKmer on a run's contig text: every contig once (rfx_dedup_contig_text) */
    public static native byte[] dedupContigText(long ctx, byte[] contigText, int minContig);

    /**
     * The dynamic-k ("meta") passes of ReflexivDSDynamicKmerFirstFour / ReflexivDSDynamicKmerIteration on flattened rows
     * (long[] blocks back to back with block offsets, the packed attribute long per row): randomReflection + passesFirstFour = 4
     * is FirstFour.assemblyFromKmer, startIteration..endIteration is Iteration.assemblyFromKmer.  Returns the number of rows
     * written to the out arrays (rfx_dyn_run).
     */
    public static native long dynRun(long ctx, long[] keyBlocks, long[] keyOff, long[] extBlocks, long[] extOff, long[] attribute, int P,
                                     int randomReflection, int passesFirstFour, int startIteration, int endIteration,
                                     long[] outKeyBlocks, long[] outKeyOff, long[] outExtBlocks, long[] outExtOff, long[] outAttribute);

    // several GPUs of one node: the shuffle of reduceByKey as an RCCL all-to-all inside the library (rfx_comm_*,
    // rfx_sharded_assemble_reads); one barrier task per GPU, see ReflexivGpuMain.assemblyResidentSharded()
    public static native byte[] commUniqueId();
    public static native long commInit(long ctx, byte[] uniqueId, int rank, int world);
    public static native void commDestroy(long comm);
    public static native void commAllReduce(long ctx, long comm, long[] vals, int op);
    // gatherBelow: the extend stage (every sortByKey) stays range-sharded over the GPUs while the record set has more records
    // than this; then rank 0 finishes (-1: the library's default, 0: never gather before the loop ends)
    public static native byte[] shardedAssembleReads(long ctx, long comm, byte[] bases, long[] readOff, int[] params, int generations,
                                                     long gatherBelow, long[] totals);
}
