package uni.bielefeld.cmg.reflexiv.pipeline;

import org.apache.spark.SparkConf;
import org.apache.spark.TaskContext;
import org.apache.spark.api.java.JavaPairRDD;
import org.apache.spark.api.java.JavaRDD;
import org.apache.spark.api.java.JavaSparkContext;
import org.apache.spark.api.java.function.FlatMapFunction;
import org.apache.spark.api.java.function.Function;
import org.apache.spark.api.java.function.Function2;
import org.apache.spark.api.java.function.PairFlatMapFunction;
import scala.Tuple2;
import scala.Tuple4;
import uni.bielefeld.cmg.reflexiv.gpu.Rfx;
import uni.bielefeld.cmg.reflexiv.gpu.RfxRecords;
import uni.bielefeld.cmg.reflexiv.util.DefaultParam;

import java.io.ByteArrayOutputStream;
import java.io.Serializable;
import java.nio.charset.StandardCharsets;
import java.util.ArrayList;
import java.util.Iterator;
import java.util.List;

/**
 * The fixed-k (k <= 31) assembler with the RDD operator surface of {@link ReflexivMain#assembly()}
 * (pipeline/ReflexivMain.java:95-322) and every operator body on the MI355X: the driver below keeps that
 * method's RDD types, operator order, iteration counter and stop rule; each inner class keeps the name and the
 * Spark interface of the class it stands in for, and its call() packs the partition into flat arrays, makes ONE
 * call into libreflexiv_hip.so ({@link Rfx}) and unpacks the result.  Spark still moves the records between
 * operators (sortByKey, reduceByKey), so this class drops into bin/reflexiv by class name.
 *
 * {@link #assemblyResident()} is the other way to use the library: reads go up once, the k-mer count, every sort
 * and every extend pass stay in HBM (RCCL all-to-all instead of the Spark shuffle on several GPUs), and only the
 * contig text comes back.
 *
 * Not compiled in the build container (no JDK, no Spark jars); tests/test_jni_sources.py checks its native
 * calls against Rfx.java, jni/reflexiv_jni.c and include/reflexiv_hip.h.
 */
public class ReflexivGpuMain implements Serializable {
    private DefaultParam param;
    /** which arithmetic twin the operators follow: the operator surface is ReflexivMain's, so RDD by default */
    private int twin = Rfx.TWIN_RDD;

    public void setParam(DefaultParam param) { this.param = param; }
    public void setTwin(int twin) { this.twin = twin; }

    private SparkConf setSparkConfiguration() {
        SparkConf conf = new SparkConf().setAppName("Reflexiv (MI355X operators)");
        conf.set("spark.kryo.registrator", "uni.bielefeld.cmg.reflexiv.serializer.SparkKryoRegistrator");
        return conf;
    }

    private static long ctx() { return Rfx.ctxForThisTask(TaskContext.getPartitionId()); }
    private static long[] onePartition(long n) { return new long[]{0L, n}; }

    // ------------------------------------------------------------------------------------------ driver

    public void assembly() {
        JavaSparkContext sc = new JavaSparkContext(setSparkConfiguration());

        JavaRDD<String> FastqRDD = sc.textFile(param.inputFqPath);
        JavaPairRDD<Long, Integer> KmerBinaryRDD;
        JavaPairRDD<Long, Tuple4<Integer, Long, Integer, Integer>> ReflexivSubKmerRDD;
        JavaPairRDD<Long, Tuple4<Integer, Long[], Integer, Integer>> ReflexivLongSubKmerRDD;

        // FASTQ grouping and the null filter stay host code (ReflexivMain.java:120-133); they are string handling
        FastqRDD = FastqRDD.map(new FastqFilterWithQual()).filter(new FastqUnitFilter());
        if (param.partitions > 0) FastqRDD = FastqRDD.repartition(param.partitions);
        if (param.cache) FastqRDD.cache();

        // k-mer extraction, count, coverage filter (:147-163)
        KmerBinaryRDD = FastqRDD.mapPartitionsToPair(new ReverseComplementKmerBinaryExtraction());
        KmerBinaryRDD = KmerBinaryRDD.reduceByKey(new KmerCounting());
        if (param.minKmerCoverage > 1) KmerBinaryRDD = KmerBinaryRDD.filter(new KmerCoverageFilter());

        // both strands, forward sub-k-mers (:168-176)
        ReflexivSubKmerRDD = KmerBinaryRDD.mapPartitionsToPair(new KmerReverseComplementAndForwardSubKmerExtraction());

        if (param.bubble) {                                                     // fork filters (:178-199)
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.sortByKey();
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(new FilterForkSubKmer());
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(new ReflectedSubKmerExtractionFromForward());
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.sortByKey();
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(new FilterForkReflectedSubKmer());
        }

        ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(new kmerRandomReflection());     // :204-205
        ReflexivSubKmerRDD = ReflexivSubKmerRDD.sortByKey();                                         // :211

        ExtendReflexivKmer KmerExtention = new ExtendReflexivKmer();
        ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(KmerExtention);                  // :221-222

        int iterations = 0;
        for (int i = 1; i < 4; i++) {                                                                // :232-241
            iterations++;
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.sortByKey();
            ReflexivSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(KmerExtention);
        }

        ReflexivSubKmerRDD = ReflexivSubKmerRDD.sortByKey();                                         // :247
        iterations++;
        ReflexivLongSubKmerRDD = ReflexivSubKmerRDD.mapPartitionsToPair(new ExtendReflexivKmerToArrayFirstTime());

        ExtendReflexivKmerToArrayLoop KmerExtenstionArrayToArray = new ExtendReflexivKmerToArrayLoop();
        int partitionNumber = ReflexivLongSubKmerRDD.getNumPartitions();
        long contigNumber = 0;
        while (iterations <= param.maximumIteration) {                                               // :265-296
            iterations++;
            if (iterations >= param.minimumIteration && iterations % 3 == 0) {
                long currentContigNumber = ReflexivLongSubKmerRDD.count();
                if (contigNumber == currentContigNumber) break;
                contigNumber = currentContigNumber;
                if (partitionNumber >= 16 && currentContigNumber / partitionNumber <= 20) {
                    partitionNumber = partitionNumber / 4 + 1;
                    ReflexivLongSubKmerRDD = ReflexivLongSubKmerRDD.coalesce(partitionNumber);
                }
            }
            ReflexivLongSubKmerRDD = ReflexivLongSubKmerRDD.sortByKey();
            ReflexivLongSubKmerRDD = ReflexivLongSubKmerRDD.mapPartitionsToPair(KmerExtenstionArrayToArray);
        }

        // records -> contig text (:302-310); ids come from zipWithIndex as in the reference
        JavaPairRDD<String, String> ContigTuple2RDD = ReflexivLongSubKmerRDD.mapPartitionsToPair(new KmerToContig());
        JavaRDD<String> ContigRDD = ContigTuple2RDD.zipWithIndex().flatMap(new TagContigID());
        ContigRDD.saveAsTextFile(param.outputPath);
        sc.stop();
    }

    /**
     * Resident form: the whole path in one native call per GPU (rfx_assemble_reads).  The reads of the job are
     * collected to the driver's GPU; for read sets beyond one GPU see {@link #assemblyResidentSharded(int)}.
     */
    public void assemblyResident() {
        JavaSparkContext sc = new JavaSparkContext(setSparkConfiguration());
        JavaRDD<String> FastqRDD = sc.textFile(param.inputFqPath).map(new FastqFilterWithQual()).filter(new FastqUnitFilter());
        List<String> units = FastqRDD.collect();
        ByteArrayOutputStream bases = new ByteArrayOutputStream();
        long[] readOff = new long[units.size() + 1];
        for (int i = 0; i < units.size(); i++) {
            byte[] seq = units.get(i).split("\\n")[1].getBytes(StandardCharsets.US_ASCII);
            bases.write(seq, 0, seq.length);
            readOff[i + 1] = bases.size();
        }
        int[] prm = Rfx.defaultParams();
        prm[Rfx.P_K] = param.kmerSize; prm[Rfx.P_MIN_COV] = param.minKmerCoverage; prm[Rfx.P_MAX_COV] = param.maxKmerCoverage;
        prm[Rfx.P_MIN_ERROR_COV] = param.minErrorCoverage; prm[Rfx.P_MIN_CONTIG] = param.minContig;
        prm[Rfx.P_MIN_ITER] = param.minimumIteration; prm[Rfx.P_MAX_ITER] = param.maximumIteration;
        prm[Rfx.P_FRONT_CLIP] = param.frontClip; prm[Rfx.P_END_CLIP] = param.endClip;
        prm[Rfx.P_PARTITIONS] = param.partitions > 0 ? param.partitions : 8; prm[Rfx.P_TWIN] = twin;
        byte[] text = Rfx.assembleReads(Rfx.ctxForThisTask(0), bases.toByteArray(), readOff, prm);
        List<String> one = new ArrayList<String>();
        String s = new String(text, StandardCharsets.US_ASCII);
        one.add(s.endsWith("\n") ? s.substring(0, s.length() - 1) : s);
        sc.parallelize(one, 1).saveAsTextFile(param.outputPath);
        sc.stop();
    }

    /**
     * Resident form on the nGpus GPUs of one node: a Spark BARRIER stage of nGpus tasks, task r on GPU r.  Every task packs
     * ITS partition of the reads and calls rfx_sharded_assemble_reads: the k-mer space is radix-sharded over the GPUs by
     * the owner of each k-mer's minimiser, super-k-mer records cross an RCCL all-to-all inside the library (the shuffle of
     * reduceByKey, ReflexivMain.java:155), every GPU counts and filters its shard, the shards are gathered on task 0 and
     * extended there.  Task 0 makes the RCCL id; BarrierTaskContext.allGather hands it round.
     */
    public void assemblyResidentSharded(final int nGpus) {
        JavaSparkContext sc = new JavaSparkContext(setSparkConfiguration());
        JavaRDD<String> FastqRDD = sc.textFile(param.inputFqPath).map(new FastqFilterWithQual()).filter(new FastqUnitFilter())
                .repartition(nGpus);
        final int[] prm = Rfx.defaultParams();
        prm[Rfx.P_K] = param.kmerSize; prm[Rfx.P_MIN_COV] = param.minKmerCoverage; prm[Rfx.P_MAX_COV] = param.maxKmerCoverage;
        prm[Rfx.P_MIN_ERROR_COV] = param.minErrorCoverage; prm[Rfx.P_MIN_CONTIG] = param.minContig;
        prm[Rfx.P_MIN_ITER] = param.minimumIteration; prm[Rfx.P_MAX_ITER] = param.maximumIteration;
        prm[Rfx.P_FRONT_CLIP] = param.frontClip; prm[Rfx.P_END_CLIP] = param.endClip;
        prm[Rfx.P_PARTITIONS] = param.partitions > 0 ? param.partitions : 8; prm[Rfx.P_TWIN] = twin;
        // (JavaRDD has no barrier(): the barrier stage is built on the underlying RDD)
        scala.reflect.ClassTag<String> tag = scala.reflect.ClassTag$.MODULE$.apply(String.class);
        JavaRDD<String> ContigRDD = JavaRDD.fromRDD(
                FastqRDD.rdd().barrier().mapPartitions(new ShardedResident(prm, nGpus), false, tag), tag);
        ContigRDD.filter(new NonEmpty()).coalesce(1).saveAsTextFile(param.outputPath);
        sc.stop();
    }

    static class NonEmpty implements Function<String, Boolean>, Serializable {
        public Boolean call(String s) { return s != null && !s.isEmpty(); }
    }

    static class ShardedResident extends scala.runtime.AbstractFunction1<scala.collection.Iterator<String>, scala.collection.Iterator<String>>
            implements Serializable {
        private final int[] prm;
        private final int nGpus;
        ShardedResident(int[] prm, int nGpus) { this.prm = prm; this.nGpus = nGpus; }

        public scala.collection.Iterator<String> apply(scala.collection.Iterator<String> it) {
            return scala.collection.JavaConverters.asScalaIteratorConverter(call(scala.collection.JavaConverters.asJavaIteratorConverter(it).asJava())).asScala();
        }

        public Iterator<String> call(Iterator<String> units) {
            org.apache.spark.BarrierTaskContext tc = org.apache.spark.BarrierTaskContext.get();
            final int rank = tc.partitionId();
            ByteArrayOutputStream bases = new ByteArrayOutputStream();
            List<Long> off = new ArrayList<Long>();
            off.add(0L);
            while (units.hasNext()) {
                byte[] seq = units.next().split("\\n")[1].getBytes(StandardCharsets.US_ASCII);
                bases.write(seq, 0, seq.length);
                off.add((long) bases.size());
            }
            long[] readOff = new long[off.size()];
            for (int i = 0; i < readOff.length; i++) readOff[i] = off.get(i);
            final long ctx = Rfx.ctxCreate(rank % nGpus);
            String mine = rank == 0 ? java.util.Base64.getEncoder().encodeToString(Rfx.commUniqueId()) : "";
            String[] all = tc.allGather(mine);                                   // the 128-byte RCCL id from task 0
            final long comm = Rfx.commInit(ctx, java.util.Base64.getDecoder().decode(all[0]), rank, nGpus);
            List<String> out = new ArrayList<String>();
            try {
                byte[] text = Rfx.shardedAssembleReads(ctx, comm, bases.toByteArray(), readOff, prm, 4, -1L, new long[3]);
                String s = new String(text, StandardCharsets.US_ASCII);
                out.add(s.endsWith("\n") ? s.substring(0, s.length() - 1) : s);
            } finally {
                Rfx.commDestroy(comm);
                Rfx.ctxDestroy(ctx);
            }
            return out.iterator();
        }
    }

    // --------------------------------------------------------------------------------------- operators

    /** ReflexivMain.java:3089-3113 (host string handling, unchanged in behaviour) */
    class FastqFilterWithQual implements Function<String, String>, Serializable {
        String line = "";
        int lineMark = 0;
        public String call(String s) {
            if (lineMark == 2) { lineMark++; line = line + "\n" + s; return null; }
            else if (lineMark == 3) { lineMark++; line = line + "\n" + s; return line; }
            else if (s.startsWith("@")) { line = s; lineMark = 1; return null; }
            else if (lineMark == 1) { line = line + "\n" + s; lineMark++; return null; }
            else return null;
        }
    }

    /** ReflexivMain.java:3080-3084 */
    class FastqUnitFilter implements Function<String, Boolean>, Serializable {
        public Boolean call(String s) { return s != null; }
    }

    /** ReflexivMain.java:3002-3075 -> rfx_extract_canon */
    class ReverseComplementKmerBinaryExtraction implements PairFlatMapFunction<Iterator<String>, Long, Integer>, Serializable {
        public Iterator<Tuple2<Long, Integer>> call(Iterator<String> s) {
            ByteArrayOutputStream bases = new ByteArrayOutputStream();
            List<Long> off = new ArrayList<Long>();
            off.add(0L);
            while (s.hasNext()) {
                byte[] seq = s.next().split("\\n")[1].getBytes(StandardCharsets.US_ASCII);
                bases.write(seq, 0, seq.length);
                off.add((long) bases.size());
            }
            long[] readOff = new long[off.size()];
            for (int i = 0; i < readOff.length; i++) readOff[i] = off.get(i);
            final long[] kmers = Rfx.extractCanon(ctx(), bases.toByteArray(), readOff, param.kmerSize, param.frontClip, param.endClip);
            return new Iterator<Tuple2<Long, Integer>>() {
                int i = 0;
                public boolean hasNext() { return i < kmers.length; }
                public Tuple2<Long, Integer> next() { return new Tuple2<Long, Integer>(kmers[i++], 1); }
                public void remove() { throw new UnsupportedOperationException(); }
            };
        }
    }

    /** ReflexivMain.java:2895-2899: the combine function stays a Java lambda-sized class; Spark calls it per pair.
     *  (The fused alternative -- partitionBy(owner) then one rfx_count_filter per partition -- is in INTEGRATION.md.) */
    class KmerCounting implements Function2<Integer, Integer, Integer>, Serializable {
        public Integer call(Integer i1, Integer i2) { return i1 + i2; }
    }

    /** ReflexivMain.java:3115-3119 */
    class KmerCoverageFilter implements Function<Tuple2<Long, Integer>, Boolean>, Serializable {
        public Boolean call(Tuple2<Long, Integer> s) { return s._2 >= param.minKmerCoverage && s._2 <= param.maxKmerCoverage; }
    }

    /** KmerReverseComplement (:2910-2930) + ForwardSubKmerExtraction (:2709-2730) -> rfx_rc_expand_subkmer */
    class KmerReverseComplementAndForwardSubKmerExtraction
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Integer>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Integer>> s) {
            List<Tuple2<Long, Integer>> l = new ArrayList<Tuple2<Long, Integer>>();
            while (s.hasNext()) l.add(s.next());
            long[] kmers = new long[l.size()];
            int[] counts = new int[l.size()];
            for (int i = 0; i < kmers.length; i++) { kmers[i] = l.get(i)._1; counts[i] = l.get(i)._2; }
            RfxRecords out = new RfxRecords(2 * kmers.length, 2 * kmers.length, 1);
            Rfx.rcExpandSubkmer(ctx(), kmers, counts, param.kmerSize, out);
            return out.toSingle();
        }
    }

    /** FilterForkSubKmer / FilterForkSubKmerWithErrorCorrection (:2412-2540; minErrorCoverage == 0 selects the first)
     *  -> rfx_fork_filter_forward */
    class FilterForkSubKmer
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.forkFilterForward(ctx(), in, onePartition(in.n), param.kmerSize, param.minErrorCoverage, twin, out, new long[2]);
            return out.toSingle();
        }
    }

    /** ReflectedSubKmerExtractionFromForward (:2742-2768) -> rfx_reflect_from_forward */
    class ReflectedSubKmerExtractionFromForward
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.reflectFromForward(ctx(), in, param.kmerSize, out);
            return out.toSingle();
        }
    }

    /** FilterForkReflectedSubKmer[WithErrorCorrection] (:2550-2696) -> rfx_fork_filter_reflected */
    class FilterForkReflectedSubKmer
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.forkFilterReflected(ctx(), in, onePartition(in.n), param.kmerSize, param.minErrorCoverage, twin, out, new long[2]);
            return out.toSingle();
        }
    }

    /** kmerRandomReflection (:2783-2885) -> rfx_random_reflection */
    class kmerRandomReflection
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.randomReflection(ctx(), in, onePartition(in.n), param.kmerSize, out);
            return out.toSingle();
        }
    }

    /** ExtendReflexivKmer (:2019-2401) -> rfx_extend_pass, stage 0 */
    class ExtendReflexivKmer
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long, Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.extendPass(ctx(), in, onePartition(in.n), param.kmerSize, twin, 0, 2, out, new long[2]);
            return out.toSingle();
        }
    }

    /** ExtendReflexivKmerToArrayFirstTime (:1564-2013) -> rfx_extend_pass, stage 1 (outputs may take two words) */
    class ExtendReflexivKmerToArrayFirstTime
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>, Long, Tuple4<Integer, Long[], Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromSingle(s);
            RfxRecords out = new RfxRecords((int) in.n, (int) in.n, 1);
            Rfx.extendPass(ctx(), in, onePartition(in.n), param.kmerSize, twin, 1, 2, out, new long[2]);
            return out.toArray();
        }
    }

    /** ExtendReflexivKmerToArrayLoop (:762-1558) -> rfx_extend_pass, stage 2 */
    class ExtendReflexivKmerToArrayLoop
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>>, Long, Tuple4<Integer, Long[], Integer, Integer>>, Serializable {
        public Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromArray(s);
            RfxRecords out = new RfxRecords((int) in.n, in.words(), 1);
            Rfx.extendPass(ctx(), in, onePartition(in.n), param.kmerSize, twin, 2, 2, out, new long[2]);
            return out.toArray();
        }
    }

    /** BinaryReflexivKmerArrayToString (:693-758) + KmerToContig (:588-638) -> rfx_contigs_text; emits (header
     *  without the running id, folded sequence) pairs as the reference's KmerToContig does, so that zipWithIndex +
     *  TagContigID number them across partitions */
    class KmerToContig
            implements PairFlatMapFunction<Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>>, String, String>, Serializable {
        public Iterator<Tuple2<String, String>> call(Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> s) {
            RfxRecords in = RfxRecords.fromArray(s);
            String text = new String(Rfx.contigsText(ctx(), in, param.kmerSize, param.minContig, twin), StandardCharsets.US_ASCII);
            List<Tuple2<String, String>> out = new ArrayList<Tuple2<String, String>>();
            for (String block : text.split(">")) {
                if (block.isEmpty()) continue;
                int nl = block.indexOf('\n');
                String header = block.substring(0, nl);                 // Contig-<len>-<local id>
                String body = block.substring(nl + 1, block.endsWith("\n") ? block.length() - 1 : block.length());
                out.add(new Tuple2<String, String>(">" + header.substring(0, header.lastIndexOf('-')), body));
            }
            return out.iterator();
        }
    }

    /** ReflexivMain.java:571-582 */
    class TagContigID implements FlatMapFunction<Tuple2<Tuple2<String, String>, Long>, String>, Serializable {
        public Iterator<String> call(Tuple2<Tuple2<String, String>, Long> s) {
            List<String> l = new ArrayList<String>();
            l.add(s._1._1 + "-" + s._2 + "\n" + s._1._2);
            return l.iterator();
        }
    }
}
