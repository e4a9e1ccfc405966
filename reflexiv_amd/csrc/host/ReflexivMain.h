// ReflexivMain.h -- C++ host mirror of P/ReflexivMain.java (the RDD-surface twin) and
// P/ReflexivCounter.java over the C ABI of libreflexiv_hip.so.
//
// The reference is Java on Spark; no JVM exists in the build or test environment, so the host
// side above the C ABI is written here in C++ with the reference's class and method names: one
// nested class per Spark operator, each with a call() that takes the partition's records and
// returns the operator's output, and a driver assembly() that applies them in the order of
// P/ReflexivMain.java:147-316.  Errors surface as exceptions (Java: unchecked exceptions ->
// task failure); there is no CPU path.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/reflexiv_hip.h"
#include "DefaultParam.h"

namespace reflexiv {

struct RfxException : std::runtime_error {
    int status;
    RfxException(int st, const std::string &where, const std::string &detail)
        : std::runtime_error(where + ": status " + std::to_string(st) + (detail.empty() ? "" : " -- " + detail)),
          status(st) {}
};

// JavaPairRDD<Long, Integer> KmerBinaryRDD  (P/ReflexivMain.java:105)
struct KmerBinaryRDD {
    std::vector<uint64_t> kmer;
    std::vector<int32_t> count;
};

// JavaPairRDD<Long, Tuple4<Integer, Long[], Integer, Integer>> (P/ReflexivMain.java:108-109);
// a single-word extension is a one-word array.  partStart = logical partition offsets.
struct ReflexivSubKmerRDD {
    std::vector<uint64_t> key, ext;
    std::vector<int32_t> marker, left, right;
    std::vector<int64_t> extOff, partStart;
    int64_t size() const { return (int64_t)key.size(); }
    rfx_records view();
    void reserve(int64_t n, int64_t words);
    void shrink(const rfx_records &r);
};

class ReflexivMain {
public:
    explicit ReflexivMain(int device = -1);
    ~ReflexivMain();
    void setParam(const DefaultParam &p) { param = p; }          // P/ReflexivMain.java:3126-3128

    // ---- operators (one per inner class; names kept)
    struct FastqFilterWithQual {            // :3089-3113 (+ FastqUnitFilter :3080-3084)
        ReflexivMain &m;
        // text -> concatenated sequence lines + offsets
        void call(const std::string &text, std::vector<uint8_t> &bases, std::vector<int64_t> &readOff) const;
    };
    struct DSFastqFilterOnlySeq {           // P/ReflexivDataFrameCounter.java:238-290 (the counter's line filter)
        ReflexivMain &m;
        void call(const std::string &text, std::vector<uint8_t> &bases, std::vector<int64_t> &readOff) const;
    };
    struct ReverseComplementKmerBinaryExtraction {   // :3002-3075
        ReflexivMain &m;
        std::vector<uint64_t> call(const std::vector<uint8_t> &bases, const std::vector<int64_t> &readOff) const;
    };
    struct KmerCounting_KmerCoverageFilter {         // reduceByKey(:2895-2899) + filter(:3115-3119)
        ReflexivMain &m;
        KmerBinaryRDD call(const std::vector<uint64_t> &kmers) const;
    };
    // ---- k > 31: P/ReflexivDataFrameCounter64.java (W = kmerBinarySlots words per k-mer)
    struct ReverseComplementKmerBinaryExtractionFromDataset64 {   // :390-687
        ReflexivMain &m;
        std::vector<uint64_t> call(const std::vector<uint8_t> &bases, const std::vector<int64_t> &readOff) const;
    };
    struct KmerBlocksCount {                         // groupBy("kmerBlocks").count() + filters :191-205
        ReflexivMain &m;
        // kmers: W words per k-mer; -> rows (W words, count) ascending
        void call(const std::vector<uint64_t> &kmers, std::vector<uint64_t> &keys, std::vector<int64_t> &counts) const;
    };
    struct DSBinaryKmerToString {                    // :335-384
        ReflexivMain &m;
        std::string call(const uint64_t *kmerBlocks) const;
    };
    struct KmerReverseComplement_ForwardSubKmerExtraction {   // :2901-2931 + :2703-2731
        ReflexivMain &m;
        ReflexivSubKmerRDD call(const KmerBinaryRDD &in) const;
    };
    struct SortByKey {                               // sortByKey() :179,191,211,235,247,286
        ReflexivMain &m;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in, int P) const;
    };
    struct FilterForkSubKmer {                       // :2406-2541 (with or without error correction)
        ReflexivMain &m;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in) const;
    };
    struct ReflectedSubKmerExtractionFromForward {   // :2734-2769
        ReflexivMain &m;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in) const;
    };
    struct FilterForkReflectedSubKmer {              // :2543-2697
        ReflexivMain &m;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in) const;
    };
    struct kmerRandomReflection {                    // :2774-2886
        ReflexivMain &m;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in) const;
    };
    struct ExtendReflexivKmer {                      // :2019-2401, :1564-2013, :762-1558 by stage
        ReflexivMain &m; int stage;
        ReflexivSubKmerRDD call(ReflexivSubKmerRDD &in) const;
    };
    struct KmerToContig {                            // :693-758 + :588-638 + :571-582
        ReflexivMain &m;
        std::string call(ReflexivSubKmerRDD &in, int64_t *nContigs) const;
    };

    // ---- drivers
    // assembly(): P/ReflexivMain.java:95-322 from FASTQ text to the contig text of saveAsTextFile
    std::string assembly(const std::string &fastqText, std::vector<int64_t> *trace = nullptr);
    // the same result through ONE call (rfx_assemble_reads): reads go up, contig text comes back
    std::string assemblyResident(const std::string &fastqText, std::vector<int64_t> *trace = nullptr);
    // the same on the nGpus GPUs of this node: one host thread, context and RCCL communicator per GPU, every thread passes its
    // share of the reads to rfx_sharded_assemble_reads (the shuffle of reduceByKey, P/ReflexivMain.java:155, over RCCL)
    // P/ReflexivDSDynamicKmerDedup.java assemblyFromKmer (:138-339) on the contig text of a run: each contig once
    std::string dedupContigText(const std::string &contigText);
    // gatherBelow: the extend stage stays range-sharded over the GPUs (sortByKey as an RCCL all-to-all) while the record set has
    // more records than this (-1: the library's default; rfx_dev_sharded_assemble)
    std::string assemblyResidentSharded(const std::string &fastqText, int nGpus, std::vector<int64_t> *trace = nullptr, int64_t gatherBelow = -1);
    // ReflexivCounter.assembly(): P/ReflexivCounter.java:109-191 -> lines "KMER,count"
    std::string counter(const std::string &fastqText);
    std::string assemblyFromCounts(const KmerBinaryRDD &counts, std::vector<int64_t> *trace = nullptr);
    // assemblyFromKmer(): P/ReflexivMain.java:327-568 / P/ReflexivDSMain.java:362-712 -- `run -kmerc`:
    // CSV rows "KMER,count" (also the legacy "(KMER,count)" tuple text) as written by `counter`
    struct KmerBinarizer {                           // P/ReflexivDSMain.java:3871-3947
        ReflexivMain &m;
        KmerBinaryRDD call(const std::string &csvText) const;
    };
    std::string assemblyFromKmer(const std::string &csvText, std::vector<int64_t> *trace = nullptr);
    // k > 31: ReflexivDSMain64.assemblyFromKmer (P/ReflexivDSMain64.java:374-826) -- `run -kmerc COUNTS -kmer 63`.
    // KmerBinarizer :10772-10836 writes kmerBinarySlotsAssemble = (k-1)/31+1 words of 31 bases per k-mer.
    struct KmerBinarizer64 {
        ReflexivMain &m;
        void call(const std::string &csvText, std::vector<uint64_t> &kmers, std::vector<int32_t> &counts) const;
    };
    std::string assemblyFromKmer64(const std::string &csvText, std::vector<int64_t> *trace = nullptr);
    // The dynamic-k ("meta") passes.  ReflexivDSDynamicKmerFirstFour.assemblyFromKmer
    // (P/ReflexivDSDynamicKmerFirstFour.java:137-224): CSV rows "KMER,marker|left|right" -> DynamicKmerBinarizerFromReducedToSubKmer
    // (:2931-3016) -> DSkmerRandomReflection -> 4 x (sort, DSExtendReflexivKmer) -> DSBinarySubKmerWithShortExtensionToString
    // (:226-263) rows "SUBKMER,marker|left|right,EXTENSION".  ReflexivDSDynamicKmerIteration.assemblyFromKmer
    // (P/ReflexivDSDynamicKmerIteration.java:134-205): those rows -> (end - start + 1) x (sort, DSExtendReflexivKmerToArrayLoop).
    std::string assemblyDynamicFirstFour(const std::string &csvText, std::vector<int64_t> *trace = nullptr);
    std::string assemblyDynamicIteration(const std::string &csvText, int startIteration, int endIteration,
                                         std::vector<int64_t> *trace = nullptr);

    rfx_ctx *ctx = nullptr;
    DefaultParam param;
    void check(int st, const char *where) const;
};

}  // namespace reflexiv
