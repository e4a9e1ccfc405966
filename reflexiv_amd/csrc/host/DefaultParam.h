// DefaultParam.h -- host-side mirror of U/DefaultParam.java (the fields the hot path reads)
// and of the `run` / `counter` flags of U/Parameter.java:302-613.  Same names, same defaults.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace reflexiv {

struct DefaultParam {
    std::string inputFqPath;                 // -fastq
    std::string inputKmerPath;               // -kmerc
    std::string outputPath;                  // -outfile
    int kmerSize = 31;                       // -kmer        DefaultParam.java:74
    int subKmerSize = 30;                    //              :75
    int minKmerCoverage = 2;                 // -cover       :103
    int maxKmerCoverage = 10000000;          // -maxcov      :104
    int minErrorCoverage = 4 * 2;            // -error       :105 (not touched by -cover, Parameter.java:482)
    int minContig = 500;                     // -mincontig   :107
    bool bubble = true;                      // -bubble clears it (Parameter.java:422-424)
    int partitions = 0;                      // -partition   :112
    int maximumIteration = 150;              // -maxiter     :114
    int minimumIteration = 15;               // -miniter     :115
    int frontClip = 0;                       // -clipf       :119
    int endClip = 0;                         // -clipe       :120
    int shufflePartition = 200;              // -partitionredu :122
    // not in the reference: the order contract's logical partition count and the
    // arithmetic twin (0 = P/ReflexivDSMain.java, 1 = P/ReflexivMain.java)
    int logicalPartitions = 8;               // --logical-partitions
    int twin = 1;                            // --twin ds|rdd
    bool resident = false;                   // --resident: one fused call, everything stays in HBM
    int startIteration = 5, endIteration = 9;  // -start / -end of `iteration` (U/DefaultParam.java:118-119)
    bool dedup = false;                      // --dedup: the contig text through rfx_dedup_contig_text (ReflexivDSDynamicKmerDedup)
    int gpus = 1;                            // --gpus N (with --resident): one host thread + context + RCCL communicator per GPU

    void setKmerSize(int k) { kmerSize = k; subKmerSize = k - 1; }   // Parameter.java:345-360
};

// Parameter(args).importCommandLine()  U/Parameter.java:302 -- `-x value` application flags only
inline DefaultParam importCommandLine(const std::vector<std::string> &args) {
    DefaultParam p;
    auto need = [&](size_t i) -> const std::string & {
        if (i + 1 >= args.size()) throw std::runtime_error("Parameter " + args[i] + " needs a value");
        return args[i + 1];
    };
    for (size_t i = 0; i < args.size(); i++) {
        const std::string &a = args[i];
        if (a == "-fastq") p.inputFqPath = need(i++);
        else if (a == "-kmerc") p.inputKmerPath = need(i++);
        else if (a == "-outfile") p.outputPath = need(i++);
        else if (a == "-kmer") {
            int k = std::stoi(need(i++));
            if (k < 1 || k > 100) throw std::runtime_error("Parameter kmer should be set between 1-100");
            p.setKmerSize(k);
        } else if (a == "-cover") p.minKmerCoverage = std::stoi(need(i++));
        else if (a == "-maxcov") p.maxKmerCoverage = std::stoi(need(i++));
        else if (a == "-error") p.minErrorCoverage = std::stoi(need(i++));
        else if (a == "-mincontig") p.minContig = std::stoi(need(i++));
        else if (a == "-miniter") p.minimumIteration = std::stoi(need(i++));
        else if (a == "-maxiter") p.maximumIteration = std::stoi(need(i++));
        else if (a == "-clipf") p.frontClip = std::stoi(need(i++));
        else if (a == "-clipe") p.endClip = std::stoi(need(i++));
        else if (a == "-partition") p.partitions = std::stoi(need(i++));
        else if (a == "-partitionredu") p.shufflePartition = std::stoi(need(i++));
        else if (a == "-bubble") p.bubble = false;
        else if (a == "--logical-partitions") p.logicalPartitions = std::stoi(need(i++));
        else if (a == "--twin") p.twin = need(i++) == "ds" ? 0 : 1;
        else if (a == "--resident") p.resident = true;
        else if (a == "--dedup") p.dedup = true;
        else if (a == "-start") p.startIteration = std::stoi(need(i++));
        else if (a == "-end") p.endIteration = std::stoi(need(i++));
        else if (a == "--gpus") p.gpus = std::max(1, std::stoi(need(i++)));
        else if (a.rfind("--", 0) == 0) { /* spark-submit options are the launcher's (bin/reflexiv:209-235) */ if (i + 1 < args.size() && args[i + 1][0] != '-') i++; }
        else throw std::runtime_error("unknown parameter " + a);
    }
    return p;
}

}  // namespace reflexiv
