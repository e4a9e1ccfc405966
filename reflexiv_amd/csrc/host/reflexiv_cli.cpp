// reflexiv_cli.cpp -- `reflexiv_host run|counter -fastq F [-kmer K -cover C ...] -outfile O`
// the two launcher sub-commands of bin/reflexiv:252-271 that reach the hot path
// (M/Main.java:59-79, M/MainOfCounter.java:60-80), on one MI355X instead of spark-submit.
#include <zlib.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <sys/stat.h>
#include <dirent.h>
#include <cstring>
#include <algorithm>

#include "ReflexivMain.h"

static std::string slurp(const std::string &path) {
    std::string out;
    gzFile f = gzopen(path.c_str(), "rb");          // reads plain text as well as .gz
    if (!f) throw std::runtime_error("cannot open " + path);
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
    gzclose(f);
    return out;
}

int main(int argc, char **argv) {
    try {
        if (argc < 2) { std::cerr << "usage: reflexiv_host <run|counter|firstfour|iteration> -fastq F[,F2...] -outfile DIR [-kmer 31 -cover 2 ...]\n"; return 2; }
        std::string cmd = argv[1];
        std::vector<std::string> args(argv + 2, argv + argc);
        reflexiv::DefaultParam param = reflexiv::importCommandLine(args);
        if (param.outputPath.empty()) throw std::runtime_error("-outfile is required");
        if (param.inputFqPath.empty() && param.inputKmerPath.empty()) throw std::runtime_error("-fastq or -kmerc is required");
        auto read_all = [&](const std::string &paths) {
            std::string text;
            std::stringstream ss(paths);
            for (std::string one; std::getline(ss, one, ',');) {
                struct stat st;
                if (stat(one.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) {      // a Spark output directory: every part-*
                    DIR *d = opendir(one.c_str());
                    std::vector<std::string> parts;
                    for (dirent *e; d && (e = readdir(d));) if (strncmp(e->d_name, "part-", 5) == 0) parts.push_back(one + "/" + e->d_name);
                    if (d) closedir(d);
                    std::sort(parts.begin(), parts.end());
                    for (auto &f : parts) text += slurp(f);
                } else {
                    text += slurp(one);
                }
                if (!text.empty() && text.back() != '\n') text.push_back('\n');
            }
            return text;
        };
        reflexiv::ReflexivMain m;
        m.setParam(param);
        std::string out, dir = param.outputPath;
        mkdir(dir.c_str(), 0755);
        if (cmd == "run") {
            // M/Main.java:74-78 -> Pipelines.reflexivDSMainPipe(): -kmerc routes to assemblyFromKmer()
            // k > 31: Pipelines.reflexivDSMainPipe64() -> ReflexivDSMain64.assemblyFromKmer(), output under Assemble_<k>
            // (P/ReflexivDSMain64.java:820-824); only the from-counts route exists for k > 31
            if (param.kmerSize > 31) {
                if (param.inputKmerPath.empty()) throw std::runtime_error("-kmer > 31 needs -kmerc (counter -> run -kmerc; the reference's run -fastq is inconsistent for k > 31)");
                out = m.assemblyFromKmer64(read_all(param.inputKmerPath));
                dir += "/Assemble_" + std::to_string(param.kmerSize);
                mkdir(dir.c_str(), 0755);
            } else
            out = !param.inputKmerPath.empty() ? m.assemblyFromKmer(read_all(param.inputKmerPath))
                  : param.resident && (param.gpus > 1 || getenv("RFX_HOST_FORCE_SHARDED")) ? m.assemblyResidentSharded(read_all(param.inputFqPath), param.gpus)
                  : param.resident          ? m.assemblyResident(read_all(param.inputFqPath))
                                            : m.assembly(read_all(param.inputFqPath));
            if (param.dedup) out = m.dedupContigText(out);
        } else if (cmd == "firstfour") {
            // Pipelines.reflexivDSDynamicKmerFirstFourPipe(): rows "KMER,marker|left|right" of the reduction -> 00firstFour
            out = m.assemblyDynamicFirstFour(read_all(param.inputKmerPath));
            dir += "/Assembly_intermediate"; mkdir(dir.c_str(), 0755);
            dir += "/00firstFour"; mkdir(dir.c_str(), 0755);
        } else if (cmd == "iteration") {
            // Pipelines.reflexivDSDynamicKmerIterationPipe(): -> 01Iteration<start>_<end>
            out = m.assemblyDynamicIteration(read_all(param.inputKmerPath), param.startIteration, param.endIteration);
            dir += "/Assembly_intermediate"; mkdir(dir.c_str(), 0755);
            dir += "/01Iteration" + std::to_string(param.startIteration) + "_" + std::to_string(param.endIteration); mkdir(dir.c_str(), 0755);
        } else if (cmd == "counter") {
            out = m.counter(read_all(param.inputFqPath));
            dir += "/Count_" + std::to_string(param.kmerSize);               // P/ReflexivDataFrameCounter.java:222-233
            mkdir(dir.c_str(), 0755);
        } else throw std::runtime_error("unknown command " + cmd);
        std::ofstream(dir + (cmd == "counter" || cmd == "firstfour" || cmd == "iteration" ? "/part-00000.csv" : "/part-00000"), std::ios::binary) << out;   // saveAsTextFile / csv
        std::ofstream(dir + "/_SUCCESS", std::ios::binary);
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "reflexiv_host: " << e.what() << "\n";
        return 1;
    }
}
