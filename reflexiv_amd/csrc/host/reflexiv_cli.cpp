// reflexiv_cli.cpp -- `reflexiv_host run|counter -fastq F [-kmer K -cover C ...] -outfile O`
// the two launcher sub-commands of bin/reflexiv:252-271 that reach the hot path
// (M/Main.java:59-79, M/MainOfCounter.java:60-80), on one MI355X instead of spark-submit.
#include <zlib.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <sys/stat.h>

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
        if (argc < 2) { std::cerr << "usage: reflexiv_host <run|counter> -fastq F[,F2...] -outfile DIR [-kmer 31 -cover 2 ...]\n"; return 2; }
        std::string cmd = argv[1];
        std::vector<std::string> args(argv + 2, argv + argc);
        reflexiv::DefaultParam param = reflexiv::importCommandLine(args);
        if (param.inputFqPath.empty() || param.outputPath.empty()) throw std::runtime_error("-fastq and -outfile are required");
        std::string text;
        std::stringstream ss(param.inputFqPath);
        for (std::string one; std::getline(ss, one, ',');) { text += slurp(one); if (!text.empty() && text.back() != '\n') text.push_back('\n'); }
        reflexiv::ReflexivMain m;
        m.setParam(param);
        std::string out;
        if (cmd == "run") out = m.assembly(text);
        else if (cmd == "counter") out = m.counter(text);
        else throw std::runtime_error("unknown command " + cmd);
        mkdir(param.outputPath.c_str(), 0755);
        std::ofstream(param.outputPath + "/part-00000", std::ios::binary) << out;      // saveAsTextFile
        std::ofstream(param.outputPath + "/_SUCCESS", std::ios::binary);
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "reflexiv_host: " << e.what() << "\n";
        return 1;
    }
}
