// ReflexivMain.cpp -- see ReflexivMain.h.  Every call() forwards to one C-ABI entry point.
#include <thread>
#include "ReflexivMain.h"

#include <algorithm>
#include <cstring>

namespace reflexiv {

rfx_records ReflexivSubKmerRDD::view() {
    rfx_records r{};
    r.n = size();
    r.key = key.data(); r.marker = marker.data(); r.ext_off = extOff.data(); r.ext = ext.data();
    r.left = left.data(); r.right = right.data();
    r.cap_n = (int64_t)key.size(); r.cap_words = (int64_t)ext.size();
    return r;
}
void ReflexivSubKmerRDD::reserve(int64_t n, int64_t words) {
    n = std::max<int64_t>(n, 1); words = std::max<int64_t>(words, 1);
    key.resize(n); marker.resize(n); left.resize(n); right.resize(n); extOff.resize(n + 1); ext.resize(words);
}
void ReflexivSubKmerRDD::shrink(const rfx_records &r) {
    key.resize(r.n); marker.resize(r.n); left.resize(r.n); right.resize(r.n); extOff.resize(r.n + 1);
    ext.resize(r.n ? extOff[r.n] : 0);
}

ReflexivMain::ReflexivMain(int device) {
    int st = rfx_ctx_create(device, &ctx);
    if (st != RFX_OK) throw RfxException(st, "rfx_ctx_create", "an MI355X (gfx950) is required; there is no CPU path");
}
ReflexivMain::~ReflexivMain() { rfx_ctx_destroy(ctx); }

void ReflexivMain::check(int st, const char *where) const {
    if (st != RFX_OK) throw RfxException(st, where, rfx_last_error(ctx));
}

// FastqFilterWithQual.call + FastqUnitFilter.call  P/ReflexivMain.java:3080-3113: the 4-line state
// machine; emits the sequence line of every completed record.
void ReflexivMain::FastqFilterWithQual::call(const std::string &text, std::vector<uint8_t> &bases,
                                             std::vector<int64_t> &readOff) const {
    int lineMark = 0;
    size_t pos = 0, seqOff = 0, seqLen = 0;
    bases.clear(); readOff.assign(1, 0);
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        size_t l = e - pos;
        if (l > 0 && text[e - 1] == '\r') l--;
        if (lineMark == 2) lineMark++;                                  // :3093-3096
        else if (lineMark == 3) {                                       // :3097-3100
            lineMark++;
            bases.insert(bases.end(), text.begin() + seqOff, text.begin() + seqOff + seqLen);
            readOff.push_back((int64_t)bases.size());
        } else if (l > 0 && text[pos] == '@') lineMark = 1;             // :3101-3104
        else if (lineMark == 1) { seqOff = pos; seqLen = l; lineMark++; }   // :3105-3108
        pos = e + 1;
    }
}

void ReflexivMain::DSFastqFilterOnlySeq::call(const std::string &text, std::vector<uint8_t> &bases,
                                              std::vector<int64_t> &readOff) const {
    auto checkSeq = [](char a) { return a == 'A' || a == 'T' || a == 'C' || a == 'G' || a == 'N'; };   // :268-289
    size_t pos = 0;
    bases.clear(); readOff.assign(1, 0);
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        size_t l = e - pos;
        if (l > 0 && text[e - 1] == '\r') l--;
        const char *s = text.data() + pos;
        if (l > 20 && s[0] != '@' && s[0] != '+' && checkSeq(s[0]) && checkSeq(s[4]) && checkSeq(s[9]) &&
            checkSeq(s[14]) && checkSeq(s[19])) {                      // :245-262
            bases.insert(bases.end(), text.begin() + pos, text.begin() + pos + l);
            readOff.push_back((int64_t)bases.size());
        }
        pos = e + 1;
    }
}

std::vector<uint64_t> ReflexivMain::ReverseComplementKmerBinaryExtraction::call(
    const std::vector<uint8_t> &bases, const std::vector<int64_t> &readOff) const {
    int64_t n = 0;
    const int64_t nr = (int64_t)readOff.size() - 1;
    int st = rfx_extract_canon(m.ctx, bases.data(), readOff.data(), nr, m.param.kmerSize, m.param.frontClip,
                               m.param.endClip, nullptr, 0, &n);
    if (st != RFX_OK && st != RFX_E_CAP) m.check(st, "rfx_extract_canon");
    std::vector<uint64_t> out((size_t)std::max<int64_t>(n, 1));
    m.check(rfx_extract_canon(m.ctx, bases.data(), readOff.data(), nr, m.param.kmerSize, m.param.frontClip,
                              m.param.endClip, out.data(), n, &n), "rfx_extract_canon");
    out.resize((size_t)n);
    return out;
}

KmerBinaryRDD ReflexivMain::KmerCounting_KmerCoverageFilter::call(const std::vector<uint64_t> &kmers) const {
    KmerBinaryRDD o;
    const int64_t n = (int64_t)kmers.size();
    o.kmer.resize((size_t)std::max<int64_t>(n, 1)); o.count.resize(o.kmer.size());
    int64_t mm = 0, d = 0;
    m.check(rfx_count_filter(m.ctx, kmers.data(), n, m.param.minKmerCoverage, m.param.maxKmerCoverage, m.param.twin,
                             o.kmer.data(), o.count.data(), n, &mm, &d), "rfx_count_filter");
    o.kmer.resize((size_t)mm); o.count.resize((size_t)mm);
    return o;
}

std::vector<uint64_t> ReflexivMain::ReverseComplementKmerBinaryExtractionFromDataset64::call(
    const std::vector<uint8_t> &bases, const std::vector<int64_t> &readOff) const {
    int64_t n = 0;
    const int64_t nr = (int64_t)readOff.size() - 1;
    const int W = m.param.kmerSize / 32 + 1;                     // kmerBinarySlots, U/DefaultParam.java:81
    int st = rfx_extract_canon_w(m.ctx, bases.data(), readOff.data(), nr, m.param.kmerSize, m.param.frontClip,
                                 m.param.endClip, nullptr, 0, &n);
    if (st != RFX_OK && st != RFX_E_CAP) m.check(st, "rfx_extract_canon_w");
    std::vector<uint64_t> out((size_t)std::max<int64_t>(n, 1) * W);
    m.check(rfx_extract_canon_w(m.ctx, bases.data(), readOff.data(), nr, m.param.kmerSize, m.param.frontClip,
                                m.param.endClip, out.data(), n, &n), "rfx_extract_canon_w");
    out.resize((size_t)n * W);
    return out;
}

void ReflexivMain::KmerBlocksCount::call(const std::vector<uint64_t> &kmers, std::vector<uint64_t> &keys,
                                         std::vector<int64_t> &counts) const {
    const int W = m.param.kmerSize / 32 + 1;
    const int64_t n = (int64_t)kmers.size() / W;
    keys.resize((size_t)std::max<int64_t>(n, 1) * W); counts.resize((size_t)std::max<int64_t>(n, 1));
    int64_t mm = 0, d = 0;
    m.check(rfx_count_filter_w(m.ctx, kmers.data(), n, m.param.kmerSize, m.param.minKmerCoverage, m.param.maxKmerCoverage,
                               keys.data(), counts.data(), n, &mm, &d), "rfx_count_filter_w");
    keys.resize((size_t)mm * W); counts.resize((size_t)mm);
}

std::string ReflexivMain::DSBinaryKmerToString::call(const uint64_t *b) const {
    static const char NUC[4] = {'A', 'C', 'G', 'T'};
    const int k = m.param.kmerSize, res = k % 32;                // kmerSizeResidue
    std::string sb;
    sb.reserve((size_t)k);
    for (int i = 0; i < (k / 32) * 32; i++) sb.push_back(NUC[(b[i / 32] >> (2 * (31 - i % 32))) & 3]);       // :346-352
    for (int i = (k / 32) * 32; i < k; i++) sb.push_back(NUC[(b[i / 32] >> (2 * (res - 1 - i % 32))) & 3]);  // :354-360
    return sb;
}

ReflexivSubKmerRDD ReflexivMain::KmerReverseComplement_ForwardSubKmerExtraction::call(const KmerBinaryRDD &in) const {
    ReflexivSubKmerRDD o;
    const int64_t n = (int64_t)in.kmer.size();
    o.reserve(2 * n, 2 * n);
    rfx_records r = o.view();
    m.check(rfx_rc_expand_subkmer(m.ctx, in.kmer.data(), in.count.data(), n, m.param.kmerSize, &r), "rfx_rc_expand_subkmer");
    o.shrink(r);
    return o;
}

ReflexivSubKmerRDD ReflexivMain::SortByKey::call(ReflexivSubKmerRDD &in, int P) const {
    ReflexivSubKmerRDD o;
    o.reserve(in.size(), (int64_t)in.ext.size());
    o.partStart.resize((size_t)P + 1);
    rfx_records ri = in.view(), ro = o.view();
    m.check(rfx_sort_records(m.ctx, &ri, P, &ro, o.partStart.data()), "rfx_sort_records");
    o.shrink(ro);
    return o;
}

static ReflexivSubKmerRDD fork_call(const ReflexivMain &m, bool reflected, ReflexivSubKmerRDD &in) {
    ReflexivSubKmerRDD o;
    const int P = (int)in.partStart.size() - 1;
    o.reserve(in.size(), in.size());
    o.partStart.resize(in.partStart.size());
    rfx_records ri = in.view(), ro = o.view();
    int st = reflected
        ? rfx_fork_filter_reflected(m.ctx, &ri, in.partStart.data(), P, m.param.kmerSize, m.param.minErrorCoverage,
                                    m.param.twin, &ro, o.partStart.data())
        : rfx_fork_filter_forward(m.ctx, &ri, in.partStart.data(), P, m.param.kmerSize, m.param.minErrorCoverage,
                                  m.param.twin, &ro, o.partStart.data());
    m.check(st, reflected ? "rfx_fork_filter_reflected" : "rfx_fork_filter_forward");
    o.shrink(ro);
    return o;
}
ReflexivSubKmerRDD ReflexivMain::FilterForkSubKmer::call(ReflexivSubKmerRDD &in) const { return fork_call(m, false, in); }
ReflexivSubKmerRDD ReflexivMain::FilterForkReflectedSubKmer::call(ReflexivSubKmerRDD &in) const { return fork_call(m, true, in); }

ReflexivSubKmerRDD ReflexivMain::ReflectedSubKmerExtractionFromForward::call(ReflexivSubKmerRDD &in) const {
    ReflexivSubKmerRDD o;
    o.reserve(in.size(), in.size());
    rfx_records ri = in.view(), ro = o.view();
    m.check(rfx_reflect_from_forward(m.ctx, &ri, m.param.kmerSize, &ro), "rfx_reflect_from_forward");
    o.shrink(ro);
    o.partStart = in.partStart;
    return o;
}

ReflexivSubKmerRDD ReflexivMain::kmerRandomReflection::call(ReflexivSubKmerRDD &in) const {
    ReflexivSubKmerRDD o;
    o.reserve(in.size(), in.size());
    rfx_records ri = in.view(), ro = o.view();
    m.check(rfx_random_reflection(m.ctx, &ri, in.partStart.data(), (int)in.partStart.size() - 1, m.param.kmerSize, &ro),
            "rfx_random_reflection");
    o.shrink(ro);
    o.partStart = in.partStart;
    return o;
}

ReflexivSubKmerRDD ReflexivMain::ExtendReflexivKmer::call(ReflexivSubKmerRDD &in) const {
    ReflexivSubKmerRDD o;
    const int P = (int)in.partStart.size() - 1;
    o.reserve(in.size(), (int64_t)in.ext.size());
    o.partStart.resize(in.partStart.size());
    rfx_records ri = in.view(), ro = o.view();
    m.check(rfx_extend_pass(m.ctx, &ri, in.partStart.data(), P, m.param.kmerSize, m.param.twin, stage, &ro,
                            o.partStart.data()), "rfx_extend_pass");
    o.shrink(ro);
    return o;
}

std::string ReflexivMain::KmerToContig::call(ReflexivSubKmerRDD &in, int64_t *nContigs) const {
    rfx_records ri = in.view();
    int64_t len = 0, nc = 0;
    int st = rfx_contigs_text(m.ctx, &ri, m.param.kmerSize, m.param.minContig, m.param.twin, nullptr, 0, &len, &nc);
    if (st != RFX_OK && st != RFX_E_CAP) m.check(st, "rfx_contigs_text");
    std::string out((size_t)len, '\0');
    m.check(rfx_contigs_text(m.ctx, &ri, m.param.kmerSize, m.param.minContig, m.param.twin, out.data(), len, &len, &nc),
            "rfx_contigs_text");
    if (nContigs) *nContigs = nc;
    return out;
}

// P/ReflexivMain.java:168-316
std::string ReflexivMain::assemblyFromCounts(const KmerBinaryRDD &counts, std::vector<int64_t> *trace) {
    if (!param.bubble)
        throw std::runtime_error("-bubble (no fork filtering) is unusable in the reference too (SURVEY.md C.6)");
    const int P0 = std::max(1, param.logicalPartitions);
    int P = P0;
    SortByKey sortByKey{*this};
    // Step: generate reverse complement k-mers, extract forward sub k-mers  :168-176
    ReflexivSubKmerRDD rdd = KmerReverseComplement_ForwardSubKmerExtraction{*this}.call(counts);
    // filter forks  :178-199
    rdd = sortByKey.call(rdd, P);
    rdd = FilterForkSubKmer{*this}.call(rdd);
    rdd = ReflectedSubKmerExtractionFromForward{*this}.call(rdd);
    rdd = sortByKey.call(rdd, P);
    rdd = FilterForkReflectedSubKmer{*this}.call(rdd);
    // Step 6  :204-205
    rdd = kmerRandomReflection{*this}.call(rdd);
    auto pass = [&](int stage) {
        rdd = sortByKey.call(rdd, P);                       // Step 7  :211
        rdd = ExtendReflexivKmer{*this, stage}.call(rdd);   // Step 8  :221-222
        if (trace) trace->push_back(rdd.size());
    };
    pass(0);
    int iterations = 0;
    for (int i = 1; i < 4; i++) { iterations++; pass(0); }  // :233-241
    iterations++;
    pass(1);                                                // :247-254
    int partitionNumber = P;                                // :263
    int64_t contigNumber = 0;
    while (iterations <= param.maximumIteration) {          // :265-296
        iterations++;
        if (iterations >= param.minimumIteration && iterations % 3 == 0) {
            int64_t currentContigNumber = rdd.size();       // count()  :270
            if (contigNumber == currentContigNumber) break;
            contigNumber = currentContigNumber;
            (void)partitionNumber;                          // coalesce (:277-281) is outside the order contract
        }
        pass(2);
    }
    int64_t nc = 0;
    return KmerToContig{*this}.call(rdd, &nc);              // Step 11  :302-316
}

std::string ReflexivMain::assembly(const std::string &fastqText, std::vector<int64_t> *trace) {
    std::vector<uint8_t> bases; std::vector<int64_t> readOff;
    FastqFilterWithQual{*this}.call(fastqText, bases, readOff);                          // Steps 1-2  :126-134
    std::vector<uint64_t> kmers = ReverseComplementKmerBinaryExtraction{*this}.call(bases, readOff);   // Step 3  :147-148
    KmerBinaryRDD counts = KmerCounting_KmerCoverageFilter{*this}.call(kmers);            // Steps 4-5  :154-163
    return assemblyFromCounts(counts, trace);
}

std::string ReflexivMain::assemblyResident(const std::string &fastqText, std::vector<int64_t> *trace) {
    if (!param.bubble)
        throw std::runtime_error("-bubble (no fork filtering) is unusable in the reference too (SURVEY.md C.6)");
    std::vector<uint8_t> bases; std::vector<int64_t> readOff;
    FastqFilterWithQual{*this}.call(fastqText, bases, readOff);
    rfx_params prm;
    rfx_default_params(&prm);
    prm.k = param.kmerSize; prm.min_cov = param.minKmerCoverage; prm.max_cov = param.maxKmerCoverage;
    prm.min_error_cov = param.minErrorCoverage; prm.min_contig = param.minContig;
    prm.min_iter = param.minimumIteration; prm.max_iter = param.maximumIteration;
    prm.front_clip = param.frontClip; prm.end_clip = param.endClip;
    prm.partitions = std::max(1, param.logicalPartitions); prm.twin = param.twin;
    const int64_t nr = (int64_t)readOff.size() - 1;
    std::vector<int64_t> tr((size_t)param.maximumIteration + 8);
    int64_t len = 0, nc = 0, nt = 0, kept = 0;
    std::string out((size_t)1 << 20, '\0');
    for (;;) {
        int st = rfx_assemble_reads(ctx, bases.data(), readOff.data(), nr, &prm, out.data(), (int64_t)out.size(), &len, &nc,
                                    tr.data(), (int64_t)tr.size(), &nt, &kept);
        if (st == RFX_E_CAP && len > (int64_t)out.size()) { out.assign((size_t)len, '\0'); continue; }
        check(st, "rfx_assemble_reads");
        break;
    }
    out.resize((size_t)len);
    if (trace) trace->assign(tr.begin(), tr.begin() + nt);
    return out;
}

namespace {
// the record set of the dynamic-k passes on the host: base codes + offsets (rfx_dyn_records)
struct DynHost {
    std::vector<uint8_t> key, ext;
    std::vector<int64_t> key_off{0}, ext_off{0};
    std::vector<int32_t> marker, left, right;
    rfx_dyn_records view() {
        rfx_dyn_records r{};
        r.n = (int64_t)marker.size();
        r.key = key.data(); r.key_off = key_off.data(); r.ext = ext.data(); r.ext_off = ext_off.data();
        r.marker = marker.data(); r.left = left.data(); r.right = right.data();
        r.cap_n = r.n; r.cap_key = (int64_t)key.size(); r.cap_ext = (int64_t)ext.size();
        return r;
    }
    void reserve(int64_t n, int64_t nk, int64_t ne) {
        key.assign((size_t)std::max<int64_t>(nk, 1), 0); ext.assign((size_t)std::max<int64_t>(ne, 1), 0);
        key_off.assign((size_t)n + 1, 0); ext_off.assign((size_t)n + 1, 0);
        marker.assign((size_t)std::max<int64_t>(n, 1), 0); left.assign(marker.size(), 0); right.assign(marker.size(), 0);
    }
};
inline uint8_t nucleotideValue(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }     // (FirstFour nucleotideValue)
inline int32_t attrClamp(long v) { return v >= 30000 ? 30000 : v <= -30000 ? -30000 : (int32_t)v; } // buildingAlongFromThreeInt read back
// one CSV row -> its comma separated fields (a leading "(" / a trailing ")" of the legacy tuple text dropped)
std::vector<std::string> fieldsOf(const std::string &line) {
    std::vector<std::string> f;
    size_t p = 0;
    while (true) {
        size_t e = line.find(',', p);
        f.push_back(line.substr(p, e == std::string::npos ? std::string::npos : e - p));
        if (e == std::string::npos) break;
        p = e + 1;
    }
    if (!f.empty() && !f[0].empty() && f[0][0] == '(') f[0].erase(0, 1);
    if (!f.empty() && !f.back().empty() && f.back().back() == ')') f.back().pop_back();
    return f;
}
void parseAttr(const std::string &a, long out[3]) {
    size_t p1 = a.find('|'), p2 = a.find('|', p1 + 1);
    out[0] = std::stol(a.substr(0, p1)); out[1] = std::stol(a.substr(p1 + 1, p2 - p1 - 1)); out[2] = std::stol(a.substr(p2 + 1));
}
// DynamicKmerBinarizerFromReducedToSubKmer.call: FirstFour (:2942-3016, rows "KMER,attr": key = the k-mer without its last base,
// extension = that base, orientation 1) or Iteration (rows "SUBKMER,attr,EXTENSION")
void dynBinarize(const std::string &csv, bool kmers, DynHost &d) {
    size_t pos = 0;
    d = DynHost();
    while (pos < csv.size()) {
        size_t e = csv.find('\n', pos);
        if (e == std::string::npos) e = csv.size();
        std::string line = csv.substr(pos, e - pos);
        pos = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> f = fieldsOf(line);
        if (f.size() < (kmers ? 2u : 3u)) throw std::runtime_error("dynamic-k row: " + line);
        long a[3];
        parseAttr(f[1], a);
        const std::string &k = f[0];
        const size_t kl = kmers ? k.size() - 1 : k.size();
        for (size_t i = 0; i < kl; i++) d.key.push_back(nucleotideValue(k[i]));
        if (kmers) d.ext.push_back(nucleotideValue(k[kl]));
        else for (char c : f[2]) d.ext.push_back(nucleotideValue(c));
        d.key_off.push_back((int64_t)d.key.size()); d.ext_off.push_back((int64_t)d.ext.size());
        d.marker.push_back(kmers ? 1 : (int32_t)a[0]); d.left.push_back(attrClamp(a[1])); d.right.push_back(attrClamp(a[2]));
    }
    if (d.key.empty()) d.key.push_back(0);
    if (d.ext.empty()) d.ext.push_back(0);
}
// DSBinarySubKmerWith{Short,Long}ExtensionToString.call (FirstFour:226-263): rows "SUBKMER,marker|left|right,EXTENSION"
std::string dynToText(const rfx_dyn_records &r) {
    std::string out;
    static const char NUC[] = "ACGT";
    for (int64_t i = 0; i < r.n; i++) {
        for (int64_t j = r.key_off[i]; j < r.key_off[i + 1]; j++) out.push_back(NUC[r.key[j] & 3]);
        out += "," + std::to_string(r.marker[i]) + "|" + std::to_string(r.left[i]) + "|" + std::to_string(r.right[i]) + ",";
        for (int64_t j = r.ext_off[i]; j < r.ext_off[i + 1]; j++) out.push_back(NUC[r.ext[j] & 3]);
        out.push_back('\n');
    }
    return out;
}
}  // namespace

static std::string dynRun(rfx_ctx *ctx, DynHost &in, int P, int reflect, int four, int start, int end, std::vector<int64_t> *trace) {
    rfx_dyn_records ri = in.view();
    DynHost out;
    int64_t cap_n = ri.n, cap_b = (int64_t)(in.key.size() + in.ext.size()) + 64;
    std::vector<int64_t> tr(256);
    int64_t nt = 0;
    for (;;) {
        out.reserve(cap_n, cap_b, cap_b);
        rfx_dyn_records ro = out.view();
        ro.cap_n = cap_n; ro.cap_key = cap_b; ro.cap_ext = cap_b;
        const int st = rfx_dyn_run(ctx, &ri, P, reflect, four, start, end, &ro, tr.data(), (int64_t)tr.size(), &nt);
        if (st == RFX_E_CAP) { cap_n = std::max(cap_n, ro.n); cap_b = std::max({cap_b, ro.need_key, ro.need_ext}) + 64; continue; }
        if (st != RFX_OK) throw std::runtime_error(std::string("rfx_dyn_run: ") + rfx_last_error(ctx));
        if (trace) trace->assign(tr.begin(), tr.begin() + nt);
        return dynToText(ro);
    }
}

std::string ReflexivMain::assemblyDynamicFirstFour(const std::string &csvText, std::vector<int64_t> *trace) {
    DynHost d;
    dynBinarize(csvText, true, d);
    return dynRun(ctx, d, std::max(1, param.logicalPartitions), 1, 4, 1, 0, trace);
}

std::string ReflexivMain::assemblyDynamicIteration(const std::string &csvText, int startIteration, int endIteration, std::vector<int64_t> *trace) {
    DynHost d;
    dynBinarize(csvText, false, d);
    // (the reference's loop runs while (iterations <= endIteration) { iterations++; ... }: endIteration - startIteration + 1 passes,
    // every one of them under param.startIteration's rules)
    return dynRun(ctx, d, std::max(1, param.logicalPartitions), 0, 0, startIteration, endIteration, trace);
}

std::string ReflexivMain::dedupContigText(const std::string &contigText) {
    std::string out(contigText.size() + 4096, '\0');
    int64_t len = 0, nc = 0;
    check(rfx_dedup_contig_text(ctx, contigText.data(), (int64_t)contigText.size(), param.minContig, out.data(), (int64_t)out.size(), &len, &nc,
                                nullptr), "rfx_dedup_contig_text");
    out.resize((size_t)len);
    return out;
}

std::string ReflexivMain::assemblyResidentSharded(const std::string &fastqText, int nGpus, std::vector<int64_t> *trace, int64_t gatherBelow) {
    if (!param.bubble)
        throw std::runtime_error("-bubble (no fork filtering) is unusable in the reference too (SURVEY.md C.6)");
    std::vector<uint8_t> bases; std::vector<int64_t> readOff;
    FastqFilterWithQual{*this}.call(fastqText, bases, readOff);
    rfx_params prm;
    rfx_default_params(&prm);
    prm.k = param.kmerSize; prm.min_cov = param.minKmerCoverage; prm.max_cov = param.maxKmerCoverage;
    prm.min_error_cov = param.minErrorCoverage; prm.min_contig = param.minContig;
    prm.min_iter = param.minimumIteration; prm.max_iter = param.maximumIteration;
    prm.front_clip = param.frontClip; prm.end_clip = param.endClip;
    prm.partitions = std::max(1, param.logicalPartitions); prm.twin = param.twin;
    const int64_t nr = (int64_t)readOff.size() - 1;
    uint8_t id[128];
    check(rfx_comm_unique_id(id), "rfx_comm_unique_id");
    std::vector<std::string> outs((size_t)nGpus), errs((size_t)nGpus);
    std::vector<int64_t> tr((size_t)param.maximumIteration + 8);
    int64_t nt = 0;
    std::vector<std::thread> th;
    for (int r = 0; r < nGpus; r++)
        th.emplace_back([&, r] {
            rfx_ctx *c = nullptr; rfx_comm *comm = nullptr;
            try {
                if (rfx_ctx_create(r, &c) != RFX_OK) throw std::runtime_error("rfx_ctx_create(" + std::to_string(r) + ")");
                if (rfx_comm_init(c, id, r, nGpus, &comm) != RFX_OK) throw std::runtime_error(std::string("rfx_comm_init: ") + rfx_last_error(c));
                const int64_t a = nr * r / nGpus, b = nr * (r + 1) / nGpus;            // contiguous shares of the reads
                std::string out(r == 0 ? (size_t)3 * (size_t)(readOff[nr] - readOff[0]) + ((size_t)1 << 20) : (size_t)1 << 12, '\0');
                int64_t len = 0, nc = 0, n_tr = 0, tot[3];
                for (;;) {
                    const int st = rfx_sharded_assemble_reads(c, comm, bases.data(), readOff.data() + a, b - a, &prm, 4, gatherBelow, out.data(),
                                                              (int64_t)out.size(), &len, &nc, r == 0 ? tr.data() : nullptr,
                                                              r == 0 ? (int64_t)tr.size() : 0, &n_tr, tot);
                    if (st == RFX_E_CAP && len > (int64_t)out.size()) { out.assign((size_t)len, '\0'); continue; }   // (on every rank at once: the retry is collective)
                    if (st != RFX_OK) throw std::runtime_error(std::string("rfx_sharded_assemble_reads: ") + rfx_last_error(c));
                    break;
                }
                out.resize((size_t)len);
                outs[(size_t)r] = out;
                if (r == 0) nt = n_tr;
            } catch (const std::exception &e) { errs[(size_t)r] = e.what(); }
            if (comm) rfx_comm_destroy(comm);
            if (c) rfx_ctx_destroy(c);
        });
    for (auto &t : th) t.join();
    for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
    if (trace) trace->assign(tr.begin(), tr.begin() + nt);
    return outs[0];
}

// KmerBinarizer.call  P/ReflexivDSMain.java:3883-3931: "KMER,count" or "(KMER,count)"; a count of
// ten or more digits saturates at 1,000,000,000; bases A0 C1 G2, anything else 3.
KmerBinaryRDD ReflexivMain::KmerBinarizer::call(const std::string &csvText) const {
    KmerBinaryRDD o;
    const int k = m.param.kmerSize;
    size_t pos = 0;
    while (pos < csvText.size()) {
        size_t e = csvText.find('\n', pos);
        if (e == std::string::npos) e = csvText.size();
        size_t l = e - pos;
        if (l > 0 && csvText[e - 1] == '\r') l--;
        std::string line = csvText.substr(pos, l);
        pos = e + 1;
        if (line.empty()) continue;
        size_t comma = line.find(',');
        if (comma == std::string::npos) throw std::runtime_error("k-mer count row without a comma: " + line);
        std::string kmer = line.substr(0, comma), cnt = line.substr(comma + 1);
        if (!kmer.empty() && kmer[0] == '(') kmer = kmer.substr(1);                    // :3892-3894
        int cover;
        if (!cnt.empty() && cnt.back() == ')') {                                       // :3896-3901
            cover = cnt.size() >= 11 ? 1000000000 : std::stoi(cnt.substr(0, cnt.size() - 1));
        } else {                                                                       // :3902-3908
            cover = cnt.size() >= 10 ? 1000000000 : std::stoi(cnt);
        }
        if ((int)kmer.size() < k) throw std::runtime_error("k-mer shorter than -kmer: " + kmer);
        uint64_t b = 0;
        for (int i = 0; i < k; i++) {                                                  // :3912-3919
            char c = kmer[(size_t)i];
            b = (b << 2) | (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u);
        }
        o.kmer.push_back(b); o.count.push_back(cover);
    }
    return o;
}

// `run -kmerc`: load -> filter(min <= count <= max) (P/ReflexivDSMain.java:458-478) -> the same driver.
// The order contract wants the (kmer,count) list ascending, whatever order the files came in.
std::string ReflexivMain::assemblyFromKmer(const std::string &csvText, std::vector<int64_t> *trace) {
    KmerBinaryRDD in = KmerBinarizer{*this}.call(csvText);
    std::vector<size_t> idx;
    for (size_t i = 0; i < in.kmer.size(); i++)
        if (in.count[i] >= param.minKmerCoverage && in.count[i] <= param.maxKmerCoverage) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return in.kmer[a] < in.kmer[b]; });
    KmerBinaryRDD f;
    for (size_t i : idx) { f.kmer.push_back(in.kmer[i]); f.count.push_back(in.count[i]); }
    return assemblyFromCounts(f, trace);
}

// KmerBinarizer.call  P/ReflexivDSMain64.java:10772-10836: the same row rules as above, the k-mer into words of 31 bases
void ReflexivMain::KmerBinarizer64::call(const std::string &csvText, std::vector<uint64_t> &kmers, std::vector<int32_t> &counts) const {
    const int k = m.param.kmerSize, W = (k - 1) / 31 + 1;                                 // kmerBinarySlotsAssemble
    size_t pos = 0;
    while (pos < csvText.size()) {
        size_t e = csvText.find('\n', pos);
        if (e == std::string::npos) e = csvText.size();
        size_t l = e - pos;
        if (l > 0 && csvText[e - 1] == '\r') l--;
        std::string line = csvText.substr(pos, l);
        pos = e + 1;
        if (line.empty()) continue;
        size_t comma = line.find(',');
        if (comma == std::string::npos) throw std::runtime_error("k-mer count row without a comma: " + line);
        std::string kmer = line.substr(0, comma), cnt = line.substr(comma + 1);
        if (!kmer.empty() && kmer[0] == '(') kmer = kmer.substr(1);                       // :10790-10792
        int cover;
        if (!cnt.empty() && cnt.back() == ')') cover = cnt.size() >= 11 ? 1000000000 : std::stoi(cnt.substr(0, cnt.size() - 1));   // :10794-10799
        else cover = cnt.size() >= 10 ? 1000000000 : std::stoi(cnt);                      // :10800-10806
        if ((int)kmer.size() < k) throw std::runtime_error("k-mer shorter than -kmer: " + kmer);
        const size_t at = kmers.size();
        kmers.resize(at + (size_t)W, 0);
        for (int i = 0; i < k; i++) {                                                     // :10811-10819
            const char c = kmer[(size_t)i];
            kmers[at + (size_t)(i / 31)] = (kmers[at + (size_t)(i / 31)] << 2) | (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u);
        }
        counts.push_back(cover);
    }
}

// `run -kmerc COUNTS -kmer 63`: load -> filter(min <= count <= max) (:458-478) -> the k > 31 driver in one call.
// The order contract wants the list ascending by base string, whatever order the files came in.
std::string ReflexivMain::assemblyFromKmer64(const std::string &csvText, std::vector<int64_t> *trace) {
    const int W = (param.kmerSize - 1) / 31 + 1;
    std::vector<uint64_t> km; std::vector<int32_t> cn;
    KmerBinarizer64{*this}.call(csvText, km, cn);
    std::vector<size_t> idx;
    for (size_t i = 0; i < cn.size(); i++)
        if (cn[i] >= param.minKmerCoverage && cn[i] <= param.maxKmerCoverage) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        return std::lexicographical_compare(km.begin() + a * W, km.begin() + (a + 1) * W, km.begin() + b * W, km.begin() + (b + 1) * W);
    });
    std::vector<uint64_t> fk; std::vector<int32_t> fc;
    for (size_t i : idx) { fk.insert(fk.end(), km.begin() + i * W, km.begin() + (i + 1) * W); fc.push_back(cn[i]); }
    rfx_params prm; rfx_default_params(&prm);
    prm.k = param.kmerSize; prm.min_cov = param.minKmerCoverage; prm.max_cov = param.maxKmerCoverage;
    prm.min_error_cov = param.minErrorCoverage; prm.min_contig = param.minContig; prm.min_iter = param.minimumIteration;
    prm.max_iter = param.maximumIteration; prm.front_clip = param.frontClip; prm.end_clip = param.endClip;
    prm.partitions = param.logicalPartitions;
    std::vector<int64_t> tr((size_t)param.maximumIteration + 8);
    int64_t nt = 0, nc = 0, len = 0;
    std::string out((size_t)(4 * (fc.size() + 16) * (size_t)(param.kmerSize + 8) + 1024), '\0');
    for (;;) {
        const int st = rfx_assemble_counts_w(ctx, fk.data(), fc.data(), (int64_t)fc.size(), &prm, &out[0], (int64_t)out.size(), &len, &nc,
                                             tr.data(), (int64_t)tr.size(), &nt);
        if (st == RFX_E_CAP && len > (int64_t)out.size()) { out.resize((size_t)len); continue; }
        check(st, "rfx_assemble_counts_w");
        break;
    }
    out.resize((size_t)len);
    if (trace) trace->assign(tr.begin(), tr.begin() + nt);
    return out;
}

// P/ReflexivCounter.java:109-191: k-mer, count text lines
std::string ReflexivMain::counter(const std::string &fastqText) {
    static const char NUC[4] = {'A', 'C', 'G', 'T'};
    std::vector<uint8_t> bases; std::vector<int64_t> readOff;
    DSFastqFilterOnlySeq{*this}.call(fastqText, bases, readOff);      // P/ReflexivDataFrameCounter.java:170-173
    if (param.kmerSize > 31) {                       // P/ReflexivDataFrameCounter64.java:133-232
        const int W = param.kmerSize / 32 + 1;
        std::vector<uint64_t> kmers = ReverseComplementKmerBinaryExtractionFromDataset64{*this}.call(bases, readOff);
        std::vector<uint64_t> keys; std::vector<int64_t> cnt;
        KmerBlocksCount{*this}.call(kmers, keys, cnt);
        std::string out;
        DSBinaryKmerToString toString{*this};
        for (size_t i = 0; i < cnt.size(); i++) {
            out += toString.call(keys.data() + i * W);
            out.push_back(',');
            out += std::to_string(cnt[i]);
            out.push_back('\n');
        }
        return out;
    }
    std::vector<uint64_t> kmers = ReverseComplementKmerBinaryExtraction{*this}.call(bases, readOff);
    KmerBinaryRDD counts = KmerCounting_KmerCoverageFilter{*this}.call(kmers);
    std::string out;
    const int k = param.kmerSize;
    for (size_t i = 0; i < counts.kmer.size(); i++) {
        for (int j = 0; j < k; j++) out.push_back(NUC[(counts.kmer[i] >> (2 * (k - 1 - j))) & 3]);
        out.push_back(',');
        out += std::to_string(counts.count[i]);
        out.push_back('\n');
    }
    return out;
}

}  // namespace reflexiv
