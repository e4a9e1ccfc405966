#!/usr/bin/env python3
"""Golden vectors made by the REFERENCE'S OWN operator classes.

The reference is Java; no JVM exists in the build container.  tools/java2py.py parses the operator classes from the
reference's source text (read from /root/reference at generation time, never stored here), translates them statement
by statement to Python on Java's integer semantics and this script runs them:

  * CHAINS -- small assemblies driven through the reference's classes in the order of its drivers
    (P/ReflexivDSMain.java:196-345, P/ReflexivMain.java:147-310, P/ReflexivDSMain64.java:458-826), from FASTQ lines
    (k <= 31) or counter rows (k > 31) to the contig text.  What sits BETWEEN two operator classes -- Spark's
    groupBy().count(), filter(), sort("k-1"), the cut into partitions -- is not reference code; it is done here under
    the order contract of DESIGN.md section 2 (ascending k-mers, stable sort, partition p starts at floor(p*n/P) moved
    forward past equal keys).  Every operator's input and output records are stored.
  * FUZZ -- random sorted partitions through single operator classes (extend passes with bubble distances and long
    extension arrays, fork filters with tied coverages, the k > 31 from-counts extras), for the branches the chains
    do not reach.

Output: tests/golden/reference_vectors.npz (data only: inputs and the reference's outputs).
tests/test_reference_vectors.py checks the oracle against them on the CPU; tests/test_gpu_reference_vectors.py checks
the HIP operators through the C ABI.

Run:  python tests/golden/make_reference_vectors.py            (needs /root/reference; ~ minutes)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import java2py as jp  # noqa: E402

REF = os.environ.get("RFX_REFERENCE", "/root/reference") + "/src/main/java/uni/bielefeld/cmg/reflexiv/"
P_ = REF + "pipeline/"
M64 = (1 << 64) - 1


def u64(x):
    return int(x) & M64


# ----------------------------------------------------------------------------------------------- families
DS_CLASSES = ["TagRowContigID", "DSKmerToContig", "DSBinaryReflexivKmerArrayToString", "DSExtendReflexivKmerToArrayLoop",
              "DSExtendReflexivKmerToArrayFirstTime", "DSExtendReflexivKmer", "DSFilterForkSubKmer",
              "DSFilterForkSubKmerWithErrorCorrection", "DSFilterForkReflectedSubKmer",
              "DSFilterForkReflectedSubKmerWithErrorCorrection", "DSForwardSubKmerExtraction",
              "DSReflectedSubKmerExtractionFromForward", "DSkmerRandomReflection", "DSKmerReverseComplementLong", "KmerBinarizer",
              "ReverseComplementKmerBinaryExtractionFromDataset", "DSFastqFilterWithQual"]
RDD_CLASSES = ["TagContigID", "KmerToContig", "BinaryReflexivKmerArrayToString", "ExtendReflexivKmerToArrayLoop",
               "ExtendReflexivKmerToArrayFirstTime", "ExtendReflexivKmer", "FilterForkSubKmer",
               "FilterForkSubKmerWithErrorCorrection", "FilterForkReflectedSubKmer",
               "FilterForkReflectedSubKmerWithErrorCorrection", "ForwardSubKmerExtraction",
               "ReflectedSubKmerExtractionFromForward", "kmerRandomReflection", "KmerReverseComplement",
               "ReverseComplementKmerBinaryExtraction", "FastqFilterWithQual", "KmerCoverageFilter", "KmerCounting"]
DS64_CLASSES = ["TagRowContigID", "DSKmerToContig", "DSBinaryReflexivKmerArrayToString", "DSReflexivAndForwardKmer",
                "DSFilterStillExtendableKmerEnds", "DSFilterStillExtendableKmerFromPairs", "DSFilterUnExtendableKmerLeftEnds",
                "DSFilterUnExtendableKmerRightEnds", "DSFilterExtendableKmerPairs", "DSFilterUnExtendableKmer",
                "DSExtendReflexivKmerToArrayLoop", "DSExtendReflexivKmerToArrayFirstTime", "DSExtendReflexivKmer",
                "DSFilterForkSubKmer", "DSFilterForkSubKmerWithErrorCorrection", "DSFilterForkReflectedSubKmer",
                "DSFilterForkReflectedSubKmerWithErrorCorrection", "DSForwardSubKmerExtraction",
                "DSReflectedSubKmerExtractionFromForward", "DSkmerRandomReflection", "DSKmerReverseComplement", "KmerBinarizer"]
CNT64_CLASSES = ["DSFastqFilterOnlySeq", "DSBinaryKmerToString", "ReverseComplementKmerBinaryExtractionFromDataset64"]

_cache = {}


def family(name):
    if name not in _cache:
        f, cl = {"ds": ("ReflexivDSMain.java", DS_CLASSES), "rdd": ("ReflexivMain.java", RDD_CLASSES),
                 "ds64": ("ReflexivDSMain64.java", DS64_CLASSES), "cnt64": ("ReflexivDataFrameCounter64.java", CNT64_CLASSES)}[name]
        _cache[name] = jp.translate_classes(P_ + f, cl)
    return _cache[name]


def default_param():
    if "param" not in _cache:
        _cache["param"] = jp.translate_plain_class(REF + "util/DefaultParam.java", "DefaultParam")
    return _cache["param"]


def make_param(k, **kw):
    p = default_param()(None)
    p.setAllbyKmerSize(jp._I(k))
    for a, b in kw.items():
        setattr(p, a, jp._I(b) if isinstance(b, int) and not isinstance(b, bool) else b)
    return p


def new_op(fam, cls, param):
    classes = family(fam)
    return classes[cls](jp.Outer(param, classes))


# ----------------------------------------------------------------------------------------------- rows <-> records
# canonical record: (key words tuple, marker, ext words tuple, left, right); all words unsigned 64 bit

def Ls(words):
    return jp.Seq([jp._L(w) for w in words])


def to_row(fam, rec, single):
    key, marker, ext, left, right = rec
    if fam == "ds":
        e = jp._L(ext[0]) if single else Ls(ext)
        return jp.Row([jp._L(key[0]), jp._I(marker), e, jp._I(left), jp._I(right)])
    if fam == "ds64":
        e = jp._L(ext[0]) if single else Ls(ext)
        return jp.Row([Ls(key), jp._I(marker), e, jp._I(left), jp._I(right)])
    if fam == "rdd":
        e = jp._L(ext[0]) if single else [jp._L(w) for w in ext]
        return jp.JTuple(jp._L(key[0]), jp.JTuple(jp._I(marker), e, jp._I(left), jp._I(right)))
    raise KeyError(fam)


def words_of(x):
    if isinstance(x, jp.Seq):
        return tuple(u64(w.v) for w in x.items)
    if isinstance(x, list):
        return tuple(u64(w.v) for w in x)
    return (u64(x.v),)


def from_row(fam, r):
    if fam == "rdd":
        t = r.items[1]
        return (words_of(r.items[0]), t.items[0].v, words_of(t.items[1]), t.items[2].v, t.items[3].v)
    v = r.vals
    return (words_of(v[0]), v[1].v, words_of(v[2]), v[3].v, v[4].v)


def drain(it):
    out = []
    while it.hasNext():
        out.append(it.next())
    return out


def run_record_op(fam, cls, param, recs, single_in):
    """one task = one fresh operator instance on one partition (the reference serialises the operator into every task)"""
    op = new_op(fam, cls, param)
    rows = [to_row(fam, r, single_in) for r in recs]
    return [from_row(fam, r) for r in drain(op.call(jp.JIter(rows)))]


def partition_starts(keys, P):
    n = len(keys)
    st, prev = [], 0
    for p in range(P):
        s = p * n // P
        s = max(s, prev)
        while 0 < s < n and keys[s] == keys[s - 1]:
            s += 1
        st.append(s)
        prev = s
    st.append(n)
    return st


def sort_records(recs):
    return sorted(recs, key=lambda r: r[0])              # stable; keys compare as unsigned words, first word first


def by_partition(fam, cls, param, recs, P, single_in, starts=None):
    """sorted records -> the operator over each logical partition -> (records, output partition starts)"""
    st = starts if starts is not None else partition_starts([r[0] for r in recs], P)
    out, ost = [], [0]
    for p in range(len(st) - 1):
        out += run_record_op(fam, cls, param, recs[st[p]:st[p + 1]], single_in)
        ost.append(len(out))
    return out, st, ost


# ----------------------------------------------------------------------------------------------- storage
class Store:
    def __init__(self):
        self.d = {}
        self.skip = ()                 # name prefixes whose per-stage records are NOT stored (large chains: trace + text only)

    def records(self, name, recs, starts=None):
        if any(name.startswith(p) for p in self.skip):
            return
        n = len(recs)
        kw = len(recs[0][0]) if n else 1
        self.d[name + "/key"] = np.array([r[0] for r in recs], np.uint64).reshape(n, kw)
        self.d[name + "/marker"] = np.array([r[1] for r in recs], np.int32)
        off = np.zeros(n + 1, np.int64)
        if n:
            off[1:] = np.cumsum([len(r[2]) for r in recs])
        self.d[name + "/ext_off"] = off
        self.d[name + "/ext"] = np.array([w for r in recs for w in r[2]], np.uint64)
        self.d[name + "/left"] = np.array([r[3] for r in recs], np.int32)
        self.d[name + "/right"] = np.array([r[4] for r in recs], np.int32)
        if starts is not None:
            self.d[name + "/starts"] = np.array(starts, np.int64)

    def put(self, name, arr):
        self.d[name] = np.asarray(arr)

    def text(self, name, s):
        self.d[name] = np.frombuffer(s.encode(), np.uint8)


# ----------------------------------------------------------------------------------------------- synthetic input
def synth_genome(rng, n, with_repeat=True, with_snp_copy=True):
    g = rng.integers(0, 4, n)
    if with_repeat and n >= 300:
        g[n - 90:n - 30] = g[40:100]                      # a 60-base repeat
    return g


def synth_reads(rng, genome, read_len, depth, err, second_allele=None):
    """reads from both strands; `second_allele`: (position, base) carried by a third of the reads (a bubble)"""
    n = len(genome)
    nreads = depth * n // read_len
    reads = []
    for _ in range(nreads):
        s = int(rng.integers(0, n - read_len + 1))
        r = genome[s:s + read_len].copy()
        if second_allele is not None and s <= second_allele[0] < s + read_len and rng.random() < 0.34:
            r[second_allele[0] - s] = second_allele[1]
        e = rng.random(read_len) < err
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) % 4
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append("".join("ACGT"[b] for b in r))
    return reads


def fastq_lines(reads):
    lines = []
    for i, r in enumerate(reads):
        lines += [f"@r{i}", r, "+", "I" * len(r)]
    return lines


# ----------------------------------------------------------------------------------------------- chains, k <= 31
def extract_count_k31(fam, param, lines):
    """FASTQ lines -> FastqFilterWithQual -> extraction class -> (ascending canonical k-mers, counts).  groupBy().count()
    and the ascending order are Spark's / the order contract's, everything else the reference's."""
    if fam == "ds":
        flt = new_op("ds", "DSFastqFilterWithQual", param)
        units = [flt.call(ln) for ln in lines]
        units = [u for u in units if u is not None]
        ex = new_op("ds", "ReverseComplementKmerBinaryExtractionFromDataset", param)
        kmers = [u64(x.v) for x in drain(ex.call(jp.JIter(units)))]
    else:
        flt = new_op("rdd", "FastqFilterWithQual", param)
        units = [flt.call(ln) for ln in lines]
        units = [u for u in units if u is not None]
        ex = new_op("rdd", "ReverseComplementKmerBinaryExtraction", param)
        kmers = [u64(t.items[0].v) for t in drain(ex.call(jp.JIter(units)))]
    cnt = {}
    for x in kmers:
        cnt[x] = cnt.get(x, 0) + 1
    keys = sorted(cnt)
    return kmers, keys, [cnt[x] for x in keys]


def chain_k31(store, name, fam, k, P, min_cov, min_error_cov, lines, max_iter=150, min_iter=15, min_contig=100):
    """the driver of P/ReflexivDSMain.java:196-345 (fam "ds") or P/ReflexivMain.java:147-310 (fam "rdd")"""
    param = make_param(k, minKmerCoverage=min_cov, minErrorCoverage=min_error_cov, maximumIteration=max_iter,
                       minimumIteration=min_iter, minContig=min_contig)
    store.put(name + "/meta", np.array([k, P, min_cov, min_error_cov, max_iter, min_iter, min_contig], np.int64))
    kmers, keys, counts = extract_count_k31(fam, param, lines)
    store.text(name + "/fastq", "\n".join(lines) + "\n")
    store.put(name + "/instances", np.array(kmers, np.uint64))
    if fam == "rdd" and not (min_cov > 1):                 # P/ReflexivMain.java:160: the filter only when min > 1
        kept = list(zip(keys, counts))
    else:
        kept = [(x, c) for x, c in zip(keys, counts) if min_cov <= c <= int(param.maxKmerCoverage)]
    store.put(name + "/kept_keys", np.array([x for x, _ in kept], np.uint64))
    store.put(name + "/kept_counts", np.array([c for _, c in kept], np.int32))
    C = {"ds": dict(rc="DSKmerReverseComplementLong", fwd="DSForwardSubKmerExtraction", ff="DSFilterForkSubKmer",
                    ffe="DSFilterForkSubKmerWithErrorCorrection", refl="DSReflectedSubKmerExtractionFromForward",
                    fr="DSFilterForkReflectedSubKmer", fre="DSFilterForkReflectedSubKmerWithErrorCorrection",
                    rnd="DSkmerRandomReflection", e1="DSExtendReflexivKmer", e2="DSExtendReflexivKmerToArrayFirstTime",
                    e3="DSExtendReflexivKmerToArrayLoop"),
         "rdd": dict(rc="KmerReverseComplement", fwd="ForwardSubKmerExtraction", ff="FilterForkSubKmer",
                     ffe="FilterForkSubKmerWithErrorCorrection", refl="ReflectedSubKmerExtractionFromForward",
                     fr="FilterForkReflectedSubKmer", fre="FilterForkReflectedSubKmerWithErrorCorrection",
                     rnd="kmerRandomReflection", e1="ExtendReflexivKmer", e2="ExtendReflexivKmerToArrayFirstTime",
                     e3="ExtendReflexivKmerToArrayLoop")}[fam]
    # RC expand + forward sub-k-mers (one partition per logical partition of the ascending list; both are 1 -> n maps)
    op = new_op(fam, C["rc"], param)
    if fam == "ds":
        rows = [jp.Row([jp._L(x), jp._L(c)]) for x, c in kept]           # groupBy().count() yields a long count
        both = drain(op.call(jp.JIter(rows)))
        op2 = new_op(fam, C["fwd"], param)
        recs = [from_row(fam, r) for r in drain(op2.call(jp.JIter(both)))]
    else:
        rows = [jp.JTuple(jp._L(x), jp._I(c)) for x, c in kept]
        both = drain(op.call(jp.JIter(rows)))
        op2 = new_op(fam, C["fwd"], param)
        recs = [from_row(fam, r) for r in drain(op2.call(jp.JIter(both)))]
    store.records(name + "/forward", recs)
    stage = [0]

    def step(label, cls, recs, single, sort=True, starts=None):
        if sort:
            recs = sort_records(recs)
        out, st, ost = by_partition(fam, cls, param, recs, P, single, starts)
        tag = f"{name}/s{stage[0]:02d}_{label}"
        store.records(tag + "/in", recs, st)
        store.records(tag + "/out", out, ost)
        stage[0] += 1
        return out, ost

    recs, _ = step("fork_forward", C["ffe"] if min_error_cov else C["ff"], recs, True)
    recs, ost = step("reflect", C["refl"], recs, True, sort=False, starts=_)
    recs, ost = step("fork_reflected", C["fre"] if min_error_cov else C["fr"], recs, True)
    recs, ost = step("random_reflection", C["rnd"], recs, True, sort=False, starts=ost)
    trace = []
    recs, _ = step("extend_single", C["e1"], recs, True)
    trace.append(len(recs))
    iterations = 0
    for _i in range(1, 4):
        iterations += 1
        recs, _ = step("extend_single", C["e1"], recs, True)
        trace.append(len(recs))
    iterations += 1
    recs, _ = step("extend_first_array", C["e2"], recs, True)
    trace.append(len(recs))
    contig_number = 0
    while iterations <= max_iter:
        iterations += 1
        if iterations >= min_iter and iterations % 3 == 0:
            if contig_number == len(recs):
                break
            contig_number = len(recs)
        recs, _ = step("extend_array", C["e3"], recs, False)
        trace.append(len(recs))
    store.put(name + "/trace", np.array(trace, np.int64))
    store.records(name + "/final", recs)
    # records -> strings -> contigs (DS :852-915, :741-800, :715-725; RDD :693-758, :588-638, :571-582)
    if fam == "ds":
        op = new_op("ds", "DSBinaryReflexivKmerArrayToString", param)
        srows = drain(op.call(jp.JIter([to_row("ds", r, False) for r in recs])))
        op = new_op("ds", "DSKmerToContig", param)
        crows = drain(op.call(jp.JIter(srows)))
        tag = new_op("ds", "TagRowContigID", param)
        text = []
        for i, r in enumerate(crows):
            text += drain(tag.call(jp.JTuple(r, jp._L(i))))
    else:
        op = new_op("rdd", "BinaryReflexivKmerArrayToString", param)
        srows = drain(op.call(jp.JIter([to_row("rdd", r, False) for r in recs])))
        op = new_op("rdd", "KmerToContig", param)
        crows = []
        for r in srows:
            crows += drain(op.call(r))
        tag = new_op("rdd", "TagContigID", param)
        text = []
        for i, r in enumerate(crows):
            text += drain(tag.call(jp.JTuple(r, jp._L(i))))
    store.text(name + "/contigs", "".join(t + "\n" for t in text))
    return trace, text


# ----------------------------------------------------------------------------------------------- chain, k > 31
def chain_k64(store, name, k, P, min_cov, min_error_cov, reads, max_iter=150, min_iter=15, min_contig=100):
    """counter (P/ReflexivDataFrameCounter64.java:176-236) -> CSV rows -> assemblyFromKmer (P/ReflexivDSMain64.java:374-826)"""
    fam = "ds64"
    param = make_param(k, minKmerCoverage=min_cov, minErrorCoverage=min_error_cov, maximumIteration=max_iter,
                       minimumIteration=min_iter, minContig=min_contig)
    store.put(name + "/meta", np.array([k, P, min_cov, min_error_cov, max_iter, min_iter, min_contig], np.int64))
    store.text(name + "/reads", "\n".join(reads) + "\n")
    # counter: extraction (32 bases per word), groupBy().count(), filter, text rows
    ex = new_op("cnt64", "ReverseComplementKmerBinaryExtractionFromDataset64", param)
    rows = drain(ex.call(jp.JIter(reads)))
    inst = [words_of(r.vals[0]) for r in rows]
    store.put(name + "/instances", np.array(inst, np.uint64))
    cnt = {}
    for x in inst:
        cnt[x] = cnt.get(x, 0) + 1
    keys = sorted(cnt)
    kept = [(x, cnt[x]) for x in keys if min_cov <= cnt[x] <= int(param.maxKmerCoverage)]
    tostr = new_op("cnt64", "DSBinaryKmerToString", param)
    srows = drain(tostr.call(jp.JIter([jp.Row([Ls(x), jp._L(c)]) for x, c in kept])))
    csv = [(r.vals[0], r.vals[1]) if isinstance(r, jp.Row) else r for r in srows]
    store.text(name + "/csv", "".join(f"{a},{b}\n" for a, b in [(str(x[0]), str(x[1])) for x in csv]))
    # assembler: KmerBinarizer (31 bases per word) + the count filter of :473-478
    binz = new_op(fam, "KmerBinarizer", param)
    brow = drain(binz.call(jp.JIter([jp.Row([str(a), str(b)]) for a, b in csv])))
    kept2 = [(words_of(r.vals[0]), r.vals[1].v) for r in brow]
    kept2 = [(x, c) for x, c in kept2 if min_cov <= c <= int(param.maxKmerCoverage)]
    store.put(name + "/asm_keys", np.array([x for x, _ in kept2], np.uint64))
    store.put(name + "/asm_counts", np.array([c for _, c in kept2], np.int32))
    op = new_op(fam, "DSKmerReverseComplement", param)
    both = drain(op.call(jp.JIter([jp.Row([Ls(x), jp._I(c)]) for x, c in kept2])))
    # (DSForwardSubKmerExtraction casts Row.get(0) to long[] (:10381): it is handed the long[] rows DSKmerReverseComplement made)
    op2 = new_op(fam, "DSForwardSubKmerExtraction", param)
    recs = [from_row(fam, r) for r in drain(op2.call(jp.JIter(both)))]
    store.records(name + "/forward", recs)
    stage = [0]

    def step(label, cls, recs, single, sort=True, starts=None, prm=param):
        if sort:
            recs = sort_records(recs)
        out, st, ost = by_partition(fam, cls, prm, recs, P, single, starts)
        tag = f"{name}/s{stage[0]:02d}_{label}"
        store.records(tag + "/in", recs, st)
        store.records(tag + "/out", out, ost)
        stage[0] += 1
        return out, ost

    recs, st = step("fork_forward", "DSFilterForkSubKmerWithErrorCorrection" if min_error_cov else "DSFilterForkSubKmer", recs, True)
    recs, ost = step("reflect", "DSReflectedSubKmerExtractionFromForward", recs, True, sort=False, starts=st)
    recs, ost = step("fork_reflected", "DSFilterForkReflectedSubKmerWithErrorCorrection" if min_error_cov
                     else "DSFilterForkReflectedSubKmer", recs, True)
    recs, ost = step("random_reflection", "DSkmerRandomReflection", recs, True, sort=False, starts=ost)
    trace = []
    recs, _ = step("extend_single", "DSExtendReflexivKmer", recs, True)
    trace.append(len(recs))
    iterations = 0
    for _i in range(1, 4):
        iterations += 1
        recs, _ = step("extend_single", "DSExtendReflexivKmer", recs, True)
        trace.append(len(recs))
    iterations += 1
    recs, _ = step("extend_first_array", "DSExtendReflexivKmerToArrayFirstTime", recs, True)
    trace.append(len(recs))
    contig_number = 0
    scramble = 2
    unext = None
    while iterations <= max_iter:
        iterations += 1
        if iterations == min_iter + 3:                                       # the from-counts extras :584-619
            recs, _ = step("x_double", "DSReflexivAndForwardKmer", recs, False)
            ext_, _ = step("x_extendable_pairs", "DSFilterExtendableKmerPairs", recs, False)
            une_, _ = step("x_unextendable", "DSFilterUnExtendableKmer", recs, False)
            recs, _ = step("x_first_of_key", "DSFilterStillExtendableKmerFromPairs", ext_, False)
            unext, _ = step("x_first_of_key", "DSFilterStillExtendableKmerFromPairs", une_, False)
        if iterations >= min_iter + 3 and iterations % 3 == 0:
            if contig_number == len(recs):
                if scramble == 2:
                    scramble = 3
                    contig_number = len(recs)
                else:
                    break
            else:
                contig_number = len(recs)
        prm = make_param(k, minKmerCoverage=min_cov, minErrorCoverage=min_error_cov, scramble=scramble)
        recs, _ = step(f"extend_array_scr{scramble}", "DSExtendReflexivKmerToArrayLoop", recs, False, prm=prm)
        trace.append(len(recs))
    if unext is not None:                                                    # :672-712
        recs = recs + unext
        recs, _ = step("x_left_ends", "DSFilterUnExtendableKmerLeftEnds", recs, False, sort=False, starts=[0, len(recs)])
        recs, _ = step("x_longer_of_key", "DSFilterStillExtendableKmerEnds", recs, False)
        recs, _ = step("x_right_ends", "DSFilterUnExtendableKmerRightEnds", recs, False, sort=False, starts=[0, len(recs)])
        recs, _ = step("x_longer_of_key", "DSFilterStillExtendableKmerEnds", recs, False)
    recs = sort_records(recs)                                                # :714
    store.put(name + "/trace", np.array(trace, np.int64))
    store.records(name + "/final", recs)
    op = new_op(fam, "DSBinaryReflexivKmerArrayToString", param)
    srows = drain(op.call(jp.JIter([to_row(fam, r, False) for r in recs])))
    op = new_op(fam, "DSKmerToContig", param)
    crows = drain(op.call(jp.JIter(srows)))
    tag = new_op(fam, "TagRowContigID", param)
    text = []
    for i, r in enumerate(crows):
        text += drain(tag.call(jp.JTuple(r, jp._L(i))))
    store.text(name + "/contigs", "".join(t + "\n" for t in text))
    return trace, text


# ----------------------------------------------------------------------------------------------- fuzz
def key_words(seq, kw):
    out = []
    for i in range(kw):
        x = 0
        for b in seq[31 * i:31 * (i + 1)]:
            x = (x << 2) | int(b)
        out.append(x)
    return tuple(out)


def ext_words(seq):
    L = len(seq)
    f = (L - 1) % 31 + 1
    x = 1
    for b in seq[:f]:
        x = (x << 2) | int(b)
    out = [x]
    for i in range(f, L, 31):
        x = 0
        for b in seq[i:i + 31]:
            x = (x << 2) | int(b)
        out.append(x)
    return tuple(out)


def fuzz_partition(rng, k, maxlen, groups, single):
    sub = k - 1
    kw = (sub - 1) // 31 + 1 if k > 32 else 1
    recs = []
    for _ in range(groups):
        keyseq = rng.integers(0, 4, sub)
        for _r in range(int(rng.integers(1, 4))):
            L = int(rng.integers(1, maxlen + 1))
            if not single and rng.random() < 0.3:
                L = int(rng.choice([30, 31, 32, 61, 62, 63, 93, 94]))   # word boundaries
                L = min(L, maxlen) if maxlen >= 30 else L
            e = rng.integers(0, 4, L)
            left = int(rng.integers(-60, 0)) if rng.random() < 0.65 else int(rng.integers(0, 3 * maxlen))
            right = int(rng.integers(-60, 0)) if rng.random() < 0.65 else int(rng.integers(0, 3 * maxlen))
            recs.append((key_words(keyseq, kw), int(rng.integers(1, 3)), ext_words(e), left, right))
    return sort_records(recs)


def fuzz_set(store, name, fam, cls, k, maxlen, single, cases, seed, groups=5, **pk):
    rng = np.random.default_rng(seed)
    param = make_param(k, **pk)
    ins, outs, ist, ost = [], [], [0], [0]
    for _ in range(cases):
        recs = fuzz_partition(rng, k, maxlen, groups, single)
        out = run_record_op(fam, cls, param, recs, single)
        ins += recs
        outs += out
        ist.append(len(ins))
        ost.append(len(outs))
    store.records(name + "/in", ins, ist)
    store.records(name + "/out", outs, ost)
    store.put(name + "/meta", np.array([k, maxlen, int(single)] + [int(v) for v in pk.values()], np.int64))


def fuzz_fork(store, name, fam, cls, k, reflected, cases, seed, **pk):
    """runs of 1-5 equal keys with tied coverages; the coverage sits in `left` for the forward filter's input (left = right =
    count) and the reflected filter compares `left` too (P/ReflexivDSMain.java:3493-3537)"""
    rng = np.random.default_rng(seed)
    param = make_param(k, **pk)
    sub = k - 1
    kw = (sub - 1) // 31 + 1 if k > 32 else 1
    ins, outs, ist, ost = [], [], [0], [0]
    for _ in range(cases):
        recs = []
        for _g in range(int(rng.integers(1, 9))):
            keyseq = rng.integers(0, 4, sub)
            for _r in range(int(rng.integers(1, 6))):
                cov = int(rng.choice([1, 2, 2, 3, 4, 8, 9, 20, 40]))
                base = int(rng.integers(0, 4))
                if reflected:
                    rec = (key_words(keyseq, kw), 2, ((1 << 2) | base,), cov, int(rng.choice([-1, -1 - cov, sub])))
                else:
                    rec = (key_words(keyseq, kw), 1, ((1 << 2) | base,), cov, cov)
                recs.append(rec)
        recs = sort_records(recs)
        out = run_record_op(fam, cls, param, recs, True)
        ins += recs
        outs += out
        ist.append(len(ins))
        ost.append(len(outs))
    store.records(name + "/in", ins, ist)
    store.records(name + "/out", outs, ost)
    store.put(name + "/meta", np.array([k, int(reflected)] + [int(v) for v in pk.values()], np.int64))


# ----------------------------------------------------------------------------------------------- main
def main():
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(HERE, "reference_vectors.npz")     # (--out: tests/test_java2py.py regenerates into a scratch directory)
    st = Store()
    rng = np.random.default_rng(20261004)

    # chains, k <= 31: both twins, with and without error correction, P in {1, 2, 3}
    g = synth_genome(rng, 420)
    reads = synth_reads(rng, g, 70, 14, 0.01, second_allele=(200, (int(g[200]) + 1) % 4))
    lines = fastq_lines(reads)
    for fam in ("ds", "rdd"):
        for (k, P, mec) in ((31, 2, 8), (31, 3, 0), (25, 1, 8)):
            nm = f"chain_{fam}_k{k}_P{P}_e{mec}"
            tr, text = chain_k31(st, nm, fam, k, P, 2, mec, lines)
            print(nm, "trace", tr[:6], "...", len(tr), "passes;", [t for t in text if t.startswith(">")], flush=True)

    # the documented example (docs/example.html:303: k = 31, -cover 3) through the RDD twin's own classes, P = 4: only the
    # trace and the contig text are stored (the test compares them with the documentation and with the oracle)
    import gzip
    ex_lines = []
    for f in ("paired_dat1.fq.gz", "paired_dat2.fq.gz"):
        with gzip.open(os.path.join(os.environ.get("RFX_REFERENCE", "/root/reference"), "example", f), "rt") as fh:
            ex_lines += [ln.rstrip("\n") for ln in fh]
    st.skip = ("example_",)
    for fam in ("rdd", "ds"):
        tr, text = chain_k31(st, f"example_{fam}_k31_P4", fam, 31, 4, 3, 8, ex_lines, min_contig=500)
        print(f"example_{fam}", "trace", tr, [t for t in text if t.startswith(">")], flush=True)
        del st.d[f"example_{fam}_k31_P4/fastq"], st.d[f"example_{fam}_k31_P4/instances"]
    st.skip = ()

    # chains, k > 31
    g2 = synth_genome(rng, 520)
    reads2 = synth_reads(rng, g2, 110, 12, 0.006, second_allele=(260, (int(g2[260]) + 2) % 4))
    for (k, P, mec) in ((63, 2, 8), (47, 1, 0), (95, 3, 8)):
        nm = f"chain_ds64_k{k}_P{P}_e{mec}"
        tr, text = chain_k64(st, nm, k, P, 2, mec, reads2)
        print(nm, "trace", tr[:6], "...", len(tr), "passes;", [t for t in text if t.startswith(">")], flush=True)

    # fuzz: the three extend stages of the three families
    seed = 1000
    for fam, names in (("ds", ("DSExtendReflexivKmer", "DSExtendReflexivKmerToArrayFirstTime", "DSExtendReflexivKmerToArrayLoop")),
                       ("rdd", ("ExtendReflexivKmer", "ExtendReflexivKmerToArrayFirstTime", "ExtendReflexivKmerToArrayLoop")),
                       ("ds64", ("DSExtendReflexivKmer", "DSExtendReflexivKmerToArrayFirstTime", "DSExtendReflexivKmerToArrayLoop"))):
        ks = (63, 47, 95) if fam == "ds64" else (31,)
        for k in ks:
            for cls, maxlen, single, cases in ((names[0], 15, True, 120), (names[1], 16, True, 160), (names[2], 130, False, 160)):
                seed += 1
                fuzz_set(st, f"fuzz_{fam}_k{k}_{cls}", fam, cls, k, maxlen, single, cases, seed)
                print("fuzz", fam, k, cls, flush=True)
    # the array loop started at marker 1 (param.scramble == 3, P/ReflexivDSMain64.java:7484-7486)
    fuzz_set(st, "fuzz_ds64_k63_DSExtendReflexivKmerToArrayLoop_scr3", "ds64", "DSExtendReflexivKmerToArrayLoop", 63, 130, False, 80,
             seed + 50, scramble=3)
    # fork filters, all four classes of each family
    for fam, names in (("ds", ("DSFilterForkSubKmer", "DSFilterForkSubKmerWithErrorCorrection", "DSFilterForkReflectedSubKmer",
                               "DSFilterForkReflectedSubKmerWithErrorCorrection")),
                       ("rdd", ("FilterForkSubKmer", "FilterForkSubKmerWithErrorCorrection", "FilterForkReflectedSubKmer",
                                "FilterForkReflectedSubKmerWithErrorCorrection")),
                       ("ds64", ("DSFilterForkSubKmer", "DSFilterForkSubKmerWithErrorCorrection", "DSFilterForkReflectedSubKmer",
                                 "DSFilterForkReflectedSubKmerWithErrorCorrection"))):
        k = 63 if fam == "ds64" else 31
        for i, cls in enumerate(names):
            seed += 1
            mec = 8 if "ErrorCorrection" in cls else 0
            fuzz_fork(st, f"fuzz_{fam}_k{k}_{cls}", fam, cls, k, i >= 2, 120, seed, minErrorCoverage=mec)
            print("fuzz", fam, k, cls, flush=True)
    # the k > 31 from-counts extras on random sorted partitions of array records
    for cls in ("DSReflexivAndForwardKmer", "DSFilterExtendableKmerPairs", "DSFilterUnExtendableKmer",
                "DSFilterStillExtendableKmerFromPairs", "DSFilterStillExtendableKmerEnds", "DSFilterUnExtendableKmerLeftEnds",
                "DSFilterUnExtendableKmerRightEnds"):
        for k in (63, 95):
            seed += 1
            fuzz_set(st, f"fuzz_ds64_k{k}_{cls}", "ds64", cls, k, 130, False, 100, seed)
            print("fuzz ds64", k, cls, flush=True)

    np.savez_compressed(out_path, **st.d)
    h = hashlib.sha256(open(out_path, "rb").read()).hexdigest()
    print(f"wrote {out_path}: {len(st.d)} arrays, {os.path.getsize(out_path)} bytes, sha256 {h}")


if __name__ == "__main__":
    main()
