#!/usr/bin/env python3
"""Golden vectors for contig RC de-duplication made by the REFERENCE'S OWN classes.

P/ReflexivDSDynamicKmerDedup.java's operator classes are translated mechanically (tools/java2py.py, from the reference's
source text at generation time) and driven in the order of its driver (`assemblyFromKmer`, :138-339): three rounds of
marker k-mer extraction -> sort -> DSMarkerKmerSelection -> groupBy().count() >= 2 -> DSMarkerKmerShorterID -> union ->
sort -> DSShorterRCContigSeqAndTargetExtraction -> sort -> removal class; the last round writes text
(TagRowContigDSID).  What sits between two classes is Spark's; here it follows the order contract of DESIGN.md:
ONE logical partition, every sort stable on the SIGNED 64-bit column (Spark's LongType order), union = left rows then
right rows, groupBy().count() rows in ascending key order, zipWithIndex = position.

Output: tests/golden/dedup_vectors.npz -- per case the input contigs, the surviving contigs after each round and the
final text; for case 0 also every intermediate row set."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, HERE)
import java2py as jp  # noqa: E402
from make_reference_vectors import make_param, drain, u64  # noqa: E402

REF = os.environ.get("RFX_REFERENCE", "/root/reference") + "/src/main/java/uni/bielefeld/cmg/reflexiv/pipeline/"
CLASSES = ["DynamicKmerBinarizerFromReducedToSubKmer", "DSShorterForwardAndRCContigRemovalArray",
           "DSShorterForwardAndRCContigRemoval", "DSShorterRCContigRemoval", "DSMarkerKmerSelection", "DSArrayTupleToDataset",
           "DSTupleToDataset", "ForwardAndReverseComplementKmerMarkerExtraction", "ReverseComplementKmerMarkerExtraction",
           "DSShorterRCContigSeqAndTargetExtraction", "DSMarkerKmerShorterID", "TagRowContigDSID"]
_cls = {}


def op(name, param):
    if not _cls:
        _cls.update(jp.translate_classes(REF + "ReflexivDSDynamicKmerDedup.java", CLASSES))
    return _cls[name](jp.Outer(param, _cls))


def blocks_of(x):
    items = x.items if isinstance(x, jp.Seq) else x
    return tuple(u64(w.v) for w in items)


def sgn(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def blocks_to_seq(b):
    """left-aligned 31-base blocks with a trailing 01 terminator -> ACGT string"""
    n = (len(b) - 1) * 31
    last = b[-1]
    tz = (last & -last).bit_length() - 1
    n += 32 - tz // 2 - 1
    return "".join("ACGT"[(b[i // 31] >> (2 * (31 - i % 31))) & 3] for i in range(n))


def dedup_pipeline(contigs, trace=None):
    """contigs: list of ACGT strings (the path's contigs, in order) -> (rounds: list of list of strings, text)"""
    param = make_param(31)
    T = (lambda k, v: trace.__setitem__(k, v)) if trace is not None else (lambda k, v: None)
    rows0 = [jp.Row([f"Contig-{len(s)}-{i}", s]) for i, s in enumerate(contigs)]
    tup = drain(op("DynamicKmerBinarizerFromReducedToSubKmer", param).call(jp.JIter(rows0)))        # (long[] blocks, long id)
    rounds = []
    text = None
    for rnd in (1, 2, 3):
        ext = "ReverseComplementKmerMarkerExtraction" if rnd == 1 else "ForwardAndReverseComplementKmerMarkerExtraction"
        marker = drain(op(ext, param).call(jp.JIter(tup)))                                          # (long kmer, long attribute)
        T(f"r{rnd}/markers", [(u64(r.vals[0].v), u64(r.vals[1].v)) for r in marker])
        marker = sorted(marker, key=lambda r: r.vals[0].v)                                          # sort("kmerBinary"), signed, stable
        pairs = drain(op("DSMarkerKmerSelection", param).call(jp.JIter(marker)))                    # (long shorter<<32|longer, 1)
        T(f"r{rnd}/pairs", [u64(r.vals[0].v) for r in pairs])
        cnt = {}
        for r in pairs:
            cnt[r.vals[0].v] = cnt.get(r.vals[0].v, 0) + 1
        idcount = [jp.Row([jp._L(x), jp._L(c)]) for x, c in sorted(cnt.items()) if c >= 2]          # groupBy.count, >= 2
        T(f"r{rnd}/candidates", [u64(r.vals[0].v) for r in idcount])
        shortid = drain(op("DSMarkerKmerShorterID", param).call(jp.JIter(idcount)))                 # (long[]{-1, target}, long shorter)
        union = list(tup) + list(shortid)
        union = sorted(union, key=lambda r: r.vals[1].v)                                            # sort("count")
        st = drain(op("DSShorterRCContigSeqAndTargetExtraction", param).call(jp.JIter(union)))
        st = sorted(st, key=lambda r: r.vals[1].v)                                                  # sort("count")
        T(f"r{rnd}/targets", [(blocks_of(r.vals[0]), sgn(r.vals[1].v)) for r in st])
        if rnd == 1:
            kept = drain(op("DSShorterRCContigRemoval", param).call(jp.JIter(st)))
        elif rnd == 2:
            kept = drain(op("DSShorterForwardAndRCContigRemovalArray", param).call(jp.JIter(st)))
        else:
            strings = drain(op("DSShorterForwardAndRCContigRemoval", param).call(jp.JIter(st)))
            rounds.append(list(strings))
            tag = op("TagRowContigDSID", param)
            lines = []
            for i, s in enumerate(strings):
                lines += drain(tag.call(jp.JTuple(s, jp._L(i))))
            text = "".join(ln + "\n" for ln in lines)
            break
        rounds.append([blocks_to_seq(blocks_of(r.vals[0])) for r in kept])
        cvt = "DSTupleToDataset" if rnd == 1 else "DSArrayTupleToDataset"                           # zipWithIndex + to (blocks, id)
        tup = drain(op(cvt, param).call(jp.JIter([jp.JTuple(r, jp._L(i)) for i, r in enumerate(kept)])))
    return rounds, text


COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def rand_seq(rng, n):
    return "".join("ACGT"[b] for b in rng.integers(0, 4, n))


def make_case(rng, kind):
    """a contig set as the fixed-k path would emit it: most sequences on both strands, some with overhangs, contained
    pieces, near-identical copies; lengths around the 300 / 2000 / 4000 thresholds of the probe layouts"""
    contigs = []
    if kind == "both_strands":
        for L in (4558, 1200, 640, 2100, 310, 299, 4100):
            s = rand_seq(rng, L)
            contigs += [s, rc(s)]
    elif kind == "overhangs":
        for L in (3000, 5200, 900, 2500):
            s = rand_seq(rng, L)
            a, b = int(rng.integers(0, 120)), int(rng.integers(0, 120))
            contigs.append(s)
            contigs.append(rc(rand_seq(rng, a) + s[200:L - 150] + rand_seq(rng, b)))    # RC of an inner piece with new flanks
            contigs.append(s[50:L // 2])                                                 # a forward piece
        contigs.append(rand_seq(rng, 700))
    elif kind == "mutated":
        for L in (4500, 2300, 1500, 800, 400):
            s = rand_seq(rng, L)
            t = list(rc(s))
            for p in rng.integers(0, L, max(1, L // 400)):
                t[p] = "ACGT"[(("ACGT".index(t[p])) + 1) % 4]
            contigs += [s, "".join(t)]
        s = rand_seq(rng, 1000)
        contigs += [s, s, rc(s)]                                                         # exact copies: equal length, the earlier id wins
    elif kind == "shuffled":
        base = []
        for L in rng.integers(300, 6000, 14):
            s = rand_seq(rng, int(L))
            base += [s, rc(s)]
        base += [rand_seq(rng, int(L)) for L in rng.integers(100, 1500, 6)]
        order = rng.permutation(len(base))
        contigs = [base[i] for i in order]
    return contigs


def pack_strings(strs):
    off = np.zeros(len(strs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in strs])
    return np.frombuffer("".join(strs).encode(), np.uint8), off


def run_case(arg):
    ci, kind, contigs = arg
    trace = {} if ci in (0, 1) else None
    rounds, text = dedup_pipeline(contigs, trace)
    return ci, kind, contigs, trace, rounds, text


def main():
    rng = np.random.default_rng(20261005)
    out = {}
    # the inputs first (they share one generator, in this order); the cases are then independent: --jobs N runs them in N
    # processes (tests/test_java2py.py regenerates with 5), the arrays are the same either way
    cases = [(ci, kind, make_case(rng, kind)) for ci, kind in enumerate(("both_strands", "overhangs", "mutated", "shuffled", "shuffled"))]
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 1
    if jobs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(jobs) as pool:
            results = pool.map(run_case, sorted(cases, key=lambda c: -sum(map(len, c[2]))), chunksize=1)
        results.sort(key=lambda r: r[0])
    else:
        results = [run_case(c) for c in cases]
    for ci, kind, contigs, trace, rounds, text in results:
        nm = f"case{ci}_{kind}"
        out[nm + "/in"], out[nm + "/in_off"] = pack_strings(contigs)
        for r, strs in enumerate(rounds):
            out[f"{nm}/round{r + 1}"], out[f"{nm}/round{r + 1}_off"] = pack_strings(strs)
        out[nm + "/text"] = np.frombuffer(text.encode(), np.uint8)
        if trace is not None:
            for rnd in (1, 2, 3):
                out[f"{nm}/r{rnd}_markers"] = np.array(trace[f"r{rnd}/markers"], np.uint64).reshape(-1, 2)
                out[f"{nm}/r{rnd}_pairs"] = np.array(trace[f"r{rnd}/pairs"], np.uint64)
                out[f"{nm}/r{rnd}_candidates"] = np.array(trace[f"r{rnd}/candidates"], np.uint64)
        print(nm, len(contigs), "contigs", sum(map(len, contigs)), "bases ->", [len(r) for r in rounds],
              sum(len(s) for s in rounds[-1]), "bases", flush=True)
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(HERE, "dedup_vectors.npz")     # (--out: tests/test_java2py.py regenerates into a scratch directory)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", hashlib.sha256(open(path, "rb").read()).hexdigest())


if __name__ == "__main__":
    main()
