#!/usr/bin/env python3
"""Golden vectors for the dynamic-k record format and passes made by the REFERENCE'S OWN classes.

P/ReflexivDSDynamicKmerFirstFour.java and P/ReflexivDSDynamicKmerIteration.java (the "meta" assembler's first four
single-long passes and its array-loop passes on the third record layout: keys of DIFFERENT lengths as left-aligned 31-base
blocks with a 01 terminator, a packed attribute long, extensions left-aligned too) are translated mechanically
(tools/java2py.py) and driven in the order of their drivers:

  FirstFour.assemblyFromKmer (:137-224):  rows (k-mer text, "marker|left|right") -> DynamicKmerBinarizerFromReducedToSubKmer
      -> DSkmerRandomReflection -> sort("k-1") -> DSExtendReflexivKmer x 4 (a sort before each) ->
      DSBinarySubKmerWithShortExtensionToString -> rows (sub-k-mer text, "marker|left|right", extension text)
  Iteration.assemblyFromKmer (:134-205):   those rows -> DynamicKmerBinarizerFromReducedToSubKmer (:...) ->
      [sort("k-1") -> DSExtendReflexivKmerToArrayLoop] x (endIteration - startIteration + 1) ->
      DSBinarySubKmerWithLongExtensionToString -> rows

Between two classes: the order contract (stable sort of the array<long> column as Spark orders it -- element by element
as SIGNED longs, a shorter array first when it is a prefix --, P logical partitions cut at floor(p*n/P) moved forward past
equal keys, a fresh operator instance per partition).

Output: tests/golden/dynamic_vectors.npz -- per case the input rows, the rows after every operator (as text rows) and the
final rows."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, HERE)
import java2py as jp  # noqa: E402
from make_reference_vectors import make_param, drain, u64, partition_starts  # noqa: E402

REF = os.environ.get("RFX_REFERENCE", "/root/reference") + "/src/main/java/uni/bielefeld/cmg/reflexiv/pipeline/"
_cls = {}


def classes(which):
    if which not in _cls:
        f, names = {"ff": ("ReflexivDSDynamicKmerFirstFour.java",
                           ["DynamicKmerBinarizerFromReducedToSubKmer", "DSkmerRandomReflection", "DSExtendReflexivKmer",
                            "DSBinarySubKmerWithShortExtensionToString"]),
                    "it": ("ReflexivDSDynamicKmerIteration.java",
                           ["DynamicKmerBinarizerFromReducedToSubKmer", "DSExtendReflexivKmerToArrayLoop",
                            "DSBinarySubKmerWithLongExtensionToString"])}[which]
        _cls[which] = jp.translate_classes(REF + f, names)
    return _cls[which]


def op(which, name, param):
    c = classes(which)
    return c[name](jp.Outer(param, c))


def sgn(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def key_of(row):
    k = row.vals[0]
    items = k.items if isinstance(k, jp.Seq) else k
    return tuple(w.v for w in items)                     # signed longs, as Spark compares them


def spark_array_order(a, b):
    """ordering of array<long>: element by element (signed), a proper prefix first"""
    for x, y in zip(a, b):
        if x != y:
            return -1 if x < y else 1
    return (len(a) > len(b)) - (len(a) < len(b))


def sort_rows(rows):
    import functools
    return sorted(rows, key=functools.cmp_to_key(lambda r, s: spark_array_order(key_of(r), key_of(s))))


def as_seq_rows(rows):
    """rows as the next Spark stage would read them: arrays come back as Seq"""
    out = []
    for r in rows:
        v = list(r.vals)
        for i in (0, 2):
            if isinstance(v[i], list):
                v[i] = jp.Seq(v[i])
        out.append(jp.Row(v))
    return out


def run_partitions(which, name, param, rows, P, sort=True):
    if sort:
        rows = sort_rows(rows)
    st = partition_starts([key_of(r) for r in rows], P)
    out = []
    for p in range(P):
        out += drain(op(which, name, param).call(jp.JIter(as_seq_rows(rows[st[p]:st[p + 1]]))))
    return out


def to_text(which, param, rows):
    name = "DSBinarySubKmerWithShortExtensionToString" if which == "ff" else "DSBinarySubKmerWithLongExtensionToString"
    return [(r.vals[0], r.vals[1], r.vals[2]) for r in drain(op(which, name, param).call(jp.JIter(as_seq_rows(rows))))]


def first_four(in_rows, P, trace):
    """in_rows: [(kmer text, "m|l|r")] -> text rows after the four passes"""
    param = make_param(31)
    rows = drain(op("ff", "DynamicKmerBinarizerFromReducedToSubKmer", param).call(jp.JIter([jp.Row([a, b]) for a, b in in_rows])))
    trace.append(("binarized", to_text("ff", param, rows)))
    st = partition_starts(list(range(len(rows))), P)                     # (the input file's partitions: equal shares)
    out = []
    for p in range(P):
        out += drain(op("ff", "DSkmerRandomReflection", param).call(jp.JIter(rows[st[p]:st[p + 1]])))
    rows = out
    trace.append(("random_reflection", to_text("ff", param, rows)))
    for it in range(4):
        rows = run_partitions("ff", "DSExtendReflexivKmer", param, rows, P)
        trace.append((f"extend{it}", to_text("ff", param, rows)))
    return to_text("ff", param, rows)


def iterations(text_rows, P, start, end, trace):
    """text rows (sub-k-mer, attribute, extension) -> text rows after iterations start..end (Iteration.assemblyFromKmer)"""
    param = make_param(31, startIteration=start, endIteration=end)
    rows = drain(op("it", "DynamicKmerBinarizerFromReducedToSubKmer", param).call(jp.JIter([jp.Row(list(t)) for t in text_rows])))
    trace.append(("it_binarized", to_text("it", param, rows)))
    it = start
    while it <= end:
        it += 1
        rows = run_partitions("it", "DSExtendReflexivKmerToArrayLoop", param, rows, P)
        trace.append((f"it_extend{it}", to_text("it", param, rows)))
    return to_text("it", param, rows)


COMP = str.maketrans("ACGT", "TGCA")


def make_input(rng, genome_len, ks, both_strands=True, cov_style=0):
    """k-mers of a random genome; the k in force changes along it (the reduction keeps the longest k-mer a region supports),
    so keys of different lengths meet where regions join and one key is a prefix of another"""
    g = "".join("ACGT"[b] for b in rng.integers(0, 4, genome_len))
    if genome_len > 400:
        g = g[:300] + g[40:110] + g[370:]                         # a repeat: forks
    rows = []
    seen = set()
    strands = [g, g.translate(COMP)[::-1]] if both_strands else [g]
    for s in strands:
        pos = 0
        while pos < len(s):
            k = int(rng.choice(ks))
            span = int(rng.integers(40, 160))
            for p in range(pos, min(pos + span, len(s) - k + 1)):
                km = s[p:p + k]
                if km in seen:
                    continue
                seen.add(km)
                if cov_style == 0:
                    left, right = -int(rng.integers(2, 40)), -int(rng.integers(2, 40))
                else:
                    left = int(rng.integers(0, 60)) if rng.random() < 0.25 else -int(rng.integers(2, 40))
                    right = int(rng.integers(0, 60)) if rng.random() < 0.25 else -int(rng.integers(2, 40))
                rows.append((km, f"1|{left}|{right}"))
            pos += span
    order = rng.permutation(len(rows))
    return [rows[i] for i in order]


def pack_rows(rows):
    text = "".join(",".join(r) + "\n" for r in rows)
    return np.frombuffer(text.encode(), np.uint8)


def main():
    rng = np.random.default_rng(20261006)
    out = {}
    cases = [("c0", 500, (23, 31, 41), 1, 0, (5, 9)), ("c1", 700, (23, 31, 41, 53, 67), 2, 1, (5, 9)),
             ("c2", 600, (31, 41, 95), 3, 1, (15, 19)), ("c3", 450, (23, 81, 95), 1, 0, (61, 64)),
             ("c4", 900, (23, 31), 2, 0, (5, 34))]             # thirty passes: extensions outgrow the keys
    for name, glen, ks, P, style, (start, end) in cases:
        in_rows = make_input(rng, glen, ks, cov_style=style)
        trace = []
        ff = first_four(in_rows, P, trace)
        fin = iterations(ff, P, start, end, trace)
        out[name + "/meta"] = np.array([P, start, end], np.int64)
        out[name + "/in"] = pack_rows(in_rows)
        for tag, rows in trace:
            out[f"{name}/{tag}"] = pack_rows(rows)
        out[name + "/final"] = pack_rows(fin)
        lens = sorted((len(r[0]) + len(r[2]) for r in fin), reverse=True)
        print(name, len(in_rows), "k-mers ->", len(ff), "after four passes ->", len(fin), "after iterations", (start, end), "longest", lens[:4], flush=True)
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(HERE, "dynamic_vectors.npz")     # (--out: tests/test_java2py.py regenerates into a scratch directory)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", hashlib.sha256(open(path, "rb").read()).hexdigest())


if __name__ == "__main__":
    main()
