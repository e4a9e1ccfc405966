"""Pins the CPU oracle: the reference's one documented known answer
(docs/example.html:303,320-343) and the SURVEY D.1 counts, then cross-checks the
C oracle against the independent string-level model (tests/pymodel.py) and the
committed golden fixtures."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import pymodel as M


@pytest.fixture(scope="module")
def ex(golden_dir):
    return np.load(os.path.join(golden_dir, "example.npz"))


@pytest.fixture(scope="module")
def planted(golden_dir):
    return np.load(os.path.join(golden_dir, "planted.npz"))


def split_contigs(text):
    out, cur = [], None
    for line in text.split("\n"):
        if line.startswith(">"):
            cur = [line, []]
            out.append(cur)
        elif line:
            cur[1].append(line)
    return [(h, "".join(s)) for h, s in out]


def test_counts_match_survey(ex):
    km = O.extract_canon(ex["bases"], ex["read_off"], 31)
    assert len(km) == 161_000 == int(ex["n_instances"])
    for mc, want in ((1, 43_748), (2, 5_768), (3, 4_612)):
        keys, counts, nd = O.count_filter(km, mc, 10_000_000)
        assert nd == 43_748
        assert len(keys) == want
    keys, counts, _ = O.count_filter(km, 3, 10_000_000)
    assert np.array_equal(keys, ex["keys_cov3"]) and np.array_equal(counts, ex["counts_cov3"])
    assert np.all(keys[1:] > keys[:-1])
    assert np.array_equal(O.extract_canon(ex["bases"][:ex["read_off"][4]], ex["read_off"][:5], 31),
                          ex["k1_first4"])


def test_extract_matches_naive_definition(ex):
    """canonical = min(kmer, revcomp) over every window, N/other -> T (ReflexivMain.java:3062-3074)."""
    bases = bytes(ex["bases"][: ex["read_off"][3]]).decode()
    code = {"A": 0, "C": 1, "G": 2}
    want = []
    for r in range(3):
        read = bases[ex["read_off"][r]: ex["read_off"][r + 1]]
        for i in range(len(read) - 30):
            v = [code.get(c, 3) for c in read[i:i + 31]]
            f = 0
            for x in v:
                f = (f << 2) | x
            rc = 0
            for x in reversed(v):
                rc = (rc << 2) | (3 - x)
            want.append(min(f, rc))
    got = O.extract_canon(ex["bases"][: ex["read_off"][3]], ex["read_off"][:4], 31)
    assert [int(x) for x in got] == want


def test_short_reads_dropped():
    """len - k - endClip <= 1 drops reads of length k and k+1 (ReflexivMain.java:3020)."""
    for ln, want in ((31, 0), (32, 0), (33, 3), (34, 4)):
        b = np.frombuffer(("ACGT" * 10)[:ln].encode(), np.uint8)
        assert len(O.extract_canon(b, np.array([0, ln]), 31)) == want
    b = np.frombuffer(("ACGT" * 10)[:36].encode(), np.uint8)
    assert len(O.extract_canon(b, np.array([0, 36]), 31, 2, 1)) == 3      # clips: 33 bases used


def test_documented_known_answer(ex):
    """`reflexiv run -kmer 31 -cover 3` on the example: two part files of 4619 bytes, each one
    contig >Contig-4558-0 whose first 1200 bases are printed in docs/example.html:331-343."""
    prm = O.default_params(min_cov=3, partitions=4, twin=O.TWIN_RDD)
    text, nc, trace, rec = O.assemble_from_counts(ex["keys_cov3"], ex["counts_cov3"], prm)
    contigs = split_contigs(text)
    assert nc == 2 and [len(s) for _, s in contigs] == [4558, 4558]
    prefix = str(ex["doc_prefix1200"])
    hit = [s for _, s in contigs if s.startswith(prefix)]
    assert len(hit) == 1
    assert hashlib.sha256(hit[0].encode()).hexdigest() == \
        "245baebd8b5b681f639217f31d647d9fcd03adfeef7f6d5ef10edd5cc12ae62c"
    other = [s for _, s in contigs if s is not hit[0]][0]
    assert other == M.revcomp(hit[0])
    assert hashlib.sha256(other.encode()).hexdigest() == \
        "66c80454f18483e7be6ad9dbc178c9f1a8be54a14d341982c7297a9c13327f60"
    # one contig per part file: header + 4558 bases wrapped at 100 + newlines = 4619 bytes
    one = O.contigs_text(O.gather(rec, np.array([int(np.argmax(np.diff(rec.ext_off)))])), 31, 500, O.TWIN_RDD)[0]
    assert one.startswith(str(ex["doc_header"]) + "\n")
    assert len(one) == int(ex["doc_part_bytes"]) == 4619


def test_golden_contigs_and_traces(ex):
    for P in (1, 2, 4, 8):
        for twin, tn in ((O.TWIN_DS, "ds"), (O.TWIN_RDD, "rdd")):
            prm = O.default_params(min_cov=3, partitions=P, twin=twin)
            text, nc, trace, _ = O.assemble_from_counts(ex["keys_cov3"], ex["counts_cov3"], prm)
            assert text == str(ex[f"contigs_{tn}_P{P}"])
            assert trace == [int(x) for x in ex[f"trace_{tn}_P{P}"]]
    # SURVEY D.2 (P=1): 731 records after pass index 8, then 560, 424, 322
    t1 = [int(x) for x in ex["trace_ds_P1"]]
    assert t1[8:12] == [731, 560, 424, 322]
    # 9224 -> 9219 -> 9214 and no fork-marked record on the example
    assert len(ex["k5_key"]) == 9224 and len(ex["k6_key"]) == 9219 and len(ex["k8_key"]) == 9214
    assert not ((ex["k8_left"] >= 0) | (ex["k8_right"] >= 0)).any()


def to_model(r: O.Records, k):
    """C-oracle records (reference word layout) -> pymodel string tuples."""
    sub = k - 1
    out = []
    for key, mk, words, left, right in r.tuple_list():
        f = 32 - ((64 - words[0].bit_length()) // 2 + 1)
        s = M.decode_kmer(words[0] & ((1 << (2 * f)) - 1), f)
        for w in words[1:]:
            s += M.decode_kmer(w, 31)
        out.append((M.decode_kmer(key, sub), mk, s, left, right))
    return out


@pytest.mark.parametrize("case,k,P,twin", [("example", 31, 4, O.TWIN_DS), ("example", 31, 3, O.TWIN_RDD),
                                           ("planted", 31, 4, O.TWIN_DS), ("planted", 31, 4, O.TWIN_RDD),
                                           ("planted", 21, 5, O.TWIN_DS)])
def test_c_oracle_equals_string_model(ex, planted, case, k, P, twin):
    if case == "example":
        keys, counts = ex["keys_cov3"], ex["counts_cov3"]
    else:
        keys, counts = planted[f"k{k}_keys"], planted[f"k{k}_counts"]
    ds = twin == O.TWIN_DS
    sub = k - 1
    # operator by operator
    r = O.sort_records(O.rc_expand_subkmer(keys, counts, k))
    m = M.stable_sort(M.rc_expand([int(x) for x in keys], [int(c) for c in counts], k))
    r, _ = O.fork_filter_forward(r, O.partition_starts(r.key, P), k, 8, twin)
    m, _ = M.fork_forward(m, M.partition_starts([x[0] for x in m], P), sub, 8, ds)
    assert [(M.decode_kmer(a, sub), b, M.NUC[c[0]], d, e) for a, b, c, d, e in r.tuple_list()] == m
    r = O.sort_records(O.reflect_from_forward(r, k))
    m = M.stable_sort(M.reflect(m))
    r, ps = O.fork_filter_reflected(r, O.partition_starts(r.key, P), k, 8, twin)
    m, ms = M.fork_reflected(m, M.partition_starts([x[0] for x in m], P), sub, 8, ds)
    assert to_model(r, k) == m and list(ps) == ms
    r = O.random_reflection(r, ps, k)
    m = M.random_reflection(m, ms, sub)
    assert to_model(r, k) == m
    for _ in range(12):
        r = O.sort_records(r)
        r, _ = O.extend_pass(r, O.partition_starts(r.key, P), k, twin)
        m = M.stable_sort(m)
        m, _ = M.extend_pass(m, M.partition_starts([x[0] for x in m], P), sub, rdd=not ds)
        assert to_model(r, k) == m
    # whole driver
    prm = O.default_params(k=k, min_cov=2, partitions=P, twin=twin, min_contig=100)
    _, _, trace, rec = O.assemble_from_counts(keys, counts, prm)
    mt = []
    mrec = M.assemble([int(x) for x in keys], [int(c) for c in counts], k, P, 8, ds=ds, trace=mt)
    assert trace == mt and to_model(rec, k) == mrec


def test_planted_exercises_bubble_branches(planted):
    """The example never marks a fork; the planted SNP/repeat case must."""
    for tn in ("ds", "rdd"):
        l, r = planted[f"k31_{tn}_k8_left"], planted[f"k31_{tn}_k8_right"]
        assert ((l >= 0) | (r >= 0)).sum() >= 4
    assert str(planted["k31_ds_contigs"]) != str(planted["k31_rdd_contigs"])


def test_planted_golden(planted):
    keys, counts = planted["k31_keys"], planted["k31_counts"]
    for twin, tn in ((O.TWIN_DS, "ds"), (O.TWIN_RDD, "rdd")):
        prm = O.default_params(k=31, min_cov=2, partitions=4, twin=twin, min_contig=100)
        text, _, trace, _ = O.assemble_from_counts(keys, counts, prm)
        assert text == str(planted[f"k31_{tn}_contigs"])
        assert trace == [int(x) for x in planted[f"k31_{tn}_trace"]]


def test_fastq_grouping_state_machine():
    """quality lines starting with '@' stay inside a record; stray lines are skipped
    (ReflexivMain.java:3092-3112)."""
    text = b"junk\n@r1\nACGT\n+\n@@@@\n@r2\nGGCC\n+r2\nIIII\nstray\n@r3\nTT\n"
    off, ln = O.fastq_group(text)
    assert [text[o:o + l] for o, l in zip(off, ln)] == [b"ACGT", b"GGCC"]


def test_synth_generator_is_deterministic_and_strand_balanced():
    g = O.synth_genome(7, 4096)
    assert np.array_equal(g, O.synth_genome(7, 4096)) and not np.array_equal(g, O.synth_genome(8, 4096))
    a, off = O.synth_reads(7, g, 4096, 0, 200, 150)
    b, _ = O.synth_reads(7, g, 4096, 100, 100, 150)
    assert np.array_equal(a[100 * 150:], b)              # counter-based: any window reproduces
    assert set(np.unique(a)) <= set(b"ACGT")
    clean, _ = O.synth_reads(7, g, 4096, 0, 200, 150, err_per_2_32=0)
    diff = (a != clean).mean()
    assert 0.001 < diff < 0.012                           # ~0.5 % substitutions


def test_counter_line_filter_only_seq():
    """DSFastqFilterOnlySeq (P/ReflexivDataFrameCounter.java:238-290): a line survives iff it is longer
    than 20, starts with neither '@' nor '+', and shows A/T/C/G/N at positions 0, 4, 9, 14, 19 --
    a quality line that passes is kept, a lower-case or short read is dropped."""
    lines = [b"@r1", b"ACGTACGTACGTACGTACGTACGT", b"+", b"IIIIIIIIIIIIIIIIIIIIIIII",       # plain record
             b"@r2", b"ACGTACGTACGTACGTACGT", b"+", b"IIIIIIIIIIIIIIIIIIII",               # 20 bases: too short
             b"@r3", b"acgtacgtacgtacgtacgtacgt", b"+", b"AIIIAIIIICIIIIGIIIITIIII",       # lower case dropped; quality passes!
             b"@r4", b"NCGTNCGTANGTACNTACGNACGTAC", b"+r4", b"@IIIIIIIIIIIIIIIIIIIIIIIII",  # N allowed; '@' quality dropped
             b"ACGTACGTACGTACGTACGXACGT"]                                                   # X at position 19
    text = b"\n".join(lines) + b"\n"
    off, ln = O.fastq_only_seq(text)
    got = [text[o:o + l] for o, l in zip(off, ln)]
    assert got == [lines[1], lines[11], lines[13]]
    # on the example reads both filters agree (2300 reads)
    import gzip
    for f in ("/root/reference/example/paired_dat1.fq.gz",):
        if os.path.exists(f):
            t = gzip.open(f, "rb").read()
            o1, l1 = O.fastq_only_seq(t)
            o2, l2 = O.fastq_group(t)
            assert np.array_equal(o1, o2) and np.array_equal(l1, l2)


@pytest.mark.parametrize("P", [16, 64, 200])
def test_coalesce_rule_matches_string_model(ex, P):
    """P/ReflexivMain.java:277-281: from 16 partitions on, a check that finds <= 20 records per partition cuts the
    partition number to P/4+1 -- which moves every logical partition boundary of the later passes"""
    from tests import pymodel as M
    keys, counts = ex["keys_cov3"], ex["counts_cov3"]
    for coalesce in (0, 1):
        prm = O.default_params(min_cov=3, partitions=P, coalesce=coalesce, min_contig=100)
        text, nc, trace, rec = O.assemble_from_counts(keys, counts, prm)
        wtrace = []
        want = M.assemble([int(x) for x in keys], [int(c) for c in counts], 31, P, 8, 15, 150, True, wtrace, bool(coalesce))
        assert trace == wtrace, coalesce
        got = [(M.decode_kmer(int(rec.key[i]), 30), int(rec.marker[i])) for i in range(rec.n)]
        assert got == [(r[0], r[1]) for r in want]
    # the rule really fires on this input: the traces with and without it differ
    t0 = O.assemble_from_counts(keys, counts, O.default_params(min_cov=3, partitions=P, coalesce=0))[2]
    t1 = O.assemble_from_counts(keys, counts, O.default_params(min_cov=3, partitions=P, coalesce=1))[2]
    assert t0 != t1 or P == 16
