"""k > 31 assembler (P/ReflexivDSMain64.java assemblyFromKmer): the oracle's multi-word restatement
(31 bases per key word) against the independent string-level model, operator by operator and through
the whole driver, on a planted genome (SNP bubble + repeat: both error-correction branches and the
bubble-distance rules fire)."""
import collections
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import pymodel as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
NUC = "ACGT"


def kmers_of_reads(bases, off, k, min_cov):
    """ascending canonical k-mers (strings) with count >= min_cov, by a plain string model"""
    text = bytes(bases).decode()
    cnt = collections.Counter()
    for r in range(len(off) - 1):
        s = "".join(c if c in "ACG" else "T" for c in text[off[r]:off[r + 1]])
        for p in range(len(s) - k + 1):
            f = s[p:p + k]
            rc = "".join(COMP[c] for c in reversed(f))
            cnt[f if f <= rc else rc] += 1
    keep = sorted(x for x, c in cnt.items() if c >= min_cov)
    return keep, [cnt[x] for x in keep]


def key_str(words, sub):
    """(k-1)-mer key words (31 bases per word, the last word the rest) -> string"""
    words = np.atleast_1d(words)
    kw = len(words)
    out = []
    for w in range(kw):
        nb = 31 if w < kw - 1 else sub - 31 * (kw - 1)
        x = int(words[w])
        assert x >> (2 * nb) == 0, "bits above the word's bases"
        out.append("".join(NUC[(x >> (2 * (nb - 1 - j))) & 3] for j in range(nb)))
    return "".join(out)


def ext_str(w, sentinel=True):
    w = [int(x) for x in w]
    if not sentinel:
        assert len(w) == 1 and w[0] < 4
        return NUC[w[0]]
    f = (w[0].bit_length() - 1) // 2
    assert w[0] >> (2 * f) == 1
    s = "".join(NUC[(w[0] >> (2 * (f - 1 - j))) & 3] for j in range(f))
    for x in w[1:]:
        assert x >> 62 == 0
        s += "".join(NUC[(x >> (2 * (30 - j))) & 3] for j in range(31))
    return s


def to_model(r: O.Records, sub, sentinel=True):
    return [(key_str(r.key[i], sub), int(r.marker[i]), ext_str(r.ext[r.ext_off[i]:r.ext_off[i + 1]], sentinel),
             int(r.left[i]), int(r.right[i])) for i in range(r.n)]


def planted_reads():
    d = np.load(os.path.join(GOLDEN, "planted.npz"))
    return d["bases"], d["read_off"]


def binarize(kmers, k):
    return np.stack([O.kmer_binarize_w(s, "1", k)[0] for s in kmers]) if kmers else np.zeros((0, O.asm_words(k)), np.uint64)


def test_word_counts_follow_default_param():
    # U/DefaultParam.java:84-85, 93-94
    for k in (32, 33, 47, 62, 63, 64, 93, 94, 95, 125):
        assert O.asm_words(k) == (k - 1) // 31 + 1
        assert O.sub_words(k) == (k - 2) // 31 + 1
    assert O.sub_words(31) == 1 and O.sub_words(32) == 1 and O.sub_words(33) == 2 and O.sub_words(63) == 2
    assert O.sub_words(64) == 3


def test_kmer_binarizer_rows():
    k = 63
    s = "ACGT" * 15 + "ACG"
    w, c = O.kmer_binarize_w(s, "17", k)
    assert c == 17 and len(w) == 3
    assert key_str(w[:2], 62) == s[:62] and int(w[2]) == NUC.index(s[62])
    # the legacy tuple text "(KMER,count)"  :10790-10806
    w2, c2 = O.kmer_binarize_w("(" + s, "17)", k)
    assert np.array_equal(w, w2) and c2 == 17
    # ten digits or more read as 1000000000
    assert O.kmer_binarize_w(s, "1234567890", k)[1] == 1000000000
    assert O.kmer_binarize_w(s, "123456789", k)[1] == 123456789
    assert O.kmer_binarize_w(s, "1234567890)", k)[1] == 1000000000
    assert O.kmer_binarize_w(s, "123456789)", k)[1] == 123456789
    # anything that is not A C G reads as T
    assert np.array_equal(O.kmer_binarize_w(s.replace("T", "N"), "1", k)[0], w)


@pytest.mark.parametrize("k", [33, 47, 63, 65, 95])
def test_counter_layout_to_assembler_layout(k):
    rng = np.random.default_rng(k)
    kmers = ["".join(rng.choice(list(NUC), size=k)) for _ in range(50)]
    bases = np.frombuffer("".join(kmers).encode(), np.uint8)
    off = np.arange(51, dtype=np.int64) * k
    w32 = O.extract_canon_w(bases, off, k)                      # one canonical k-mer per read
    w31 = O.counter_to_asm_w(w32, k)
    for a, b in zip(w32, w31):
        txt = O.kmer_text_w(a, k)
        assert np.array_equal(b, O.kmer_binarize_w(txt, "1", k)[0])
    # spot-check the words against the text
    txt = O.kmer_text_w(w32[0], k)
    W = O.asm_words(k)
    got = "".join(key_str(w31[0][i:i + 1], 31 if i < W - 1 else k - 31 * (W - 1)) for i in range(W))
    assert got == txt


@pytest.mark.parametrize("k,P,min_err", [(63, 4, 8), (63, 1, 0), (47, 3, 8), (33, 4, 8), (32, 2, 8), (62, 4, 8),
                                         (64, 4, 8), (93, 2, 8), (95, 4, 0)])
def test_operator_chain_matches_string_model(k, P, min_err):
    bases, off = planted_reads()
    kmers, counts = kmers_of_reads(bases, off, k, 2)
    assert len(kmers) > 3000
    sub = k - 1
    km = binarize(kmers, k)
    cn = np.asarray(counts, np.int32)
    r = O.rc_expand_subkmer(km, cn, k)
    want = M.rc_expand_str(kmers, counts)
    assert to_model(r, sub, sentinel=False) == want
    # sort + forward fork filter
    r = O.sort_records(r); want = M.stable_sort(want)
    assert to_model(r, sub, sentinel=False) == want
    ps = O.partition_starts(r.key, P)
    assert list(ps) == M.partition_starts([x[0] for x in want], P)
    r, ops = O.fork_filter_forward(r, ps, k, min_err, O.TWIN_DS)
    want, wst = M.fork_forward(want, list(ps), sub, min_err, True)
    assert to_model(r, sub, sentinel=False) == want and list(ops) == wst
    # reflect, sort, reflected fork filter, random reflection
    r = O.reflect_from_forward(r, k); want = M.reflect(want)
    assert to_model(r, sub) == want
    r = O.sort_records(r); want = M.stable_sort(want)
    ps = O.partition_starts(r.key, P)
    r, ops = O.fork_filter_reflected(r, ps, k, min_err, O.TWIN_DS)
    want, wst = M.fork_reflected(want, list(ps), sub, min_err, True)
    assert to_model(r, sub) == want and list(ops) == wst
    marked = sum(1 for x in want if x[3] >= 0 or x[4] >= 0)
    r = O.random_reflection(r, ops, k); want = M.random_reflection(want, wst, sub)
    assert to_model(r, sub) == want
    # extend passes, both start markers
    for it in range(12):
        r = O.sort_records(r); want = M.stable_sort(want)
        ps = O.partition_starts(r.key, P)
        start = 1 if it >= 9 else 2
        r, ops = O.extend_pass(r, ps, k, O.TWIN_DS, start)
        want, wst = M.extend_pass(want, list(ps), sub, False, start)
        assert to_model(r, sub) == want and list(ops) == wst, it
    assert marked > 0 or min_err == 0 or k >= 93, "the planted bubble should mark forks"
    text, nc = O.contigs_text(r, k, 100)
    wtext, wnc = M.contigs_text_w(want, 100)
    assert (text, nc) == (wtext, wnc)


@pytest.mark.parametrize("k,P", [(63, 4), (63, 8), (47, 4), (33, 1), (95, 4)])
def test_driver_matches_string_model(k, P):
    bases, off = planted_reads()
    kmers, counts = kmers_of_reads(bases, off, k, 2)
    prm = O.default_params(k=k, min_cov=2, partitions=P, min_contig=100, extras=0)
    text, nc, trace, rec = O.assemble_from_counts(binarize(kmers, k), np.asarray(counts, np.int32), prm)
    wtrace = []
    want = M.assemble_w(kmers, counts, k, P, prm.min_error_cov, prm.min_iter, prm.max_iter, trace=wtrace)
    assert trace == wtrace
    assert to_model(rec, k - 1) == want
    assert (text, nc) == M.contigs_text_w(want, 100)
    assert nc >= 2 and len(trace) >= prm.min_iter + 3
    # every k-mer of every contig was in the filtered set (SURVEY.md B.7 invariant)
    kept = set(kmers)
    for line in [b for b in text.split(">") if b]:
        s = "".join(line.split("\n")[1:])
        for p in range(len(s) - k + 1):
            f = s[p:p + k]
            rc = "".join(COMP[c] for c in reversed(f))
            assert (f if f <= rc else rc) in kept


def test_scramble_gives_a_second_chance():
    """the first repeat of the record count does not stop the k > 31 loop (:639-645)"""
    k, P = 63, 4
    bases, off = planted_reads()
    kmers, counts = kmers_of_reads(bases, off, k, 2)
    prm = O.default_params(k=k, min_cov=2, partitions=P, min_contig=100, extras=0)
    _, _, trace, _ = O.assemble_from_counts(binarize(kmers, k), np.asarray(counts, np.int32), prm)
    # checks happen at iterations 18, 21, ... = trace indices 13, 16, ... (trace[i] = count after pass i)
    checks = [trace[i] for i in range(prm.min_iter + 3 - 5, len(trace), 3)]
    repeats = [i for i in range(1, len(checks)) if checks[i] == checks[i - 1]]
    assert repeats, "the count must repeat at least once before the loop stops"
    assert len(trace) > (prm.min_iter + 3 - 5) + 3 * repeats[0]


# ---- the from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712)

def state_before_extras(k, P, n_pass=13):
    """records as the driver holds them when iteration minimumIteration + 3 starts (13 passes with the defaults)"""
    bases, off = planted_reads()
    kmers, counts = kmers_of_reads(bases, off, k, 2)
    r = O.rc_expand_subkmer(binarize(kmers, k), np.asarray(counts, np.int32), k)
    r = O.sort_records(r)
    r, ops = O.fork_filter_forward(r, O.partition_starts(r.key, P), k, 8, O.TWIN_DS)
    r = O.sort_records(O.reflect_from_forward(r, k))
    r, ops = O.fork_filter_reflected(r, O.partition_starts(r.key, P), k, 8, O.TWIN_DS)
    r = O.random_reflection(r, ops, k)
    for _ in range(n_pass):
        r = O.sort_records(r)
        r, _ = O.extend_pass(r, O.partition_starts(r.key, P), k, O.TWIN_DS)
    return r


@pytest.mark.parametrize("k,P,n_pass", [(63, 4, 13), (47, 3, 6), (33, 8, 9), (95, 2, 4)])
def test_extras_operators_match_string_model(k, P, n_pass):
    sub = k - 1
    r = O.sort_records(state_before_extras(k, P, n_pass))
    want = M.stable_sort(to_model(r, sub))
    assert to_model(r, sub) == want
    d = O.double_records(r, k)
    wd = M.double_w(want, sub)
    assert to_model(d, sub) == wd and d.n == 2 * r.n
    d = O.sort_records(d); wd = M.stable_sort(wd)
    ps = O.partition_starts(d.key, P)
    outs = {}
    for op, name in ((O.OP_EXTENDABLE_PAIRS, "pairs"), (O.OP_UNEXTENDABLE, "unext")):
        g, ops = O.key_filter(op, d, ps, k)
        w = M.key_filter_w(name, wd, list(ps), sub)
        assert to_model(g, sub) == w, name
        outs[name] = (O.sort_records(g), M.stable_sort(w))
    assert outs["pairs"][0].n > 0 and outs["unext"][0].n > 0
    for name in ("pairs", "unext"):
        g, w = outs[name]
        ps = O.partition_starts(g.key, P)
        f, _ = O.key_filter(O.OP_FIRST_OF_KEY, g, ps, k)
        wf = M.key_filter_w("first", w, list(ps), sub)
        assert to_model(f, sub) == wf
        outs[name] = (f, wf)
    u = O.Records(np.concatenate([outs["pairs"][0].key, outs["unext"][0].key]),
                  np.concatenate([outs["pairs"][0].marker, outs["unext"][0].marker]),
                  np.concatenate([outs["pairs"][0].ext_off[:-1], outs["unext"][0].ext_off + outs["pairs"][0].ext_off[-1]]),
                  np.concatenate([outs["pairs"][0].ext, outs["unext"][0].ext]),
                  np.concatenate([outs["pairs"][0].left, outs["unext"][0].left]),
                  np.concatenate([outs["pairs"][0].right, outs["unext"][0].right]))
    wu = outs["pairs"][1] + outs["unext"][1]
    for m in (1, 2):
        u = O.flip_all(u, k, m); wu = M.flip_all_w(wu, m, sub)
        assert to_model(u, sub) == wu
        u = O.sort_records(u); wu = M.stable_sort(wu)
        ps = O.partition_starts(u.key, P)
        u, _ = O.key_filter(O.OP_LONGER_OF_KEY, u, ps, k)
        wu = M.key_filter_w("longer", wu, list(ps), sub)
        assert to_model(u, sub) == wu


@pytest.mark.parametrize("k,P", [(63, 4), (63, 8), (47, 4), (33, 1), (95, 4)])
def test_driver_with_extras_matches_string_model(k, P):
    bases, off = planted_reads()
    kmers, counts = kmers_of_reads(bases, off, k, 2)
    prm = O.default_params(k=k, min_cov=2, partitions=P, min_contig=100)
    assert prm.extras == 1                       # the reference's from-counts driver always runs them
    text, nc, trace, rec = O.assemble_from_counts(binarize(kmers, k), np.asarray(counts, np.int32), prm)
    wtrace = []
    want = M.assemble_w(kmers, counts, k, P, prm.min_error_cov, prm.min_iter, prm.max_iter, trace=wtrace, extras=True)
    assert trace == wtrace
    assert to_model(rec, k - 1) == want
    assert (text, nc) == M.contigs_text_w(want, 100)
    assert nc >= 1
