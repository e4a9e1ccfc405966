"""The sequence-level oracle (oracle/reflexiv_oracle.c: unpack, concatenate, repack) against the reference's own
bit arithmetic written out statement by statement (oracle/literal/reflexiv_literal.c, from
P/ReflexivDSMain.java:3153-3325 and :2702-2967): flips and merges of the single-word and first-array extend stages,
both output orientations, free ends and bubble distances, fuzzed.  A mechanical pin of the sequence model to the
reference's code."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LITDIR = os.path.join(os.path.dirname(HERE), "oracle")


class LitRec(C.Structure):
    _fields_ = [("key", C.c_int64), ("marker", C.c_int32), ("ext", C.c_int64 * 2), ("n_ext", C.c_int32),
                ("left", C.c_int32), ("right", C.c_int32)]


@pytest.fixture(scope="module")
def lit():
    subprocess.check_call(["make", "-s", "-C", LITDIR, "literal/libliteral.so"])
    return C.CDLL(os.path.join(LITDIR, "literal", "libliteral.so"))


def pack(bases):
    x = 0
    for b in bases:
        x = (x << 2) | int(b)
    return x


def sentinel(bases):
    return (1 << (2 * len(bases))) | pack(bases)


def as_records(recs):
    """[(key, marker, ext_word, left, right)] single-word -> oracle Records"""
    return O.Records.from_single(np.array([r[0] for r in recs], np.uint64), np.array([r[1] for r in recs], np.int32),
                                 np.array([r[2] for r in recs], np.uint64), np.array([r[3] for r in recs], np.int32),
                                 np.array([r[4] for r in recs], np.int32))


def oracle_pass(recs, k, m):
    out, _ = O.extend_pass(as_records(recs), np.array([0, len(recs)], np.int64), k, O.TWIN_DS, m)
    return out


def same(out, lr, n_words):
    assert out.n == 1
    assert int(out.key[0]) == lr.key & 0xFFFFFFFFFFFFFFFF
    assert int(out.marker[0]) == lr.marker
    assert int(out.ext_off[1]) == n_words == lr.n_ext
    for w in range(n_words):
        assert int(out.ext[w]) == lr.ext[w] & 0xFFFFFFFFFFFFFFFF, w
    assert (int(out.left[0]), int(out.right[0])) == (lr.left, lr.right)


@pytest.mark.parametrize("k", [31, 21, 11])
def test_flips_equal_the_reference_bit_code(lit, k):
    rng = np.random.default_rng(k)
    sub = k - 1
    for _ in range(20000):
        L = int(rng.integers(1, min(31, sub)))                 # the reference refuses L >= subKmerSize
        key = pack(rng.integers(0, 4, sub)); ext = sentinel(rng.integers(0, 4, L))
        marker = int(rng.integers(1, 3)); m = int(rng.integers(1, 3))
        left, right = int(rng.integers(-50, 40)), int(rng.integers(-50, 40))
        cur = LitRec(key, marker, (C.c_int64 * 2)(ext, 0), 1, left, right)
        for fn in (lit.lit_flip_single, lit.lit_flip_first):
            got = LitRec()
            assert fn(C.byref(cur), m, sub, C.byref(got)) == 0
            same(oracle_pass([(key, marker, ext, left, right)], k, m), got, 1)


def bubble_case(rng, lenF, lenR):
    """(left of F, right of R, bubbleDistance as call() computes it for input order [F, R]) -- never BLOCK"""
    c = int(rng.integers(0, 3))
    if c == 0:
        return int(rng.integers(-60, 0)), int(rng.integers(-60, 0)), -1
    if c == 1:                                                     # s = R: right of R first  (P/ReflexivDSMain.java:1851-1857)
        b = lenF + int(rng.integers(0, 30))
        return int(rng.integers(-60, 0)), b, b - lenF
    a = lenR + int(rng.integers(0, 30))
    return a, int(rng.integers(-60, 0)), a - lenR


@pytest.mark.parametrize("k", [31, 21])
def test_single_word_merges_equal_the_reference_bit_code(lit, k):
    rng = np.random.default_rng(100 + k)
    sub = k - 1
    for _ in range(20000):
        lenF, lenR = int(rng.integers(1, 16)), int(rng.integers(1, 16))        # merged <= 30 bases: one word
        if lenF >= sub or lenR >= sub:
            continue
        key = pack(rng.integers(0, 4, sub))
        eF, eR = sentinel(rng.integers(0, 4, lenF)), sentinel(rng.integers(0, 4, lenR))
        a, b, d = bubble_case(rng, lenF, lenR)
        fr, rl = int(rng.integers(-60, 40)), int(rng.integers(-60, 40))
        m = int(rng.integers(1, 3))
        f = LitRec(key, 1, (C.c_int64 * 2)(eF, 0), 1, a, fr)
        r = LitRec(key, 2, (C.c_int64 * 2)(eR, 0), 1, rl, b)
        got = LitRec()
        assert lit.lit_merge_single(C.byref(f), C.byref(r), d, m, sub, C.byref(got)) == 0
        same(oracle_pass([(key, 1, eF, a, fr), (key, 2, eR, rl, b)], k, m), got, 1)


@pytest.mark.parametrize("k", [31, 25])
def test_first_array_merges_equal_the_reference_bit_code(lit, k):
    """merged length up to 2 x 16 = 32 bases in the driver; fuzzed up to 31 + 29 to exercise the two-word branch widely"""
    rng = np.random.default_rng(200 + k)
    sub = k - 1
    two_word = 0
    for _ in range(20000):
        lenF, lenR = int(rng.integers(1, min(31, sub))), int(rng.integers(1, min(31, sub)))
        if lenF + lenR > 61:
            continue
        key = pack(rng.integers(0, 4, sub))
        eF, eR = sentinel(rng.integers(0, 4, lenF)), sentinel(rng.integers(0, 4, lenR))
        a, b, d = bubble_case(rng, lenF, lenR)
        fr, rl = int(rng.integers(-60, 40)), int(rng.integers(-60, 40))
        m = int(rng.integers(1, 3))
        f = LitRec(key, 1, (C.c_int64 * 2)(eF, 0), 1, a, fr)
        r = LitRec(key, 2, (C.c_int64 * 2)(eR, 0), 1, rl, b)
        got = LitRec()
        assert lit.lit_merge_first(C.byref(f), C.byref(r), d, m, sub, C.byref(got)) == 0
        nw = 2 if lenF + lenR > 31 else 1
        two_word += nw == 2
        same(oracle_pass([(key, 1, eF, a, fr), (key, 2, eR, rl, b)], k, m), got, nw)
    assert two_word > 3000


@pytest.mark.parametrize("k", [33, 40, 47, 63, 65, 95, 127])
@pytest.mark.parametrize("clips", [(0, 0), (3, 5)])
def test_counter64_extraction_literal_equals_oracle(lit, k, clips):
    """k > 31 counter: the reference's rolling multi-word forward / reverse-complement arrays and
    compareLongArrayBlocks, written out statement by statement (lit_counter64_extract, from
    P/ReflexivDataFrameCounter64.java:391-687), against the oracle's sequence-level orc_extract_canon_w -- random
    reads with N's and lower case, reads shorter than k, palindromes, both clips."""
    fc, ec = clips
    rng = np.random.default_rng(1000 * k + fc)
    W = k // 32 + 1
    lit.lit_counter64_extract.restype = C.c_int64
    reads = []
    for _ in range(300):
        L = int(rng.integers(k - 3, 3 * k + 40))
        alphabet = np.frombuffer(b"ACGTACGTACGTNacgt", np.uint8)
        reads.append(bytes(alphabet[rng.integers(0, len(alphabet), L)]))
    # a reverse-complement palindrome of length >= k + clips: forward == reverse complement, the forward strand is kept
    half = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, k)])
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads.append(b"A" * fc + half + half.translate(comp)[::-1] + b"C" * ec)
    bases = np.frombuffer(b"".join(reads), np.uint8)
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    want = O.extract_canon_w(bases, off, k, fc, ec)
    got = []
    for r in reads:
        cap = max(1, len(r))
        buf = np.zeros((cap, W), np.uint64)
        n = lit.lit_counter64_extract(r, len(r), k, fc, ec, buf.ctypes.data_as(C.c_void_p), C.c_int64(cap))
        assert 0 <= n <= cap
        got.append(buf[:n])
    got = np.concatenate(got) if got else np.zeros((0, W), np.uint64)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


class LitRow(C.Structure):
    _fields_ = [("key", C.c_int64), ("marker", C.c_int32), ("ext", C.c_int64), ("left", C.c_int32), ("right", C.c_int32)]


@pytest.mark.parametrize("reflected", [False, True])
@pytest.mark.parametrize("k", [31, 21])
def test_fork_filters_literal_equal_oracle(lit, k, reflected):
    """DSFilterForkSubKmer / DSFilterForkReflectedSubKmer (P/ReflexivDSMain.java:3369-3417, :3486-3538) written out
    statement by statement against the oracle's segmented fold, on sorted partitions with runs of 1..5 equal keys,
    tied coverages and tied / differing first extension bases (the reflected filter's comparison of a first base with
    the kept row's SENTINEL included)."""
    rng = np.random.default_rng(7 * k + int(reflected))
    sub = k - 1
    fn = lit.lit_fork_reflected if reflected else lit.lit_fork_forward
    fn.restype = C.c_int64
    for _ in range(300):
        rows = []
        for _g in range(int(rng.integers(1, 12))):
            key = int(rng.integers(0, 1 << 40))
            for _r in range(int(rng.integers(1, 6))):
                L = int(rng.integers(1, 4))
                ext = sentinel(rng.integers(0, 4, L))
                rows.append((key, int(rng.integers(1, 3)), ext, int(rng.integers(1, 4)), int(rng.integers(1, 4))))
        rows.sort(key=lambda r: r[0])                    # (stable: equal keys keep their order, as after sortByKey)
        n = len(rows)
        arr = (LitRow * n)(*[LitRow(*r) for r in rows])
        out = (LitRow * n)()
        m = fn(arr, C.c_int64(n), sub, out)
        recs = as_records(rows)
        f = O.fork_filter_reflected if reflected else O.fork_filter_forward
        got, _ = f(recs, np.array([0, n], np.int64), k, 0, O.TWIN_DS)
        assert got.n == m
        for i in range(m):
            assert (int(got.key[i]), int(got.marker[i]), int(got.ext[i]), int(got.left[i]), int(got.right[i])) == \
                   (out[i].key, out[i].marker, out[i].ext, out[i].left, out[i].right), (i, rows)
