"""k > 31: the oracle's restatement of ReflexivDataFrameCounter64 (rolling W-word arithmetic)
against an independent string-level model, and against the committed fixture."""
import collections
import os

import numpy as np
import pytest

from oracle import oracle as O

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def model_extract(reads, k, fc=0, ec=0):
    """canonical k-mers as strings: A C G, anything else T; skip rule of :410; ties -> forward."""
    out = []
    for r in reads:
        L = len(r)
        if L - k - ec + 1 <= 0 or fc > L:
            continue
        s = "".join(c if c in "ACG" else "T" for c in r)
        body = s[fc:L - ec]
        for p in range(len(body) - k + 1):
            f = body[p:p + k]
            rc = "".join(COMP[c] for c in reversed(f))
            out.append(f if f <= rc else rc)
    return out


def random_reads(rng, n, lo, hi):
    reads = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(lo, hi)), p=[.24, .24, .24, .24, .04]))
             for _ in range(n)]
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    return reads, bases, off


@pytest.mark.parametrize("k", [33, 47, 63, 65, 95, 127])
@pytest.mark.parametrize("clips", [(0, 0), (3, 5)])
def test_extract_and_count_match_string_model(k, clips):
    rng = np.random.default_rng(k)
    reads, bases, off = random_reads(rng, 40, 20, 220)
    fc, ec = clips
    got = O.extract_canon_w(bases, off, k, fc, ec)
    want = model_extract(reads, k, fc, ec)
    assert got.shape == (len(want), k // 32 + 1)
    assert [O.kmer_text_w(g, k) for g in got] == want
    # last word holds k % 32 bases right-aligned, nothing above them
    assert int(got[:, -1].max(initial=0)) < 1 << (2 * (k % 32))
    for min_cov, max_cov in ((1, 10_000_000), (2, 10_000_000), (1, 2)):
        keys, counts, nd = O.count_filter_w(got, k, min_cov, max_cov)
        cnt = collections.Counter(want)
        keep = sorted(s for s, c in cnt.items() if (min_cov <= 1 or c >= min_cov) and c <= max_cov)
        assert nd == len(cnt)
        assert [O.kmer_text_w(x, k) for x in keys] == keep
        assert [int(c) for c in counts] == [cnt[s] for s in keep]


def test_rejects_unsupported_k():
    bases = np.frombuffer(b"ACGT" * 40, np.uint8)
    off = np.array([0, 160], np.int64)
    for k in (31, 32, 64):
        with pytest.raises(ValueError):
            O.extract_canon_w(bases, off, k)


def test_palindrome_and_short_reads():
    k = 33
    # an odd-length k-mer cannot equal its reverse complement; use reads shorter than k and exactly k
    reads = ["ACGT" * 8, "A" * 33, "ACGTACGTAC" * 4]
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    got = O.extract_canon_w(bases, off, k)
    assert [O.kmer_text_w(g, k) for g in got] == model_extract(reads, k)
    assert len(got) == 1 + 8          # 32-base read: none; 33: one; 40: eight


def test_golden_wide_fixture(golden_dir):
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    gw = np.load(os.path.join(golden_dir, "wide.npz"))
    bases, off = ex["bases"], ex["read_off"]
    for k in (63, 47):
        km = O.extract_canon_w(bases, off, k)
        assert len(km) == int(gw[f"k{k}_n_instances"])
        assert np.array_equal(km[:len(gw[f"k{k}_first4"])], gw[f"k{k}_first4"])
        keys, counts, nd = O.count_filter_w(km, k, 3)
        assert nd == int(gw[f"k{k}_n_distinct"])
        assert np.array_equal(keys, gw[f"k{k}_keys_cov3"]) and np.array_equal(counts, gw[f"k{k}_counts_cov3"])
        assert O.kmer_text_w(keys[0], k) == str(gw[f"k{k}_text_first"])
