"""Contig RC de-duplication: the oracle (oracle/reflexiv_dedup.c) against vectors made by the REFERENCE'S
OWN classes of P/ReflexivDSDynamicKmerDedup.java (tests/golden/dedup_vectors.npz, written by
tests/golden/make_dedup_vectors.py through tools/java2py.py): contigs on both strands, RC pieces with new flanks, forward
pieces, mutated copies, exact copies, shuffled sets -- the survivors after each of the three rounds and the final text."""
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "dedup_vectors.npz")


def cases():
    z = np.load(VEC)
    return sorted({k.split("/")[0] for k in z.files})


def unpack(z, name):
    b, off = bytes(z[name]), z[name + "_off"]
    return [b[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]


@pytest.mark.parametrize("case", cases())
def test_dedup_rounds_and_text_equal_the_reference_classes(case):
    z = np.load(VEC)
    contigs = unpack(z, case + "/in")
    got = O.dedup_contigs(contigs)
    for r in range(3):
        want = unpack(z, f"{case}/round{r + 1}")
        assert got["rounds"][r] == want, (case, r, [len(x) for x in got["rounds"][r]], [len(x) for x in want])
    assert got["text"] == bytes(z[case + "/text"]).decode()
    if f"{case}/r1_pairs" in z.files:
        for r in range(3):
            assert got["pairs"][r] == len(z[f"{case}/r{r + 1}_pairs"])
            assert got["candidates"][r] == len(z[f"{case}/r{r + 1}_candidates"])


def test_both_strands_of_a_contig_set_collapse_to_one():
    """what the fixed-k path emits (every contig on both strands) comes back once; contigs under 300 bases are left alone"""
    rng = np.random.default_rng(3)
    comp = str.maketrans("ACGT", "TGCA")
    seqs = ["".join("ACGT"[b] for b in rng.integers(0, 4, n)) for n in (5000, 2500, 800, 299)]
    contigs = []
    for s in seqs:
        contigs += [s, s.translate(comp)[::-1]]
    got = O.dedup_contigs(contigs)
    final = got["rounds"][2]
    assert sorted(len(x) for x in final) == [299, 299, 800, 2500, 5000]
    assert got["text"].count(">Contig-") == 3                 # minContig 500: the 299s are not printed
