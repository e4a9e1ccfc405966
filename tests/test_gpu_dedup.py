"""Contig RC de-duplication on the GPU (rfx_dedup_contigs / rfx_dedup_contig_text) against the vectors made
by the REFERENCE'S OWN classes of P/ReflexivDSDynamicKmerDedup.java (tests/golden/dedup_vectors.npz) and against the oracle
on larger random sets; the documented example (2 x 4558 -> 1) and the planted fixture through the path's own contig text."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_dedup import VEC, cases, unpack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


@pytest.mark.parametrize("case", cases())
def test_gpu_dedup_equals_the_reference_classes(rfx, case):
    z = np.load(VEC)
    contigs = unpack(z, case + "/in")
    surv, text, rn = rfx.dedup_contigs(contigs)
    want = unpack(z, f"{case}/round3")
    assert surv == want, (case, [len(x) for x in surv], [len(x) for x in want])
    assert text == bytes(z[case + "/text"]).decode()
    assert rn == [len(unpack(z, f"{case}/round{r}")) for r in (1, 2, 3)]


COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def rand_seq(rng, n):
    return "".join("ACGT"[b] for b in rng.integers(0, 4, n))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_dedup_equals_the_oracle_on_larger_sets(rfx, seed):
    """both strands, RC pieces with new flanks, forward pieces, near-copies, repeats shared between contigs, long contigs
    (>= 100 kbp: tables and distance lists of real size) -- every round's survivor count and the final text"""
    rng = np.random.default_rng(seed)
    base = []
    rep = rand_seq(rng, 400)
    for L in list(rng.integers(300, 9000, 30)) + [120_000, 65_000, 31 * 200, 31 * 97 + 30]:
        s = rand_seq(rng, int(L))
        if rng.random() < 0.3 and L > 1500:
            p = int(rng.integers(100, L - 500))
            s = s[:p] + rep + s[p + 400:]                        # a repeat shared between contigs
        base.append(s)
        r = rng.random()
        if r < 0.55:
            base.append(rc(s))
        elif r < 0.7:
            a, b = int(rng.integers(0, 200)), int(rng.integers(0, 200))
            base.append(rc(rand_seq(rng, a) + s[int(L) // 5: int(L) * 4 // 5] + rand_seq(rng, b)))
        elif r < 0.8:
            base.append(s[int(L) // 10: int(L) // 2])
        elif r < 0.9:
            t = list(rc(s))
            for p in rng.integers(0, len(t), max(1, len(t) // 300)):
                t[p] = "ACGT"[("ACGT".index(t[p]) + 1) % 4]
            base.append("".join(t))
    base += [rand_seq(rng, int(L)) for L in rng.integers(50, 400, 8)]
    order = rng.permutation(len(base))
    contigs = [base[i] for i in order]
    want = O.dedup_contigs(contigs)
    surv, text, rn = rfx.dedup_contigs(contigs)
    assert rn == [len(r) for r in want["rounds"]]
    assert surv == want["rounds"][2]
    assert text == want["text"]


def test_gpu_dedup_of_the_documented_example_and_the_planted_fixture(rfx, golden_dir):
    """the path's own output through rfx_dedup_contig_text: the documented example's two strands of the 4558-base contig
    come back as one (2 x 4558 -> 1); the planted fixture equals the oracle"""
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    for twin in ("rdd", "ds"):
        text = str(ex[f"contigs_{twin}_P4"])
        assert text.count(">Contig-4558-") == 2
        out, nc, rn = rfx.dedup_contig_text(text)
        assert nc == 1 and out.startswith(">Contig-4558-0\n") and out.count(">") == 1
        seq = out.split("\n", 1)[1].replace("\n", "")
        seqs = ["".join(part.split("\n")[1:]) for part in text.split(">")[1:]]
        assert seq in seqs                                          # one of the two strands, untouched
    planted = np.load(os.path.join(golden_dir, "planted.npz"))
    text = str(planted["k31_ds_contigs"])
    contigs = ["".join(part.split("\n")[1:]) for part in text.split(">")[1:]]
    want = O.dedup_contigs(contigs, 100)
    out, nc, rn = rfx.dedup_contig_text(text, 100)
    assert out == want["text"] and nc == len(want["rounds"][2])
    assert rfx.dedup_contigs([], 100) == ([], "", [0, 0, 0])


def test_cpp_host_run_with_dedup(tmp_path, golden_dir):
    """`reflexiv_host run --resident --dedup`: the documented example comes out as ONE contig of 4558 bases"""
    import gzip
    import subprocess
    import reflexiv_amd._lib as L
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    fq = str(tmp_path / "ex.fq.gz")
    with gzip.open(fq, "wb") as f:
        b, off = bytes(ex["bases"]), ex["read_off"]
        for i in range(len(off) - 1):
            s = b[off[i]:off[i + 1]]
            f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
    out = str(tmp_path / "result")
    subprocess.check_call([host, "run", "--resident", "--dedup", "-fastq", fq, "-outfile", out, "-kmer", "31", "-cover", "3",
                           "--logical-partitions", "4", "--twin", "rdd"])
    text = open(os.path.join(out, "part-00000")).read()
    assert text.count(">") == 1 and text.startswith(">Contig-4558-0\n")
