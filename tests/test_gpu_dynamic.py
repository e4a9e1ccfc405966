"""The dynamic-k record format and passes on the GPU (rfx_dyn_*) against the vectors made by the REFERENCE'S
OWN classes of P/ReflexivDSDynamicKmerFirstFour.java / ...Iteration.java (tests/golden/dynamic_vectors.npz): every operator
fed with the reference's previous output (sort + one pass per call), the resident drivers end to end, and a larger random
set against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_dynamic import VEC, cases, rows_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


def check(rows, want, tag):
    assert len(rows) == len(want), (tag, len(rows), len(want))
    for i, (a, b) in enumerate(zip(rows, want)):
        assert a == b, (tag, i, a, b)


@pytest.mark.parametrize("case", cases())
def test_gpu_dynamic_operators_equal_the_reference_classes(rfx, case):
    from reflexiv_amd.api import DynRecords
    z = np.load(VEC)
    P, start, end = (int(x) for x in z[case + "/meta"])
    r = DynRecords.from_kmer_rows(rows_of(z, case + "/in"))
    check(r.rows(), rows_of(z, case + "/binarized"), "binarized")
    n = r.n
    st = np.array([p * n // P for p in range(P)] + [n], np.int64)
    g = rfx.dyn_random_reflection(r, st)
    check(g.rows(), rows_of(z, case + "/random_reflection"), "random_reflection")
    prev = "random_reflection"
    for it in range(4):
        r = DynRecords.from_rows(rows_of(z, f"{case}/{prev}"))          # the reference's previous output
        s, ps = rfx.dyn_sort(r, P)
        o = O.dyn_sort(O.dyn_binarize_rows(rows_of(z, f"{case}/{prev}")))
        check(s.rows(), o.rows(), f"sort before extend{it}")
        assert np.array_equal(ps, O.dyn_partition_starts(o, P))
        g, _ = rfx.dyn_extend_pass(s, ps, 0)
        check(g.rows(), rows_of(z, f"{case}/extend{it}"), f"extend{it}")
        prev = f"extend{it}"
    prev = "it_binarized"
    it = start
    while it <= end:
        it += 1
        r = DynRecords.from_rows(rows_of(z, f"{case}/{prev}"))
        s, ps = rfx.dyn_sort(r, P)
        g, _ = rfx.dyn_extend_pass(s, ps, 1, start)
        check(g.rows(), rows_of(z, f"{case}/it_extend{it}"), f"it_extend{it}")
        prev = f"it_extend{it}"


@pytest.mark.parametrize("case", cases())
def test_gpu_dynamic_drivers_end_to_end(rfx, case):
    """FirstFour.assemblyFromKmer and Iteration.assemblyFromKmer with the records resident in HBM: rows in, rows out"""
    from reflexiv_amd.api import DynRecords
    z = np.load(VEC)
    P, start, end = (int(x) for x in z[case + "/meta"])
    r = DynRecords.from_kmer_rows(rows_of(z, case + "/in"))
    ff, tr = rfx.dyn_run(r, P, random_reflection=True, passes_first_four=4)
    check(ff.rows(), rows_of(z, case + "/extend3"), "first four")
    assert tr == [len(rows_of(z, f"{case}/extend{i}")) for i in range(4)]
    # (the reference's loop: iterations = start; while (iterations <= end) { iterations++; sort; pass } -- end - start + 1 passes,
    # all under param.startIteration's rules)
    fin, tr = rfx.dyn_run(DynRecords.from_rows(ff.rows()), P, start_iteration=start, end_iteration=end)
    check(fin.rows(), rows_of(z, case + "/final"), "iterations")
    assert len(tr) == end - start + 1


def test_gpu_dynamic_on_a_larger_set_equals_the_oracle(rfx):
    from reflexiv_amd.api import DynRecords
    rng = np.random.default_rng(5)
    comp = str.maketrans("ACGT", "TGCA")
    g = "".join("ACGT"[b] for b in rng.integers(0, 4, 30_000))
    rows, seen = [], set()
    for s in (g, g.translate(comp)[::-1]):
        pos = 0
        while pos < len(s):
            k = int(rng.choice([23, 31, 41, 53, 67, 81, 95]))
            span = int(rng.integers(60, 400))
            for p in range(pos, min(pos + span, len(s) - k + 1)):
                km = s[p:p + k]
                if km not in seen:
                    seen.add(km)
                    l = int(rng.integers(0, 80)) if rng.random() < 0.2 else -int(rng.integers(2, 60))
                    r = int(rng.integers(0, 80)) if rng.random() < 0.2 else -int(rng.integers(2, 60))
                    rows.append((km, f"1|{l}|{r}"))
            pos += span
    rows = [rows[i] for i in rng.permutation(len(rows))]
    P = 4
    want_ff, _ = O.dyn_first_four(rows, P)
    ff, _ = rfx.dyn_run(DynRecords.from_kmer_rows(rows), P, random_reflection=True, passes_first_four=4)
    check(ff.rows(), want_ff, "first four")
    want, _ = O.dyn_iterations(want_ff, P, 5, 14)
    fin, _ = rfx.dyn_run(DynRecords.from_rows(want_ff), P, start_iteration=5, end_iteration=14)
    check(fin.rows(), want, "iterations 5..14")
    assert len(want) < len(want_ff) // 4


def test_cpp_host_firstfour_and_iteration(tmp_path):
    """`reflexiv_host firstfour` / `iteration`: CSV rows in, CSV rows out (the wire format between the reference's jobs)"""
    import subprocess
    import reflexiv_amd._lib as L
    import os
    z = np.load(VEC)
    case = "c1"
    P, start, end = (int(x) for x in z[case + "/meta"])
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    src = tmp_path / "reduced.csv"
    src.write_bytes(bytes(z[case + "/in"]))
    out = str(tmp_path / "out")
    subprocess.check_call([host, "firstfour", "-kmerc", str(src), "-outfile", out, "--logical-partitions", str(P)])
    ff = os.path.join(out, "Assembly_intermediate", "00firstFour", "part-00000.csv")
    assert open(ff).read() == bytes(z[case + "/extend3"]).decode()
    subprocess.check_call([host, "iteration", "-kmerc", ff, "-outfile", out, "--logical-partitions", str(P), "-start", str(start), "-end", str(end)])
    it = os.path.join(out, "Assembly_intermediate", f"01Iteration{start}_{end}", "part-00000.csv")
    assert open(it).read() == bytes(z[case + "/final"]).decode()


@pytest.mark.parametrize("case", cases())
def test_gpu_binarizer_equals_the_reference_class(rfx, case):
    """DynamicKmerBinarizerFromReducedToSubKmer ON THE DEVICE (rfx_dyn_binarize) -- the first operator of the chain -- against the
    reference class's own output: FirstFour's (k-mer, attribute) rows, and every Iteration-form row set of the case
    ((sub-k-mer, attribute, extension): the reference's outputs read back in)."""
    z = np.load(VEC)
    r = rfx.dyn_binarize(rows_of(z, case + "/in"))
    check(r.rows(), rows_of(z, case + "/binarized"), "binarized")
    for name in sorted(n for n in z.files if n.startswith(case + "/") and (n.split("/")[1].startswith("extend") or n.split("/")[1].startswith("it_"))):
        rows = rows_of(z, name)
        g = rfx.dyn_binarize(rows, form=1)
        check(g.rows(), rows, name)                               # text -> records -> text is the identity on the reference's rows


def test_gpu_binarizer_on_the_edges(rfx):
    """tuple text with parentheses, negative and out-of-range attribute values (read back clamped to +-30000), bases outside
    ACG, empty extensions, CRLF line ends, an empty row set: the device parser against the harness's Python one."""
    from reflexiv_amd.api import DynRecords
    rng = np.random.default_rng(11)
    rows0, rows1 = [], []
    for i in range(5000):
        L = int(rng.integers(2, 125))
        kmer = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        a = f"{int(rng.integers(1, 3))}|{int(rng.integers(-40000, 40000))}|{int(rng.integers(-40000, 40000))}"
        rows0.append((("(" if i % 3 == 0 else "") + kmer, a + (")" if i % 3 == 0 else "")))
        ext = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 200))))
        rows1.append((("(" if i % 5 == 0 else "") + kmer, a, ext + (")" if i % 5 == 0 else "")))
    for rows, ref in ((rows0, DynRecords.from_kmer_rows(rows0)),
                      (rows1, DynRecords.from_rows([(k, a, e[:-1] if e.endswith(")") else e) for k, a, e in rows1]))):
        g = rfx.dyn_binarize(rows)
        assert g.n == ref.n
        for f in ("key_off", "ext_off", "marker", "left", "right"):
            assert np.array_equal(getattr(g, f), getattr(ref, f)), f
        assert np.array_equal(g.key[:g.key_off[-1]], ref.key[:ref.key_off[-1]]) and np.array_equal(g.ext[:g.ext_off[-1]], ref.ext[:ref.ext_off[-1]])
    assert rfx.dyn_binarize([]).n == 0
    g = rfx.dyn_binarize([("ACGTA", "1|-1|7\r\n")])
    assert g.rows() == [("ACGT", "1|-1|7", "A")]
