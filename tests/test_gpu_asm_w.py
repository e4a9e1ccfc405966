"""GPU parity, k > 31 assembler (P/ReflexivDSMain64.java assemblyFromKmer): every operator on multi-word
(k-1)-mer keys through the C ABI against the CPU oracle, the device driver against the oracle's driver,
and the resident path reads -> counter -> KmerBinarizer -> contigs.  Bit-exact (integer work)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def planted(golden_dir):
    return np.load(os.path.join(golden_dir, "planted.npz"))


def same_records(a, b):
    assert a.n == b.n
    for f in ("key", "marker", "ext_off", "ext", "left", "right"):
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape and np.array_equal(x, y), f


def filtered_kmers(bases, off, k, min_cov):
    """(k-mers in the assembler layout uint64[n, (k-1)//31+1], int32 counts) ascending, from the oracle's counter"""
    if k % 32 == 0 or k <= 32:
        # the k > 31 counter does not take these k: go through text
        import collections
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        text = bytes(bases).decode()
        cnt = collections.Counter()
        for r in range(len(off) - 1):
            s = "".join(c if c in "ACG" else "T" for c in text[off[r]:off[r + 1]])
            for p in range(len(s) - k + 1):
                f = s[p:p + k]
                rc = "".join(comp[c] for c in reversed(f))
                cnt[f if f <= rc else rc] += 1
        keep = sorted(x for x, c in cnt.items() if c >= min_cov)
        km = np.stack([O.kmer_binarize_w(x, "1", k)[0] for x in keep])
        return km, np.asarray([cnt[x] for x in keep], np.int32)
    keys, counts, _ = O.count_filter_w(O.extract_canon_w(bases, off, k), k, min_cov)
    return O.counter_to_asm_w(keys, k), counts.astype(np.int32)


def run_chain_w(rfx, km, counts, k, P, min_err, n_pass=12):
    """GPU and oracle side by side, every operator fed with the ORACLE's previous output."""
    twin = O.TWIN_DS
    o = O.rc_expand_subkmer(km, counts, k)
    g = rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(km, counts, k)
    same_records(g, o)
    o = O.sort_records(o)
    ops = O.partition_starts(o.key, P)
    g, gps = rfx.sortByKey(g, P)
    same_records(g, o)
    assert np.array_equal(gps, ops)
    o2, ops2 = O.fork_filter_forward(o, ops, k, min_err, twin)
    g2, gps2 = rfx.FilterForkSubKmer(o, ops, k, min_err, twin)
    same_records(g2, o2)
    assert np.array_equal(gps2, ops2)
    o3 = O.reflect_from_forward(o2, k)
    same_records(rfx.ReflectedSubKmerExtractionFromForward(o2, k), o3)
    o3 = O.sort_records(o3)
    ops3 = O.partition_starts(o3.key, P)
    o4, ops4 = O.fork_filter_reflected(o3, ops3, k, min_err, twin)
    g4, gps4 = rfx.FilterForkReflectedSubKmer(o3, ops3, k, min_err, twin)
    same_records(g4, o4)
    assert np.array_equal(gps4, ops4)
    o5 = O.random_reflection(o4, ops4, k)
    same_records(rfx.kmerRandomReflection(o4, ops4, k), o5)
    cur = o5
    for i in range(n_pass):
        cur = O.sort_records(cur)
        ps = O.partition_starts(cur.key, P)
        gs, gps = rfx.sortByKey(cur, P)
        same_records(gs, cur)
        assert np.array_equal(gps, ps)
        scramble = 3 if i >= n_pass - 3 else 2
        nxt, nps = O.extend_pass(cur, ps, k, twin, 1 if scramble == 3 else 2)
        stage = 0 if i < 4 else (1 if i == 4 else 2)
        g, gps = rfx.ExtendReflexivKmer(cur, ps, k, twin, stage, scramble)
        same_records(g, nxt)
        assert np.array_equal(gps, nps)
        cur = nxt
    text, nc = rfx.KmerToContig(cur, k, 100)
    assert (text, nc) == O.contigs_text(cur, k, 100)
    return cur


@pytest.mark.parametrize("k,P,min_err", [(63, 4, 8), (63, 1, 0), (47, 3, 8), (33, 4, 8), (32, 2, 8), (62, 4, 8),
                                         (64, 4, 8), (95, 2, 8)])
def test_operator_chain_w(rfx, planted, k, P, min_err):
    km, counts = filtered_kmers(planted["bases"], planted["read_off"], k, 2)
    assert len(km) > 1000
    run_chain_w(rfx, km, counts, k, P, min_err)


@pytest.mark.parametrize("k", [63, 40, 33])
@pytest.mark.parametrize("run", [3, 40])
def test_sort_two_word_keys_with_equal_prefixes(rfx, k, run):
    """sortByKey on two-word keys is ONE radix sort on the first 64 bits plus a mending of the runs of equal prefixes
    (k_tie_fix2).  Groups of `run` keys that share those 64 bits and differ only beyond them (or not at all), in random
    order: short runs are mended in place, runs of 40 (> TIE_MAX) send the call down the two-pass form -- either way
    the oracle's stable order."""
    rng = np.random.default_rng(100 * k + run)
    sub = k - 1
    res = sub - 31
    n_groups = 600
    w0 = rng.integers(0, 1 << 62, n_groups, dtype=np.uint64)
    keys = []
    for g in range(n_groups):
        top = int(rng.integers(0, 4)) << (2 * res - 2) if res >= 1 else 0
        for _ in range(run):
            low = int(rng.integers(0, 1 << (2 * res - 2))) if res > 1 else 0
            if rng.random() < 0.3 and keys and keys[-1][0] == int(w0[g]):
                low = keys[-1][1] & ((1 << (2 * res - 2)) - 1) if res > 1 else 0      # a fully equal key: order must be stable
            keys.append((int(w0[g]), top | low))
    order = rng.permutation(len(keys))
    key = np.array([keys[i] for i in order], np.uint64).reshape(-1, 2)
    n = len(key)
    marker = rng.integers(1, 3, n).astype(np.int32)
    ext = np.array([(1 << 2) | int(rng.integers(0, 4)) for _ in range(n)], np.uint64)
    left = np.arange(n, dtype=np.int32)                    # the input position: shows the stability
    right = rng.integers(-1, 5, n).astype(np.int32)
    r = O.Records(key, marker, np.arange(n + 1, dtype=np.int64), ext, left, right)
    want = O.sort_records(r)
    got, gps = rfx.sortByKey(r, 3)
    same_records(got, want)
    assert np.array_equal(gps, O.partition_starts(want.key, 3))


def test_operator_edge_cases_w(rfx):
    k = 63
    empty = O.Records(np.zeros((0, 2), np.uint64), np.zeros(0, np.int32), np.zeros(1, np.int64),
                      np.zeros(0, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.int32))
    ps = np.zeros(3, np.int64)
    g, gps = rfx.ExtendReflexivKmer(empty, ps, k)
    assert g.n == 0 and list(gps) == [0, 0, 0]
    g = rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(np.zeros((0, 3), np.uint64), np.zeros(0, np.int32), k)
    assert g.n == 0
    # one k-mer: two records (itself and its reverse complement), one pass each way
    s = "ACGT" * 15 + "ACG"
    km = O.kmer_binarize_w(s, "5", k)[0].reshape(1, 3)
    o = O.rc_expand_subkmer(km, np.array([5], np.int32), k)
    same_records(rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(km, np.array([5], np.int32), k), o)
    # a key-width that does not belong to k is refused, not misread
    import reflexiv_amd
    with pytest.raises(reflexiv_amd.RfxError):
        rfx.ReflectedSubKmerExtractionFromForward(o, 31)


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else a.dtype)).cuda()


@pytest.mark.parametrize("k,P", [(63, 4), (63, 8), (47, 4), (33, 1), (95, 4), (64, 2)])
def test_driver_w_matches_oracle(rfx, torch_mod, planted, k, P):
    import reflexiv_amd
    torch = torch_mod
    km, counts = filtered_kmers(planted["bases"], planted["read_off"], k, 2)
    dk, dc = to_dev(torch, km), to_dev(torch, counts)
    torch.cuda.synchronize()
    text, nc, trace = rfx.assemble_w_dev(dk.data_ptr(), dc.data_ptr(), len(counts),
                                         reflexiv_amd.default_params(k=k, min_cov=2, partitions=P, min_contig=100))
    otext, onc, otrace, _ = O.assemble_from_counts(km, counts, O.default_params(k=k, min_cov=2, partitions=P, min_contig=100))
    assert trace == otrace and nc == onc and text == otext
    assert nc >= 2


@pytest.mark.parametrize("k", [63, 47, 33, 95])
def test_counter_to_asm_dev(rfx, torch_mod, planted, k):
    torch = torch_mod
    keys, counts, _ = O.count_filter_w(O.extract_canon_w(planted["bases"], planted["read_off"], k), k, 1)
    counts = counts.copy()
    counts[::97] = 12_345_678_901                         # >= 10 digits reads as 1000000000 (:10801-10806)
    for lo, hi in ((1, 10_000_000), (3, 40), (2, 1_000_000_000)):
        c32 = np.where(counts >= 1_000_000_000, 1_000_000_000, counts).astype(np.int32)
        keep = (c32 >= lo) & (c32 <= hi)
        want_k = O.counter_to_asm_w(keys[keep], k)
        dk, dc = to_dev(torch, keys), to_dev(torch, counts)
        ok = torch.empty(len(keys) * O.asm_words(k), dtype=torch.int64, device="cuda")
        oc = torch.empty(len(keys), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m = rfx.counter_to_asm_dev(dk.data_ptr(), dc.data_ptr(), len(keys), k, ok.data_ptr(), oc.data_ptr(), lo, hi)
        assert m == int(keep.sum())
        assert np.array_equal(ok[:m * O.asm_words(k)].cpu().numpy().view(np.uint64).reshape(m, -1), want_k)
        assert np.array_equal(oc[:m].cpu().numpy(), c32[keep])


def make_reads_dev(rfx, torch, seed, genome_len, n_reads, read_len):
    wpr = (read_len + 31) // 32
    dg = torch.empty((genome_len + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(seed, genome_len, dg.data_ptr())
    rfx.synth_reads_dev(seed, dg.data_ptr(), genome_len, 0, n_reads, read_len, wpr, dw.data_ptr())
    rfx.sync()
    return dg, dw, wpr


@pytest.mark.parametrize("k,G,n_reads,P", [(63, 200_000, 60_000, 8), (47, 60_000, 16_000, 4)])
def test_resident_reads_to_contigs_w(rfx, torch_mod, k, G, n_reads, P):
    """reads in HBM -> k > 31 counter -> KmerBinarizer + filter -> driver -> text, against the oracle end to end;
    contigs of tens of kbp: the long-extension emit path on two-word keys."""
    import reflexiv_amd
    torch = torch_mod
    seed, L, cov = 5, 150, 3
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    W = k // 32 + 1
    N = rfx.kmers_per_read_w(L, k) * n_reads
    cap = N // 2
    dk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, cov)
    ak = torch.empty(m * O.asm_words(k), dtype=torch.int64, device="cuda"); ac = torch.empty(m, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m2 = rfx.counter_to_asm_dev(dk.data_ptr(), dc.data_ptr(), m, k, ak.data_ptr(), ac.data_ptr(), cov)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter_w(O.extract_canon_w(bases, off, k), k, cov)
    assert (m, nd, m2) == (len(wk), wd, len(wk))
    km = O.counter_to_asm_w(wk, k)
    assert np.array_equal(ak.cpu().numpy().view(np.uint64).reshape(m, -1), km)
    prm = reflexiv_amd.default_params(k=k, min_cov=cov, partitions=P)
    text, nc, trace = rfx.assemble_w_dev(ak.data_ptr(), ac.data_ptr(), m2, prm)
    otext, onc, otrace, _ = O.assemble_from_counts(km, wc.astype(np.int32), O.default_params(k=k, min_cov=cov, partitions=P))
    assert trace == otrace and nc == onc and text == otext
    lens = [int(h.split("-")[1]) for h in text.split("\n") if h.startswith(">")]
    assert max(lens) > 20_000


# ---- the from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712)

def _before_extras(k, P, n_pass):
    from tests.test_oracle_asm_w import state_before_extras
    return state_before_extras(k, P, n_pass)


@pytest.mark.parametrize("k,P,n_pass", [(63, 4, 13), (47, 3, 6), (33, 8, 9), (95, 2, 4)])
def test_extras_operators_match_oracle(rfx, k, P, n_pass):
    """every operator class of the extras through the C ABI against the oracle, each fed with the oracle's previous output"""
    r = O.sort_records(_before_extras(k, P, n_pass))
    ps = O.partition_starts(r.key, P)
    d = O.double_records(r, k)
    g, gps = rfx.extras_operator(0, r, ps, k)
    same_records(g, d)
    assert np.array_equal(gps, 2 * ps)
    d = O.sort_records(d)
    ps = O.partition_starts(d.key, P)
    outs = []
    for op in (O.OP_EXTENDABLE_PAIRS, O.OP_UNEXTENDABLE):
        want, wps = O.key_filter(op, d, ps, k)
        g, gps = rfx.extras_operator(op, d, ps, k)
        same_records(g, want)
        assert np.array_equal(gps, wps)
        s = O.sort_records(want)
        sps = O.partition_starts(s.key, P)
        want2, wps2 = O.key_filter(O.OP_FIRST_OF_KEY, s, sps, k)
        g2, gps2 = rfx.extras_operator(O.OP_FIRST_OF_KEY, s, sps, k)
        same_records(g2, want2)
        assert np.array_equal(gps2, wps2)
        outs.append(want2)
    a, b = outs
    u = O.Records(np.concatenate([a.key, b.key]), np.concatenate([a.marker, b.marker]),
                  np.concatenate([a.ext_off[:-1], b.ext_off + a.ext_off[-1]]), np.concatenate([a.ext, b.ext]),
                  np.concatenate([a.left, b.left]), np.concatenate([a.right, b.right]))
    for m, op in ((1, 5), (2, 6)):
        want = O.flip_all(u, k, m)
        g, _ = rfx.extras_operator(op, u, np.array([0, u.n], np.int64), k)
        same_records(g, want)
        u = O.sort_records(want)
        ps = O.partition_starts(u.key, P)
        want, wps = O.key_filter(O.OP_LONGER_OF_KEY, u, ps, k)
        g, gps = rfx.extras_operator(O.OP_LONGER_OF_KEY, u, ps, k)
        same_records(g, want)
        assert np.array_equal(gps, wps)
        u = want


@pytest.mark.parametrize("k,P", [(63, 4), (47, 8), (33, 1)])
@pytest.mark.parametrize("extras", [0, 1])
def test_driver_w_with_and_without_extras(rfx, torch_mod, planted, k, P, extras):
    import reflexiv_amd
    torch = torch_mod
    km, counts = filtered_kmers(planted["bases"], planted["read_off"], k, 2)
    dk, dc = to_dev(torch, km), to_dev(torch, counts)
    torch.cuda.synchronize()
    text, nc, trace = rfx.assemble_w_dev(dk.data_ptr(), dc.data_ptr(), len(counts),
                                         reflexiv_amd.default_params(k=k, min_cov=2, partitions=P, min_contig=100, extras=extras))
    otext, onc, otrace, _ = O.assemble_from_counts(km, counts, O.default_params(k=k, min_cov=2, partitions=P, min_contig=100, extras=extras))
    assert trace == otrace and nc == onc and text == otext


def test_extras_on_a_large_record_set(rfx, torch_mod):
    """the split happens while the record set is still large (big-pass kernels): 1.2 Mbp genome, k = 63"""
    import reflexiv_amd
    torch = torch_mod
    seed, G, n_reads, L, k, cov, P = 9, 1_200_000, 320_000, 150, 63, 3, 8
    O.set_threads(O.host_cores())
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, _, _ = O.count_reads_omp(bases, off, k, cov, cap=1 << 23)
    O.set_threads(1)
    km = O.counter_to_asm_w(wk, k)
    c32 = wc.astype(np.int32)
    dk, dc = to_dev(torch, km), to_dev(torch, c32)
    torch.cuda.synchronize()
    text, nc, trace = rfx.assemble_w_dev(dk.data_ptr(), dc.data_ptr(), len(c32), reflexiv_amd.default_params(k=k, min_cov=cov, partitions=P))
    otext, onc, otrace, _ = O.assemble_from_counts(km, c32, O.default_params(k=k, min_cov=cov, partitions=P))
    assert trace == otrace and nc == onc and text == otext
    assert trace[12] > 4096                       # the record set was beyond the small-pass kernel when it was split


def test_cpp_host_counter_then_run_kmerc_k63(tmp_path, planted):
    """the only working k > 31 route of the reference: `counter -kmer 63` writes "KMER,count"
    rows (P/ReflexivDataFrameCounter64.java:222-233), `run -kmerc ... -kmer 63` parses them into 31-base words
    (KmerBinarizer, P/ReflexivDSMain64.java:10772-10836) and assembles (assemblyFromKmer :374-826) into
    <outfile>/Assemble_63 -- through the C++ mirror of the driver; rows shuffled and in the legacy tuple text."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    k = 63
    fq = str(tmp_path / "p.fq")
    bases, off = planted["bases"], planted["read_off"]
    with open(fq, "w") as f:
        for i in range(len(off) - 1):
            s = bytes(bases[off[i]:off[i + 1]]).decode()
            f.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
    out = str(tmp_path / "cnt")
    subprocess.check_call([host, "counter", "-fastq", fq, "-outfile", out, "-kmer", str(k), "-cover", "2"])
    cdir = os.path.join(out, f"Count_{k}")
    lines = open(os.path.join(cdir, "part-00000.csv")).read().split("\n")[:-1]
    wk, wc, _ = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 2)
    assert lines == [O.kmer_text_w(a, k) + "," + str(int(c)) for a, c in zip(wk, wc)]
    rng = np.random.default_rng(1)
    legacy = ["(" + l + ")" for l in lines]
    rng.shuffle(legacy)
    open(os.path.join(cdir, "part-00000.csv"), "w").write("\n".join(legacy) + "\n")
    res = str(tmp_path / "asm")
    subprocess.check_call([host, "run", "-kmerc", cdir, "-outfile", res, "-kmer", str(k), "-cover", "2", "-mincontig", "100",
                           "--logical-partitions", "4"])
    want, _, _, _ = O.assemble_from_counts(O.counter_to_asm_w(wk, k), wc.astype(np.int32),
                                           O.default_params(k=k, min_cov=2, partitions=4, min_contig=100))
    assert open(os.path.join(res, f"Assemble_{k}", "part-00000")).read() == want
