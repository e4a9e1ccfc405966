"""GPU parity tests: every operator of the hot path, called through the C ABI
(libreflexiv_hip.so), against the CPU oracle on the same inputs -- bit-exact (integer work).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import hashlib
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
def ex(golden_dir):
    return np.load(os.path.join(golden_dir, "example.npz"))


@pytest.fixture(scope="module")
def planted(golden_dir):
    return np.load(os.path.join(golden_dir, "planted.npz"))


def same_records(a, b):
    assert a.n == b.n
    for f in ("key", "marker", "ext_off", "ext", "left", "right"):
        x, y = getattr(a, f), getattr(b, f)
        assert np.array_equal(np.asarray(x), np.asarray(y)), f


def test_native_library_is_loaded(rfx):
    """The HIP extension must be the thing that runs (no eager / CPU fallback exists)."""
    import reflexiv_amd
    maps = open("/proc/self/maps").read()
    assert "libreflexiv_hip.so" in maps
    assert reflexiv_amd.lib().rfx_version() >= 100


# ------------------------------------------------------------------ K1 extraction

def test_extract_example(rfx, ex):
    got = rfx.ReverseComplementKmerBinaryExtraction(ex["bases"], ex["read_off"], 31)
    want = O.extract_canon(ex["bases"], ex["read_off"], 31)
    assert len(got) == 161_000 and np.array_equal(got, want)
    assert np.array_equal(got[:280], ex["k1_first4"])


@pytest.mark.parametrize("k,fc,ec", [(31, 0, 0), (31, 3, 5), (21, 0, 0), (15, 2, 0), (5, 0, 1)])
def test_extract_ragged_and_edge_reads(rfx, k, fc, ec):
    rng = np.random.default_rng(k * 100 + fc * 10 + ec)
    lens = np.concatenate([[0, 1, k - 1, k, k + 1, k + 2, k + 3, 64, 65, 96, 97, 200, 31, 32, 33],
                           rng.integers(0, 300, 200)])
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    alphabet = np.frombuffer(b"ACGTNacgt", np.uint8)       # N and lower case -> code 3
    bases = alphabet[rng.integers(0, len(alphabet), off[-1])]
    got = rfx.ReverseComplementKmerBinaryExtraction(bases, off, k, fc, ec)
    want = O.extract_canon(bases, off, k, fc, ec)
    assert np.array_equal(got, want)


def test_extract_empty(rfx):
    got = rfx.ReverseComplementKmerBinaryExtraction(np.zeros(0, np.uint8), np.zeros(1, np.int64), 31)
    assert len(got) == 0
    got = rfx.ReverseComplementKmerBinaryExtraction(np.frombuffer(b"ACGT", np.uint8), np.array([0, 4]), 31)
    assert len(got) == 0


# ------------------------------------------------------------ K2/K3 count + filter

@pytest.mark.parametrize("min_cov,twin", [(1, O.TWIN_DS), (2, O.TWIN_DS), (3, O.TWIN_RDD), (1, O.TWIN_RDD)])
def test_count_filter_example(rfx, ex, min_cov, twin):
    km = O.extract_canon(ex["bases"], ex["read_off"], 31)
    keys, counts, nd = rfx.KmerCounting_and_CoverageFilter(km, min_cov, 10_000_000, twin)
    wk, wc, wd = O.count_filter(km, min_cov, 10_000_000, twin)
    assert nd == wd == 43_748
    assert np.array_equal(keys, wk) and np.array_equal(counts, wc)


def test_count_filter_max_cov_and_collisions(rfx):
    """heavy hitters, a max-coverage cut and many duplicates of few keys (hash-table collisions)."""
    rng = np.random.default_rng(5)
    few = rng.integers(0, 1 << 62, 50, dtype=np.uint64)
    km = np.concatenate([np.repeat(few, rng.integers(1, 3000, 50)),
                         rng.integers(0, 1 << 62, 300_000, dtype=np.uint64),
                         np.full(70_000, 12345, np.uint64), np.zeros(10, np.uint64)])
    rng.shuffle(km)
    for mn, mx in ((1, 10_000_000), (2, 1000), (5, 69_999), (3, 70_000)):
        keys, counts, nd = rfx.KmerCounting_and_CoverageFilter(km, mn, mx)
        wk, wc, wd = O.count_filter(km, mn, mx)
        assert nd == wd and np.array_equal(keys, wk) and np.array_equal(counts, wc)


def test_count_filter_empty_and_single(rfx):
    k, c, d = rfx.KmerCounting_and_CoverageFilter(np.zeros(0, np.uint64), 1)
    assert len(k) == 0 and d == 0
    k, c, d = rfx.KmerCounting_and_CoverageFilter(np.array([7, 7, 7], np.uint64), 2)
    assert list(k) == [7] and list(c) == [3] and d == 1


def test_count_filter_multi_level(rfx):
    """enough instances for two radix levels and low-coverage leaves that need the split fallback."""
    rng = np.random.default_rng(11)
    base = rng.integers(0, 1 << 62, 3_000_000, dtype=np.uint64)
    km = np.concatenate([base, base[:1_500_000], base[:400_000], base[:400_000]])
    rng.shuffle(km)
    keys, counts, nd = rfx.KmerCounting_and_CoverageFilter(km, 2)
    wk, wc, wd = O.count_filter(km, 2)
    assert nd == wd and np.array_equal(keys, wk) and np.array_equal(counts, wc)


# ---------------------------------------------------- K4..K9 and the extend passes

def run_operator_chain(rfx, keys, counts, k, P, min_err, twin, n_pass=8):
    """GPU and oracle side by side, every operator fed with the ORACLE's previous output
    so one mismatch does not cascade; returns nothing, asserts record for record."""
    o = O.rc_expand_subkmer(keys, counts, k)
    g = rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(keys, counts, k)
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
        nxt, nps = O.extend_pass(cur, ps, k, twin)
        stage = 0 if i < 4 else (1 if i == 4 else 2)
        g, gps = rfx.ExtendReflexivKmer(cur, ps, k, twin, stage)
        same_records(g, nxt)
        assert np.array_equal(gps, nps)
        cur = nxt
    return cur


@pytest.mark.parametrize("P,twin", [(4, O.TWIN_DS), (1, O.TWIN_RDD), (7, O.TWIN_DS)])
def test_operator_chain_example(rfx, ex, P, twin):
    run_operator_chain(rfx, ex["keys_cov3"], ex["counts_cov3"], 31, P, 8, twin, n_pass=10)


@pytest.mark.parametrize("k,twin,min_err", [(31, O.TWIN_DS, 8), (31, O.TWIN_RDD, 8), (31, O.TWIN_DS, 0),
                                            (21, O.TWIN_DS, 8)])
def test_operator_chain_planted_bubbles(rfx, planted, k, twin, min_err):
    """SNP bubble + repeat: exercises the left/right >= 0 (bubble distance) branches."""
    run_operator_chain(rfx, planted[f"k{k}_keys"], planted[f"k{k}_counts"], k, 4, min_err, twin, n_pass=12)


def test_late_passes_with_long_extensions(rfx, ex):
    """few records, many words each: the long-record emit path (one workgroup per record)."""
    prm = O.default_params(min_cov=3, partitions=4)
    cur = O.rc_expand_subkmer(ex["keys_cov3"], ex["counts_cov3"], 31)
    cur = O.sort_records(cur)
    cur, ps = O.fork_filter_forward(cur, O.partition_starts(cur.key, 4), 31, 8, O.TWIN_DS)
    cur = O.sort_records(O.reflect_from_forward(cur, 31))
    cur, ps = O.fork_filter_reflected(cur, O.partition_starts(cur.key, 4), 31, 8, O.TWIN_DS)
    cur = O.random_reflection(cur, ps, 31)
    for i in range(26):
        cur = O.sort_records(cur)
        ps = O.partition_starts(cur.key, 4)
        nxt, nps = O.extend_pass(cur, ps, 31, O.TWIN_DS)
        if i >= 12:
            g, gps = rfx.ExtendReflexivKmer(cur, ps, 31, O.TWIN_DS, 2)
            same_records(g, nxt)
            assert np.array_equal(gps, nps)
        cur = nxt
    assert np.diff(cur.ext_off).max() > 20          # really did exercise long records
    text, nc = rfx.KmerToContig(cur, 31, 500, O.TWIN_RDD)
    assert (text, nc) == O.contigs_text(cur, 31, 500, O.TWIN_RDD)


def test_extend_pass_empty_and_singletons(rfx):
    empty = O.Records(np.zeros(0, np.uint64), np.zeros(0, np.int32), np.zeros(1, np.int64),
                      np.zeros(0, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.int32))
    ps = np.zeros(3, np.int64)
    g, gps = rfx.ExtendReflexivKmer(empty, ps, 31)
    assert g.n == 0 and list(gps) == [0, 0, 0]
    one = O.Records.from_single(np.array([5], np.uint64), np.array([1], np.int32), np.array([6], np.uint64),
                                np.array([-1], np.int32), np.array([-1], np.int32))
    ps = np.array([0, 1], np.int64)
    want, wps = O.extend_pass(one, ps, 31)
    g, gps = rfx.ExtendReflexivKmer(one, ps, 31, O.TWIN_DS, 0)
    same_records(g, want)


# ----------------------------------------------------------------- whole driver

def dev_assemble(rfx, torch, keys, counts, prm):
    dk = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64)).cuda()
    dc = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
    torch.cuda.synchronize()
    return rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), len(keys), prm)


@pytest.mark.parametrize("P", [1, 2, 4, 8])
@pytest.mark.parametrize("tn", ["ds", "rdd"])
def test_assemble_example_matches_golden(rfx, torch_mod, ex, P, tn):
    import reflexiv_amd
    twin = O.TWIN_DS if tn == "ds" else O.TWIN_RDD
    prm = reflexiv_amd.default_params(min_cov=3, partitions=P, twin=twin)
    text, nc, trace = dev_assemble(rfx, torch_mod, ex["keys_cov3"], ex["counts_cov3"], prm)
    assert text == str(ex[f"contigs_{tn}_P{P}"])
    assert trace == [int(x) for x in ex[f"trace_{tn}_P{P}"]]


@pytest.mark.parametrize("P", [16, 64, 200])
def test_assemble_with_the_coalesce_rule(rfx, torch_mod, ex, planted, P):
    """rfx_params.coalesce (P/ReflexivMain.java:277-281): the partition number drops to P/4+1 at a check that finds
    <= 20 records per partition; both the big-pass loop and the small-pass kernel carry it"""
    import reflexiv_amd
    for keys, counts, cov, mc in ((ex["keys_cov3"], ex["counts_cov3"], 3, 500), (planted["k31_keys"], planted["k31_counts"], 2, 100)):
        prm = reflexiv_amd.default_params(min_cov=cov, partitions=P, coalesce=1, min_contig=mc)
        text, nc, trace = dev_assemble(rfx, torch_mod, keys, counts, prm)
        otext, onc, otrace, _ = O.assemble_from_counts(keys, counts, O.default_params(min_cov=cov, partitions=P, coalesce=1, min_contig=mc))
        assert trace == otrace and (text, nc) == (otext, onc)


def test_documented_known_answer_on_gpu(rfx, torch_mod, ex):
    """docs/example.html:303,320-343 end to end on the GPU: reads -> k-mers -> contigs."""
    import reflexiv_amd
    km = rfx.ReverseComplementKmerBinaryExtraction(ex["bases"], ex["read_off"], 31)
    keys, counts, nd = rfx.KmerCounting_and_CoverageFilter(km, 3)
    assert len(keys) == 4612
    prm = reflexiv_amd.default_params(min_cov=3, partitions=4, twin=O.TWIN_RDD)
    text, nc, trace = dev_assemble(rfx, torch_mod, keys, counts, prm)
    seqs = ["".join(b.split("\n")[1:]) for b in text.split(">")[1:]]
    assert sorted(len(s) for s in seqs) == [4558, 4558]
    hit = [s for s in seqs if s.startswith(str(ex["doc_prefix1200"]))]
    assert len(hit) == 1
    assert hashlib.sha256(hit[0].encode()).hexdigest() == \
        "245baebd8b5b681f639217f31d647d9fcd03adfeef7f6d5ef10edd5cc12ae62c"
    assert text.split("\n")[0] == str(ex["doc_header"])


@pytest.mark.parametrize("tn", ["ds", "rdd"])
def test_assemble_planted_matches_golden(rfx, torch_mod, planted, tn):
    import reflexiv_amd
    twin = O.TWIN_DS if tn == "ds" else O.TWIN_RDD
    prm = reflexiv_amd.default_params(k=31, min_cov=2, partitions=4, twin=twin, min_contig=100)
    text, nc, trace = dev_assemble(rfx, torch_mod, planted["k31_keys"], planted["k31_counts"], prm)
    assert text == str(planted[f"k31_{tn}_contigs"])
    assert trace == [int(x) for x in planted[f"k31_{tn}_trace"]]


# ------------------------------------------------ synthetic reads + fused count path

def packed_to_ascii(words, n_reads, wpr, read_len):
    w = words.reshape(n_reads, wpr)
    out = np.empty((n_reads, read_len), np.uint8)
    nuc = np.frombuffer(b"ACGT", np.uint8)
    for j in range(read_len):
        out[:, j] = nuc[((w[:, j // 32] >> np.uint64(62 - 2 * (j % 32))) & np.uint64(3)).astype(np.int64)]
    return out.reshape(-1)


def make_reads_dev(rfx, torch, seed, genome_len, n_reads, read_len, first_read=0):
    wpr = (read_len + 31) // 32
    dg = torch.empty((genome_len + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(seed, genome_len, dg.data_ptr())
    rfx.synth_reads_dev(seed, dg.data_ptr(), genome_len, first_read, n_reads, read_len, wpr, dw.data_ptr())
    rfx.sync()
    return dg, dw, wpr


def test_synth_generator_matches_oracle(rfx, torch_mod):
    seed, G, n, L = 99, 50_000, 4000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch_mod, seed, G, n, L, first_read=10)
    g = O.synth_genome(seed, G)
    assert np.array_equal(dg.cpu().numpy().view(np.uint64), g)
    want, _ = O.synth_reads(seed, g, G, 10, n, L)
    got = packed_to_ascii(dw.cpu().numpy().view(np.uint64), n, wpr, L)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n_reads,L,k,min_cov", [(3000, 100, 31, 2), (120_000, 150, 31, 3), (50_000, 150, 21, 2)])
def test_fused_count_from_packed_reads(rfx, torch_mod, n_reads, L, k, min_cov):
    torch = torch_mod
    seed, G = 1234, 200_000
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    nk = rfx.kmers_per_read(L, k)
    N = nk * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda")
    dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
    assert inst == N
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon(bases, off, k)
    assert len(km) == N
    wk, wc, wd = O.count_filter(km, min_cov)
    assert nd == wd and m == len(wk)
    assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk)
    assert np.array_equal(dc[:m].cpu().numpy(), wc)


def test_fused_count_properties_at_scale(rfx, torch_mod):
    """Size-independent properties on a workload the oracle would take minutes for:
    counts of all distinct k-mers sum to N; output strictly ascending; raising min_cov
    yields a subset with identical counts."""
    torch = torch_mod
    seed, G, n_reads, L, k = 77, 2_000_000, 2_000_000, 150, 31
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N // 2, dtype=torch.int64, device="cuda")
    dc = torch.empty(N // 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m1, nd1, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N // 2, 1)
    assert inst == N and m1 == nd1
    k1, c1 = dk[:m1].clone(), dc[:m1].clone()
    assert int(c1.sum(dtype=torch.int64)) == N
    assert bool((k1[1:] > k1[:-1]).all())
    m3, nd3, _ = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N // 2, 3)
    assert nd3 == nd1
    sel = c1 >= 3
    assert m3 == int(sel.sum())
    assert torch.equal(dk[:m3], k1[sel]) and torch.equal(dc[:m3], c1[sel])


def test_bucket_by_owner_partitions_kmer_space(rfx, torch_mod):
    """multi-GPU support kernel: every instance lands in exactly one owner bucket, and an owner's
    bucket holds exactly the k-mers whose hash falls in its shard."""
    torch = torch_mod
    seed, G, n_reads, L, k, owners = 5, 100_000, 20_000, 150, 31, 8
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    out = torch.empty(N, dtype=torch.int64, device="cuda")
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    h = rfx.bucket_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, out.data_ptr(), N, doff.data_ptr())
    assert h[0] == 0 and h[-1] == N and np.all(np.diff(h) > 0)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = np.sort(O.extract_canon(bases, off, k))
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(np.sort(got), km)
    from tests.test_dist_gloo import owner_of
    for o in range(owners):
        assert np.all(owner_of(got[h[o]:h[o + 1]], owners) == o)
    # per-owner counting of the buckets == global counting restricted to the bucket
    wk, wc, _ = O.count_filter(km, 2)
    parts = []
    for o in range(owners):
        seg = got[h[o]:h[o + 1]]
        kk, cc, _ = O.count_filter(seg, 2)
        parts.append((kk, cc))
    allk = np.concatenate([p[0] for p in parts]); allc = np.concatenate([p[1] for p in parts])
    order = np.argsort(allk, kind="stable")
    assert np.array_equal(allk[order], wk) and np.array_equal(allc[order], wc)


def test_sort_pairs_is_stable(rfx, torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(3)
    for n, bits in ((1, 60), (100, 8), (2048, 60), (2049, 60), (300_000, 60), (1_000_000, 20)):
        keys = rng.integers(0, 1 << min(bits, 12 if n > 2048 else bits), n, dtype=np.uint64)   # many ties
        vals = np.arange(n, dtype=np.uint32)
        dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
        tk = torch.empty_like(dk); tv = torch.empty_like(dv)
        torch.cuda.synchronize()
        rfx.sort_pairs_dev(dk.data_ptr(), dv.data_ptr(), n, bits, tk.data_ptr(), tv.data_ptr())
        rfx.sync()
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(dk.cpu().numpy().view(np.uint64), keys[order])
        assert np.array_equal(dv.cpu().numpy().view(np.uint32), vals[order])


def test_sort_pairs_top_digit_path_is_stable(rfx, torch_mod):
    """2 K .. 400 K pairs whose top digit is spread take one global pass on that digit and finish the buckets
    on chip; ties inside a bucket keep arrival order, odd key widths work, and one oversized bucket (skewed
    keys) sends the call down the LSD passes."""
    torch = torch_mod
    rng = np.random.default_rng(31)
    cases = []
    for n, bits in ((2049, 60), (30_000, 60), (130_000, 62), (131_073, 60), (390_000, 64), (50_000, 13), (50_000, 9), (20_000, 37)):
        top = rng.integers(0, 256, n, dtype=np.uint64) << np.uint64(max(0, bits - 8))
        low = rng.integers(0, 1 << min(bits - 8, 4) if bits > 8 else 1, n, dtype=np.uint64)         # many ties inside a bucket
        cases.append((n, bits, (top | low) & np.uint64((1 << bits) - 1 if bits < 64 else 0xFFFFFFFFFFFFFFFF)))
        cases.append((n, bits, rng.integers(0, 1 << min(bits, 63), n, dtype=np.uint64)))              # spread keys, few ties
    skew = rng.integers(0, 1 << 52, 100_000, dtype=np.uint64); skew[:3000] |= np.uint64(0xAB) << np.uint64(52)
    cases.append((100_000, 60, skew))                                                                 # bucket 0 holds 97 000 pairs
    for n, bits, keys in cases:
        vals = np.arange(n, dtype=np.uint32)
        dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
        tk = torch.empty_like(dk); tv = torch.empty_like(dv)
        torch.cuda.synchronize()
        rfx.sort_pairs_dev(dk.data_ptr(), dv.data_ptr(), n, bits, tk.data_ptr(), tv.data_ptr())
        rfx.sync()
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(dk.cpu().numpy().view(np.uint64), keys[order]), (n, bits)
        assert np.array_equal(dv.cpu().numpy().view(np.uint32), vals[order]), (n, bits)


def test_sort_pairs_two_level_msd_is_stable(rfx, torch_mod):
    """more than 400 K pairs: two stable MSD levels, then the final buckets on chip.  Spread keys with few ties, keys
    with many ties inside a final bucket, values that are NOT the arrival index (the multi-word key sort feeds a
    permutation), empty first-level buckets, odd key widths, and skew that sends the call down the LSD passes."""
    torch = torch_mod
    rng = np.random.default_rng(77)
    cases = []
    for n, bits in ((400_000, 60), (1_000_003, 60), (3_000_000, 62), (9_300_000, 60), (2_000_000, 64), (700_000, 17), (5_000_000, 33)):
        cases.append((bits, rng.integers(0, 1 << min(bits, 63), n, dtype=np.uint64)))
    # ties: only 5000 distinct keys among 2 M
    pool = rng.integers(0, 1 << 60, 5000, dtype=np.uint64)
    cases.append((60, pool[rng.integers(0, 5000, 2_000_000)]))
    # empty first-level buckets: top byte only takes 3 values -> buckets far larger than a tile -> LSD fallback
    few = (rng.integers(0, 3, 1_500_000, dtype=np.uint64) << np.uint64(52)) | rng.integers(0, 1 << 40, 1_500_000, dtype=np.uint64)
    cases.append((60, few))
    # half of the first-level buckets empty, the others spread
    half = ((rng.integers(0, 128, 2_500_000, dtype=np.uint64) * np.uint64(2)) << np.uint64(52)) | rng.integers(0, 1 << 52, 2_500_000, dtype=np.uint64)
    cases.append((60, half))
    for bits, keys in cases:
        n = len(keys)
        vals = rng.permutation(n).astype(np.uint32)
        dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
        tk = torch.empty_like(dk); tv = torch.empty_like(dv)
        torch.cuda.synchronize()
        rfx.sort_pairs_dev(dk.data_ptr(), dv.data_ptr(), n, bits, tk.data_ptr(), tv.data_ptr())
        rfx.sync()
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(dk.cpu().numpy().view(np.uint64), keys[order]), (n, bits)
        assert np.array_equal(dv.cpu().numpy().view(np.uint32), vals[order]), (n, bits)


def test_sort_pairs_three_level_msd_is_stable(rfx, torch_mod):
    """more than 2^26 pairs (the survivors of a human-scale share): three stable MSD levels, the final buckets on chip.
    Spread 62-bit keys with values that are a permutation; many ties (stability); and a skewed set -- a fifth of the keys
    on one top byte -- whose final buckets outgrow a tile, so the LSD passes finish from the regrouped state.  Checked
    on the device against torch's stable sort."""
    torch = torch_mod
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    n = (1 << 26) + 1_234_567
    for case in ("spread", "ties", "skew"):
        if case == "spread":
            keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
        elif case == "ties":
            pool = torch.randint(0, 1 << 62, (1 << 22,), dtype=torch.int64, device="cuda", generator=g)
            keys = pool[torch.randint(0, 1 << 22, (n,), device="cuda", generator=g)]
        else:
            keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
            hot = torch.rand(n, device="cuda", generator=g) < 0.2
            keys = torch.where(hot, (keys & ((1 << 30) - 1)) | (5 << 54), keys)
        vals = torch.randperm(n, device="cuda", generator=g).to(torch.int32)
        want_k, order = torch.sort(keys, stable=True)
        want_v = vals[order]
        dk, dv = keys.clone(), vals.clone()
        tk, tv = torch.empty_like(dk), torch.empty_like(dv)
        torch.cuda.synchronize()
        rfx.sort_pairs_dev(dk.data_ptr(), dv.data_ptr(), n, 62, tk.data_ptr(), tv.data_ptr())
        rfx.sync()
        assert bool((dk == want_k).all()), case
        assert bool((dv == want_v).all()), case
        del want_k, order, want_v, dk, dv, tk, tv, keys, vals


# ------------------------------------------------ C++ host mirror of the reference driver

def write_fastq(path, bases, read_off, gz=False):
    import gzip
    opener = gzip.open if gz else open
    with opener(path, "wb") as fh:
        for i in range(len(read_off) - 1):
            seq = bytes(bases[read_off[i]:read_off[i + 1]])
            qual = b"@" * len(seq)                       # quality lines starting with '@' on purpose
            fh.write(b"@r%d/1\n" % i + seq + b"\n+\n" + qual + b"\n")


@pytest.mark.parametrize("gz", [False, True])
def test_cpp_host_run_matches_documented_example(tmp_path, ex, gz):
    """`reflexiv_host run -fastq ... -kmer 31 -cover 3` (C++ mirror of ReflexivMain.assembly()
    calling one C-ABI entry per Spark operator) reproduces the documented contigs."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    assert os.path.exists(host), "build with reflexiv_amd.build()"
    fq = str(tmp_path / ("ex.fq.gz" if gz else "ex.fq"))
    write_fastq(fq, ex["bases"], ex["read_off"], gz)
    out = str(tmp_path / "result")
    subprocess.check_call([host, "run", "-fastq", fq, "-outfile", out, "-kmer", "31", "-cover", "3",
                           "--logical-partitions", "4", "--twin", "rdd"])
    text = open(os.path.join(out, "part-00000")).read()
    assert os.path.exists(os.path.join(out, "_SUCCESS"))
    assert text == str(ex["contigs_rdd_P4"])
    assert text.startswith(str(ex["doc_header"]) + "\n")


def test_cpp_host_counter_then_run_kmerc(tmp_path, ex):
    """`counter` writes Count_<k>/part-*.csv rows "KMER,count" (P/ReflexivDataFrameCounter.java:222-233);
    `run -kmerc` parses them (KmerBinarizer, P/ReflexivDSMain.java:3883-3931) and assembles -- the
    working k-mer-count hand-off of the reference."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    fq = str(tmp_path / "ex.fq")
    write_fastq(fq, ex["bases"], ex["read_off"])
    out = str(tmp_path / "cnt")
    subprocess.check_call([host, "counter", "-fastq", fq, "-outfile", out, "-kmer", "31", "-cover", "3"])
    cdir = os.path.join(out, "Count_31")
    lines = open(os.path.join(cdir, "part-00000.csv")).read().split("\n")[:-1]
    assert len(lines) == 4612
    nuc = "ACGT"
    want = ["".join(nuc[(int(k) >> (2 * (30 - j))) & 3] for j in range(31)) + "," + str(int(c))
            for k, c in zip(ex["keys_cov3"], ex["counts_cov3"])]
    assert lines == want
    # legacy tuple text + shuffled row order must parse to the same assembly
    rng = np.random.default_rng(0)
    legacy = ["(" + l.replace(",", ",") + ")" for l in lines]
    rng.shuffle(legacy)
    open(os.path.join(cdir, "part-00000.csv"), "w").write("\n".join(legacy) + "\n")
    res = str(tmp_path / "asm")
    subprocess.check_call([host, "run", "-kmerc", cdir, "-outfile", res, "-kmer", "31", "-cover", "3",
                           "--logical-partitions", "4", "--twin", "rdd"])
    assert open(os.path.join(res, "part-00000")).read() == str(ex["contigs_rdd_P4"])


def test_sharded_count_engine_world1(rfx, torch_mod):
    """The multi-GPU code path (owner buckets -> [all-to-all] -> count of an explicit k-mer
    array) on one GPU must equal the fused single-GPU path."""
    torch = torch_mod
    from reflexiv_amd import dist as rd
    seed, G, n_reads, L, k = 21, 300_000, 150_000, 150, 31
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=k)
    keys, counts, tot = rd.sharded_count(rd.HipEngine(rfx), reads, 3, 10_000_000, 0)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 3)
    assert tot == [inst, nd, m]
    assert torch.equal(keys, dk[:m]) and torch.equal(counts, dc[:m])


def test_deep_coverage_recovers_exactly_the_genome_kmers(rfx, torch_mod):
    """At ~200x coverage with a cut-off of 10 the survivors must be exactly the canonical k-mers
    of the genome (error k-mers never reach 10, genomic ones never fall below it): a
    size-independent check of the whole count stage with ~1000 leaves per workgroup, which is
    where barrier races in the persistent kernels show up."""
    torch = torch_mod
    seed, G, n_reads, L, k = 5, 1_000_000, 1_400_000, 150, 31
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N // 8, dtype=torch.int64, device="cuda")
    dc = torch.empty(N // 8, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N // 8, 10)
    g = O.synth_genome(seed, G)
    nuc = np.frombuffer(b"ACGT", np.uint8)
    gb = nuc[((g[np.arange(G) >> 5] >> (np.uint64(62) - np.uint64(2) * (np.arange(G, dtype=np.uint64) & np.uint64(31))))
              & np.uint64(3)).astype(np.int64)]
    allk = O.extract_canon(gb, np.array([0, G]), k)
    want = np.unique(allk)
    core = np.unique(allk[400:-400])          # the genome's two ends are sampled thinly
    got = dk[:m].cpu().numpy().view(np.uint64)
    assert np.all(got[1:] > got[:-1])
    assert np.isin(got, want).all()           # no error k-mer survives
    assert np.isin(core, got).all()           # no genomic k-mer is lost
    assert len(want) - m < 200
    assert int(dc[:m].min()) >= 10


@pytest.mark.parametrize("k,owners", [(31, 4), (31, 32), (31, 64), (25, 3), (21, 24)])
def test_bucket_records_by_owner_then_count(rfx, torch_mod, k, owners):
    """multi-GPU record path on one GPU: super-k-mer records bucketed by owner (up to the 64 bins that 8 GPUs x 8
    generations use); counting every owner's bucket on its own gives disjoint shards whose union is the
    oracle's global count."""
    torch = torch_mod
    seed, G, n_reads, L = 8, 150_000, 40_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    args = (dw.data_ptr(), n_reads, wpr, L, k, owners)
    nrec, h = rfx.bucket_records_by_owner_dev(*args, 0, 0, doff.data_ptr())
    assert h is None and nrec > 0
    recs = torch.empty(2 * nrec, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nrec2, h = rfx.bucket_records_by_owner_dev(*args, recs.data_ptr(), nrec, doff.data_ptr())
    N = rfx.kmers_per_read(L, k) * n_reads
    assert nrec2 == nrec and h[0] == 0 and h[-1] == nrec and N / nrec > 3      # several windows per record
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter(O.extract_canon(bases, off, k), 2)
    allk, allc, nd_sum = [], [], 0
    for o in range(owners):
        seg = recs[2 * h[o]: 2 * h[o + 1]].contiguous()
        n_o = int(h[o + 1] - h[o])
        dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, d = rfx.count_records_dev(seg.data_ptr(), n_o, 0, k, dk.data_ptr(), dc.data_ptr(), N, 2)
        kk = dk[:m].cpu().numpy().view(np.uint64)
        assert np.all(kk[1:] > kk[:-1])
        allk.append(kk); allc.append(dc[:m].cpu().numpy()); nd_sum += d
    allk = np.concatenate(allk); allc = np.concatenate(allc)
    assert nd_sum == wd and len(np.unique(allk)) == len(allk)
    order = np.argsort(allk, kind="stable")
    assert np.array_equal(allk[order], wk) and np.array_equal(allc[order], wc)


@pytest.mark.parametrize("k", [28, 29, 30, 27, 21])
def test_fused_count_other_k(rfx, torch_mod, k):
    """k = 28..31 take the super-k-mer record path (W = k-12 = 16..19), smaller k the k-mer path."""
    torch = torch_mod
    seed, G, n_reads, L = 31 + k, 100_000, 30_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 2)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter(O.extract_canon(bases, off, k), 2)
    assert (m, nd) == (len(wk), wd)
    assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk) and np.array_equal(dc[:m].cpu().numpy(), wc)


def test_fused_count_with_clips_and_short_reads(rfx, torch_mod):
    """front/end clips shift the window origin inside the packed words; reads shorter than k+2 emit nothing."""
    torch = torch_mod
    for L, k, fc, ec in ((100, 31, 3, 5), (64, 31, 0, 0), (40, 31, 2, 1), (33, 31, 0, 0), (32, 31, 0, 0), (150, 29, 7, 0)):
        seed, G, n_reads = 1000 + L, 50_000, 5_000
        wpr = (L + 31) // 32
        dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
        dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        rfx.synth_genome_dev(seed, G, dg.data_ptr())
        rfx.synth_reads_dev(seed, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr())
        rfx.sync()
        N = max(1, rfx.kmers_per_read(L, k, fc, ec) * n_reads)
        dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 1,
                                          front_clip=fc, end_clip=ec)
        g = O.synth_genome(seed, G)
        bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
        km = O.extract_canon(bases, off, k, fc, ec)
        wk, wc, wd = O.count_filter(km, 1)
        assert inst == len(km) and (m, nd) == (len(wk), wd), (L, k, fc, ec)
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk) and np.array_equal(dc[:m].cpu().numpy(), wc)


def test_assemble_long_contigs_whole_grid_emit(rfx, torch_mod):
    """640 kbp genome at 40x: contigs of several hundred kbp (> 8192 extension words), which take
    the whole-grid emission path of the late extend passes; the contig text must equal the oracle's."""
    import reflexiv_amd
    torch = torch_mod
    seed, G, n_reads, L, k = 3, 640_000, 170_000, 150, 31
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N // 4, dtype=torch.int64, device="cuda"); dc = torch.empty(N // 4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N // 4, 3)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter(O.extract_canon(bases, off, k), 3)
    assert (m, nd) == (len(wk), wd)
    for P, twin in ((8, O.TWIN_DS), (3, O.TWIN_RDD)):
        text, nc, trace = rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), m, reflexiv_amd.default_params(min_cov=3, partitions=P, twin=twin))
        otext, onc, otrace, _ = O.assemble_from_counts(wk, wc, O.default_params(min_cov=3, partitions=P, twin=twin))
        assert trace == otrace and nc == onc and text == otext
        assert max(int(h.split("-")[1]) for h in text.split("\n") if h.startswith(">")) > 31 * 8192


# ------------------------------------------------------------------ k > 31

def _ragged_reads(seed, n, lo, hi):
    rng = np.random.default_rng(seed)
    reads = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(lo, hi)), p=[.24, .24, .24, .24, .04]))
             for _ in range(n)]
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    return bases, off


@pytest.mark.gpu
@pytest.mark.parametrize("k", [33, 47, 63, 65, 95, 127])
@pytest.mark.parametrize("clips", [(0, 0), (3, 5)])
def test_wide_extraction_matches_oracle(rfx, k, clips):
    """ReverseComplementKmerBinaryExtractionFromDataset64 on ragged reads with N, reads shorter than k, clips."""
    bases, off = _ragged_reads(k, 300, 10, 260)
    fc, ec = clips
    want = O.extract_canon_w(bases, off, k, fc, ec)
    got = rfx.ReverseComplementKmerBinaryExtractionFromDataset64(bases, off, k, fc, ec)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [33, 63, 95])
def test_wide_count_filter_matches_oracle(rfx, k):
    bases, off = _ragged_reads(100 + k, 400, 100, 200)
    # repeat the reads so that counts > 1 occur, in a different order
    bases2 = np.concatenate([bases, bases[:off[200]], bases])
    off2 = np.concatenate([off, off[-1] + off[1:201], off[-1] + off[200] + off[1:]])
    km = O.extract_canon_w(bases2, off2, k)
    for min_cov, max_cov in ((1, 10_000_000), (2, 10_000_000), (3, 10_000_000), (1, 2)):
        wk, wc, wd = O.count_filter_w(km, k, min_cov, max_cov)
        gk, gc, gd = rfx.groupBy_count_filter_w(km, k, min_cov, max_cov)
        assert gd == wd and np.array_equal(gk, wk) and np.array_equal(gc, wc)
    # empty input
    gk, gc, gd = rfx.groupBy_count_filter_w(np.empty((0, k // 32 + 1), np.uint64), k, 1)
    assert len(gk) == 0 and gd == 0


@pytest.mark.gpu
def test_wide_golden_example(rfx, ex, golden_dir):
    gw = np.load(os.path.join(golden_dir, "wide.npz"))
    bases, off = ex["bases"], ex["read_off"]
    for k in (63, 47):
        km = rfx.ReverseComplementKmerBinaryExtractionFromDataset64(bases, off, k)
        assert len(km) == int(gw[f"k{k}_n_instances"])
        assert np.array_equal(km[:len(gw[f"k{k}_first4"])], gw[f"k{k}_first4"])
        keys, counts, nd = rfx.groupBy_count_filter_w(km, k, 3)
        assert nd == int(gw[f"k{k}_n_distinct"])
        assert np.array_equal(keys, gw[f"k{k}_keys_cov3"]) and np.array_equal(counts, gw[f"k{k}_counts_cov3"])


@pytest.mark.gpu
@pytest.mark.parametrize("n_reads,L,k,min_cov", [(4000, 150, 63, 2), (20_000, 150, 63, 3), (5000, 100, 47, 1)])
def test_wide_fused_count_from_packed_reads(rfx, torch_mod, n_reads, L, k, min_cov):
    torch = torch_mod
    seed, G = 31 + k, 30_000
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    W = k // 32 + 1
    N = rfx.kmers_per_read_w(L, k) * n_reads
    assert N == (L - k + 1) * n_reads
    dk = torch.empty(N * W, dtype=torch.int64, device="cuda")
    dc = torch.empty(N, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
    assert inst == N
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon_w(bases, off, k)
    wk, wc, wd = O.count_filter_w(km, k, min_cov)
    assert nd == wd and m == len(wk)
    assert np.array_equal(dk[:m * W].cpu().numpy().view(np.uint64).reshape(m, W), wk)
    assert np.array_equal(dc[:m].cpu().numpy(), wc)


@pytest.mark.gpu
def test_wide_count_properties_at_scale(rfx, torch_mod):
    """k = 63 on 1 M reads (8.8e7 instances): counts sum to N, keys strictly ascending word-wise,
    raising min_cov yields the subset with identical counts."""
    torch = torch_mod
    seed, G, n_reads, L, k = 63, 1_000_000, 1_000_000, 150, 63
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read_w(L, k) * n_reads
    cap = N // 2
    dk = torch.empty(cap * 2, dtype=torch.int64, device="cuda")
    dc = torch.empty(cap, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m1, nd1, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 1)
    assert inst == N and m1 == nd1
    k1 = dk[:2 * m1].clone().view(m1, 2); c1 = dc[:m1].clone()
    assert int(c1.sum()) == N
    # unsigned word-wise order: compare through a bias flip of the sign bit
    bias = torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
    a = k1 ^ bias
    asc = (a[1:, 0] > a[:-1, 0]) | ((a[1:, 0] == a[:-1, 0]) & (a[1:, 1] > a[:-1, 1]))
    assert bool(asc.all())
    m3, nd3, _ = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
    sel = c1 >= 3
    assert nd3 == nd1 and m3 == int(sel.sum())
    assert torch.equal(dk[:2 * m3].view(m3, 2), k1[sel]) and torch.equal(dc[:m3], c1[sel])


@pytest.mark.gpu
def test_cpp_host_counter_k63(tmp_path, ex, golden_dir):
    """`counter -kmer 63`: the C++ mirror of ReflexivDataFrameCounter64.assembly() (extraction,
    groupBy/count/filter, DSBinaryKmerToString) writes the rows the oracle predicts."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    gw = np.load(os.path.join(golden_dir, "wide.npz"))
    fq = str(tmp_path / "ex.fq")
    write_fastq(fq, ex["bases"], ex["read_off"])
    out = str(tmp_path / "cnt")
    subprocess.check_call([host, "counter", "-fastq", fq, "-outfile", out, "-kmer", "63", "-cover", "3"])
    lines = open(os.path.join(out, "Count_63", "part-00000.csv")).read().split("\n")[:-1]
    want = [O.kmer_text_w(k, 63) + "," + str(int(c)) for k, c in zip(gw["k63_keys_cov3"], gw["k63_counts_cov3"])]
    assert lines == want


@pytest.mark.gpu
def test_sharded_extend_hip_ops_world1(rfx, ex, planted):
    """reflexiv_amd.dist.sharded_assemble with the C ABI operators (HipOps) on one GPU: the same
    driver the multi-GPU path runs per rank, checked against the golden contigs."""
    from reflexiv_amd import dist as rd
    import reflexiv_amd
    ops = rd.HipOps(rfx)
    rng = np.random.default_rng(9)
    o = rng.permutation(len(ex["keys_cov3"]))
    for P, twin, tn in ((4, O.TWIN_DS, "ds"), (8, O.TWIN_RDD, "rdd")):
        prm = reflexiv_amd.default_params(min_cov=3, partitions=P, twin=twin)
        trace = []
        text, nc = rd.sharded_assemble(ops, ex["keys_cov3"][o], ex["counts_cov3"][o], prm, trace=trace)
        assert text == str(ex[f"contigs_{tn}_P{P}"])
        assert trace == [int(x) for x in ex[f"trace_{tn}_P{P}"]]
    prm = reflexiv_amd.default_params(k=31, min_cov=2, partitions=4, twin=O.TWIN_DS, min_contig=100)
    text, nc = rd.sharded_assemble(ops, planted["k31_keys"], planted["k31_counts"], prm)
    assert text == str(planted["k31_ds_contigs"])


@pytest.mark.gpu
def test_device_resident_sharded_extend_one_rank_rccl(rfx, torch_mod, ex, planted):
    """reflexiv_amd.dist.sharded_assemble_dev: records stay in HBM, every operator is a device-level C-ABI call, the
    splitter search runs on the device and the shuffle is all_to_all_single on device tensors -- on a one-rank RCCL
    group with the exchange forced, against the golden contigs, and on a 640 kbp genome against the oracle."""
    torch = torch_mod
    import torch.distributed as dist
    import reflexiv_amd
    from reflexiv_amd import dist as rd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rfx.use_stream(torch.cuda.current_stream().cuda_stream)
        ops = rd.HipDevOps(rfx)
        rng = np.random.default_rng(9)
        o = rng.permutation(len(ex["keys_cov3"]))
        dk = torch.from_numpy(ex["keys_cov3"][o].view(np.int64).copy()).cuda()
        dc = torch.from_numpy(ex["counts_cov3"][o].astype(np.int32)).cuda()
        for P, twin, tn, force in ((4, O.TWIN_DS, "ds", True), (8, O.TWIN_RDD, "rdd", True), (4, O.TWIN_RDD, "rdd", False)):
            prm = reflexiv_amd.default_params(min_cov=3, partitions=P, twin=twin)
            trace = []
            text, nc = rd.sharded_assemble_dev(ops, dk, dc, prm, trace=trace, force_exchange=force)
            assert text == str(ex[f"contigs_{tn}_P{P}"])
            assert trace == [int(x) for x in ex[f"trace_{tn}_P{P}"]]
        prm = reflexiv_amd.default_params(k=31, min_cov=2, partitions=4, twin=O.TWIN_DS, min_contig=100)
        text, nc = rd.sharded_assemble_dev(ops, torch.from_numpy(planted["k31_keys"].view(np.int64).copy()).cuda(),
                                           torch.from_numpy(planted["k31_counts"].astype(np.int32)).cuda(), prm, force_exchange=True)
        assert text == str(planted["k31_ds_contigs"])
        # a larger record set: 640 kbp at 40x
        seed, G, n_reads, L, k = 3, 640_000, 170_000, 150, 31
        dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
        N = rfx.kmers_per_read(L, k) * n_reads
        kk = torch.empty(N // 4, dtype=torch.int64, device="cuda"); cc = torch.empty(N // 4, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, kk.data_ptr(), cc.data_ptr(), N // 4, 3)
        prm = reflexiv_amd.default_params(min_cov=3, partitions=8)
        trace = []
        text, nc = rd.sharded_assemble_dev(ops, kk[:m], cc[:m], prm, trace=trace, force_exchange=True)
        otext, onc, otrace, _ = O.assemble_from_counts(kk[:m].cpu().numpy().view(np.uint64), cc[:m].cpu().numpy(),
                                                       O.default_params(min_cov=3, partitions=8))
        assert trace == otrace and (text, nc) == (otext, onc)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_cpp_host_counter_uses_the_counters_line_filter(tmp_path):
    """`counter` reads lines through DSFastqFilterOnlySeq (P/ReflexivDataFrameCounter.java:238-290), not
    through the assembler's 4-line grouping: a quality line that looks like sequence is counted, reads of
    <= 20 bases and lower-case reads are not."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    rng = np.random.default_rng(4)
    recs = []
    for i in range(200):
        seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(15, 80))))
        qual = "".join(rng.choice(list("ACGTNI5"), size=len(seq)))        # some quality lines pass the filter
        if i % 17 == 0:
            seq = seq.lower()
        recs.append(f"@r{i}\n{seq}\n+\n{qual}\n")
    text = "".join(recs).encode()
    fq = tmp_path / "odd.fq"
    fq.write_bytes(text)
    out = str(tmp_path / "cnt")
    k = 21
    subprocess.check_call([host, "counter", "-fastq", str(fq), "-outfile", out, "-kmer", str(k), "-cover", "1"])
    lines = open(os.path.join(out, f"Count_{k}", "part-00000.csv")).read().split("\n")[:-1]
    off, ln = O.fastq_only_seq(text)
    tb = np.frombuffer(text, np.uint8)
    bases = np.concatenate([tb[o:o + l] for o, l in zip(off, ln)])
    roff = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
    assert len(off) > 200 * 0.5 and len(off) != len(O.fastq_group(text)[0])      # the two filters differ here
    keys, counts, _ = O.count_filter(O.extract_canon(bases, roff, k), 1)
    nuc = "ACGT"
    want = ["".join(nuc[(int(x) >> (2 * (k - 1 - j))) & 3] for j in range(k)) + "," + str(int(c))
            for x, c in zip(keys, counts)]
    assert lines == want


def _count_ascii_reads_dev(rfx, torch, reads, L, k, min_cov):
    """uniform-length ASCII reads -> rfx_dev_encode_reads -> fused device count (the ingest path)."""
    n = len(reads)
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = (np.arange(n + 1, dtype=np.int64) * L)
    wpr = (L + 31) // 32
    db = torch.from_numpy(bases.copy()).cuda(); do = torch.from_numpy(off).cuda()
    dw = torch.empty(n * wpr, dtype=torch.int64, device="cuda")
    N = rfx.kmers_per_read(L, k) * n
    dk = torch.empty(max(N, 1), dtype=torch.int64, device="cuda"); dc = torch.empty(max(N, 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rfx.encode_reads_dev(db.data_ptr(), do.data_ptr(), n, wpr, dw.data_ptr())
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
    return bases, off, dk[:m].cpu().numpy().view(np.uint64), dc[:m].cpu().numpy(), nd, inst


@pytest.mark.gpu
@pytest.mark.parametrize("k", [31, 21])
def test_count_survives_extreme_skew(rfx, torch_mod, k):
    """Low-complexity input: every read poly-A / a dinucleotide repeat / one repeated read -- one
    minimiser bucket and one leaf receive (almost) everything, one k-mer occurs millions of times.
    Bucket sizes are exact, so only balance may suffer; the counts must still be the oracle's."""
    L = 150
    rng = np.random.default_rng(k)
    one = "".join(rng.choice(list("ACGT"), size=L))
    reads = ["A" * L] * 12000 + ["AC" * (L // 2)] * 6000 + [one] * 5000 + ["T" * L] * 3000 + \
            ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(2000)] + ["N" * L] * 10
    rng.shuffle(reads)
    bases, off, gk, gc, nd, inst = _count_ascii_reads_dev(rfx, torch_mod, reads, L, k, 2)
    km = O.extract_canon(bases, off, k)
    assert inst == len(km)
    wk, wc, wd = O.count_filter(km, 2)
    assert nd == wd and np.array_equal(gk, wk) and np.array_equal(gc, wc)
    assert int(gc.max()) >= 15000 * (L - k + 1)                      # poly-A and poly-T (and N) are one canonical k-mer


@pytest.mark.gpu
@pytest.mark.parametrize("k,mode", [(31, "sweep"), (25, "sweep"), (31, "overflow"), (31, "skew"), (31, "bits10,4"), (31, "bits7,7"),
                                    (31, "bits5,5")])
def test_one_sweep_level1_matches_oracle(rfx, torch_mod, k, mode, monkeypatch):
    """Level 1 of the record path in ONE sweep (k_sk_onesweep: regions sized from a sampled histogram, extents off
    per-bucket cursors, holes closed by k_fix_holes) forced at a size the default would send through the two-pass
    form: counts equal the oracle's.  'overflow': regions a hundredth of the estimate -- the sweep must notice and
    the two-pass form take over; 'skew': half the reads poly-A, one bucket far beyond its sampled estimate or not,
    either way the oracle's counts."""
    torch = torch_mod
    monkeypatch.setenv("RFX_SK_ONESWEEP", "2")
    if mode == "overflow":
        monkeypatch.setenv("RFX_SK_ONESWEEP_CAP", "1")
    if not mode.startswith("bits"):
        monkeypatch.setenv("RFX_LEVEL_BITS", "9,1")             # (two levels, 512 bins first: a single level -- this size's plan -- takes the two-pass form, and a tile must not put more on a bin than an extent takes)
    if mode.startswith("bits"):
        # 1024 buckets (one workgroup per CU), 128 (extents of 128 or 256), 32 (a tile puts more on a bucket than the largest
        # extent takes: the sweep is not tried)
        monkeypatch.setenv("RFX_LEVEL_BITS", mode[4:])
    seed, G, n_reads, L = 11 + k, 30_000, 40_000, 150
    if mode == "skew":
        rng = np.random.default_rng(3)
        g = O.synth_genome(seed, G)
        bases, off = O.synth_reads(seed, g, G, 0, n_reads // 2, L)
        reads = [bytes(bases[off[i]:off[i + 1]]).decode() for i in range(n_reads // 2)] + ["A" * L] * (n_reads // 2)
        rng.shuffle(reads)
        b2, o2, gk, gc, nd, inst = _count_ascii_reads_dev(rfx, torch, reads, L, k, 2)
        km = O.extract_canon(b2, o2, k)
        wk, wc, wd = O.count_filter(km, 2)
        assert inst == len(km) and nd == wd and np.array_equal(gk, wk) and np.array_equal(gc, wc)
        return
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon(bases, off, k)
    for min_cov in (1, 2):
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
        wk, wc, wd = O.count_filter(km, min_cov)
        assert inst == len(km) and nd == wd and m == len(wk)
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk) and np.array_equal(dc[:m].cpu().numpy(), wc)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [31, 21])
def test_heavy_leaf_slices_merge_exactly(rfx, torch_mod, k, monkeypatch):
    """Force every leaf through the heavy-leaf path (slices counted by the whole grid, partial counts
    sorted and merged, partial buffer grown once) and compare with the oracle."""
    torch = torch_mod
    monkeypatch.setenv("RFX_HEAVY", "40,16,64" if k == 31 else "300,128,64")
    seed, G, n_reads, L = 5 + k, 20_000, 30_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon(bases, off, k)
    for min_cov in (1, 3):
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
        wk, wc, wd = O.count_filter(km, min_cov)
        assert inst == len(km) and nd == wd and m == len(wk)
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk) and np.array_equal(dc[:m].cpu().numpy(), wc)


@pytest.mark.gpu
@pytest.mark.parametrize("k,clips,sweep", [(31, (0, 0), False), (31, (2, 3), False), (28, (0, 0), False), (21, (0, 0), False), (21, (1, 4), False),
                                           # level 1 in ONE sweep with 32 windows per thread, forced at this size: segments that run
                                           # past a read's end, reads without a window in their last segment, clips that shift the stream
                                           (31, (0, 0), True), (31, (2, 3), True), (31, (17, 9), True), (28, (5, 0), True), (25, (0, 6), True), (21, (1, 4), True)])
def test_ragged_reads_device_count(rfx, torch_mod, k, clips, sweep, monkeypatch):
    """Reads of different lengths (trimmed FASTQ) through rfx_dev_encode_reads + rfx_dev_count_reads_ragged:
    reads shorter than k, exactly k + 1, and up to 251 bases, with N; record path (k = 28..31) and k-mer path."""
    torch = torch_mod
    if sweep:
        monkeypatch.setenv("RFX_SK_ONESWEEP", "2")
        monkeypatch.setenv("RFX_LEVEL_BITS", "9,1")             # (two levels, 512 bins first: a single level takes the two-pass form)
        monkeypatch.setenv("RFX_TRACE", "1")
    rng = np.random.default_rng(100 + k)
    genome = "".join(rng.choice(list("ACGT"), size=5000))
    reads = []
    for _ in range(6000):
        L = int(rng.choice([10, k, k + 1, k + 2, 60, 100, 150, 151, 200, 251]))
        p = int(rng.integers(0, len(genome) - L))
        s = list(genome[p:p + L])
        if rng.random() < 0.05:
            s[int(rng.integers(0, L))] = "N"
        reads.append("".join(s))
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    fc, ec = clips
    maxlen = max(len(r) for r in reads)
    wpr = (maxlen + 31) // 32
    n = len(reads)
    db = torch.from_numpy(bases.copy()).cuda(); do = torch.from_numpy(off).cuda()
    dw = torch.empty(n * wpr, dtype=torch.int64, device="cuda")
    dl = torch.empty(n, dtype=torch.int32, device="cuda")
    km = O.extract_canon(bases, off, k, fc, ec)
    dk = torch.empty(len(km) + 1, dtype=torch.int64, device="cuda"); dc = torch.empty(len(km) + 1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rfx.encode_reads_dev(db.data_ptr(), do.data_ptr(), n, wpr, dw.data_ptr(), dl.data_ptr())
    for min_cov in (1, 2):
        m, nd, inst = rfx.count_reads_ragged_dev(dw.data_ptr(), dl.data_ptr(), n, wpr, maxlen, k, dk.data_ptr(),
                                                 dc.data_ptr(), len(km) + 1, min_cov, front_clip=fc, end_clip=ec)
        wk, wc, wd = O.count_filter(km, min_cov)
        assert inst == len(km) and nd == wd and m == len(wk)
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk) and np.array_equal(dc[:m].cpu().numpy(), wc)


@pytest.mark.gpu
@pytest.mark.parametrize("twin", ["rdd", "ds"])
def test_cpp_host_resident_run(tmp_path, ex, planted, twin):
    """`reflexiv_host run --resident`: ONE C-ABI call (rfx_assemble_reads: upload, 2-bit encode, ragged
    count, device-resident extend loop) gives the same contigs as the operator-by-operator driver."""
    import subprocess
    import reflexiv_amd._lib as L
    host = os.path.join(os.path.dirname(L.LIB_PATH), "reflexiv_host")
    fq = str(tmp_path / "ex.fq.gz")
    write_fastq(fq, ex["bases"], ex["read_off"], True)
    out = str(tmp_path / "result")
    subprocess.check_call([host, "run", "--resident", "-fastq", fq, "-outfile", out, "-kmer", "31", "-cover", "3",
                           "--logical-partitions", "4", "--twin", twin])
    assert open(os.path.join(out, "part-00000")).read() == str(ex[f"contigs_{twin}_P4"])
    # planted bubbles / repeat, 100-base reads, min contig 100
    fq2 = str(tmp_path / "pl.fq")
    write_fastq(fq2, planted["bases"], planted["read_off"], False)
    out2 = str(tmp_path / "result2")
    subprocess.check_call([host, "run", "--resident", "-fastq", fq2, "-outfile", out2, "-kmer", "31", "-cover", "2",
                           "-mincontig", "100", "--logical-partitions", "4", "--twin", twin])
    assert open(os.path.join(out2, "part-00000")).read() == str(planted[f"k31_{twin}_contigs"])
    # the multi-GPU form of the same command (`--gpus N`: one host thread + context + RCCL communicator per GPU,
    # rfx_sharded_assemble_reads) forced onto one GPU: same text
    out3 = str(tmp_path / "result3")
    subprocess.check_call([host, "run", "--resident", "--gpus", "1", "-fastq", fq, "-outfile", out3, "-kmer", "31", "-cover", "3",
                           "--logical-partitions", "4", "--twin", twin], env=dict(os.environ, RFX_HOST_FORCE_SHARDED="1"))
    assert open(os.path.join(out3, "part-00000")).read() == str(ex[f"contigs_{twin}_P4"])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [63, 33])
def test_wide_count_survives_extreme_skew(rfx, torch_mod, k):
    """k > 31 fast path under low-complexity input (poly-A, dinucleotide repeats, duplicated reads)."""
    torch = torch_mod
    L = 150
    rng = np.random.default_rng(k)
    one = "".join(rng.choice(list("ACGT"), size=L))
    reads = ["A" * L] * 6000 + ["AC" * (L // 2)] * 3000 + [one] * 3000 + ["T" * L] * 1500 + \
            ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(1500)]
    rng.shuffle(reads)
    n = len(reads)
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.arange(n + 1, dtype=np.int64) * L
    wpr = (L + 31) // 32
    db = torch.from_numpy(bases.copy()).cuda(); do = torch.from_numpy(off).cuda()
    dw = torch.empty(n * wpr, dtype=torch.int64, device="cuda")
    N = rfx.kmers_per_read_w(L, k) * n
    dk = torch.empty(2 * N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.encode_reads_dev(db.data_ptr(), do.data_ptr(), n, wpr, dw.data_ptr())
    m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 2)
    km = O.extract_canon_w(bases, off, k)
    wk, wc, wd = O.count_filter_w(km, k, 2)
    assert inst == len(km) and nd == wd and m == len(wk)
    assert np.array_equal(dk[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2), wk)
    assert np.array_equal(dc[:m].cpu().numpy(), wc)


@pytest.mark.gpu
def test_fused_count_fuzz(rfx, torch_mod):
    """Seeded sweep over read length, k, clips, coverage cut-off, genome size and read count (different
    level plans, ring sizes, descriptor strides, leaf loads): the fused device count equals the oracle."""
    torch = torch_mod
    rng = np.random.default_rng(20261004)
    for case in range(14):
        k = int(rng.choice([31, 31, 30, 29, 28, 27, 21, 15]))
        L = int(rng.choice([k + 1, 64, 100, 150, 151, 250]))
        L = max(L, k + 2)
        G = int(rng.choice([3_000, 40_000, 400_000]))
        G = max(G, L + 1)
        n_reads = int(rng.choice([700, 9_000, 60_000, 200_000]))
        fc, ec = (0, 0) if rng.random() < 0.6 else (int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        min_cov = int(rng.choice([1, 2, 3, 5]))
        seed = 1000 + case
        dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
        N = rfx.kmers_per_read(L, k, fc, ec) * n_reads
        if N <= 0:
            continue
        dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov,
                                          front_clip=fc, end_clip=ec)
        g = O.synth_genome(seed, G)
        bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
        km = O.extract_canon(bases, off, k, fc, ec)
        wk, wc, wd = O.count_filter(km, min_cov)
        tag = f"case {case}: k={k} L={L} G={G} reads={n_reads} clips=({fc},{ec}) cov={min_cov}"
        assert inst == len(km) and nd == wd and m == len(wk), tag
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk), tag
        assert np.array_equal(dc[:m].cpu().numpy(), wc), tag


@pytest.mark.gpu
def test_sharded_count_through_rccl_one_rank(rfx, torch_mod):
    """The multi-GPU count path on real RCCL with a one-rank process group: device tensors through
    all_to_all_single (blocking and async/chunked), exactly the calls `bench.py --gpus N` makes."""
    torch = torch_mod
    import torch.distributed as dist
    from reflexiv_amd import dist as rd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        seed, G, n_reads, L = 23, 200_000, 100_000, 150
        dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
        rfx.use_stream(torch.cuda.current_stream().cuda_stream)
        # k = 31 / 24: super-k-mer records (or pairs) cross the exchange; k = 17: 8-byte k-mers (or pairs)
        for k, forms in ((31, ((1, False), (4, False), (1, True), (3, True), (-4, False), (-16, False))),
                         (24, ((2, False), (2, True), (-8, False))), (17, ((3, False), (2, True), (-4, False)))):
            reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=k)
            N = rfx.kmers_per_read(L, k) * n_reads
            dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 3)
            for chunks, combine in forms:
                eng = rd.HipEngine(rfx, combine=combine)
                eng.force_exchange = True
                gens = -chunks if chunks < 0 else 1            # chunks < 0: generations of the hash space instead
                keys, counts, tot = rd.sharded_count(eng, reads, 3, 10_000_000, 0, chunks=max(1, chunks), generations=gens)
                assert tot == [N, nd, m], (k, chunks, combine)
                # one owner: the shard is everything, but in hash-leaf order -> compare as sorted sets
                o = torch.argsort(keys)
                assert torch.equal(keys[o], dk[:m]) and torch.equal(counts[o], dc[:m]), (k, chunks, combine)
                gk, gc = rd.gather_survivors(keys, counts)
                assert gk is keys or torch.equal(gk, keys)
        # RCCL 2.26 corrupts per-peer messages above 1 GiB; dist._alltoallv caps them (rounds of 512 MiB)
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        n = (1 << 27) + 4099                                      # just over 1 GiB of int64
        x = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device="cuda", generator=gen)
        r, _ = rd._alltoallv(x, [n])
        assert torch.equal(r, x)
        r, works = rd._alltoallv(x, [n], async_op=True)
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        assert torch.equal(r, x)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_degenerate_inputs(rfx, torch_mod):
    """Empty and minimal inputs through every device entry point: no reads, reads shorter than k, one
    window, one read -- nothing to count is an answer, not an error."""
    torch = torch_mod
    one = torch.zeros(8, dtype=torch.int64, device="cuda")
    dk = torch.empty(64, dtype=torch.int64, device="cuda"); dc = torch.empty(64, dtype=torch.int32, device="cuda")
    dc64 = torch.empty(64, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    # no reads
    assert rfx.count_reads_dev(one.data_ptr(), 0, 1, 20, 31, dk.data_ptr(), dc.data_ptr(), 64, 1) == (0, 0, 0)
    assert rfx.count_reads_w_dev(one.data_ptr(), 0, 3, 80, 63, dk.data_ptr(), dc64.data_ptr(), 32, 1) == (0, 0, 0)
    # reads shorter than k (the reference skips them: len - k - endClip <= 1)
    assert rfx.count_reads_dev(one.data_ptr(), 2, 1, 31, 31, dk.data_ptr(), dc.data_ptr(), 64, 1) == (0, 0, 0)
    # one read, a few windows: poly-A
    for L, k in ((40, 31), (33, 31), (35, 21)):
        wpr = (L + 31) // 32
        w = torch.zeros(wpr, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_dev(w.data_ptr(), 1, wpr, L, k, dk.data_ptr(), dc.data_ptr(), 64, 1)
        want = O.extract_canon(np.frombuffer(b"A" * L, np.uint8), np.array([0, L], np.int64), k)
        assert inst == len(want) and (m, nd) == (1, 1) and int(dc[0]) == len(want) and int(dk[0]) == int(want[0])
    # k > 31, one read of exactly k + 2 bases
    L, k = 65, 63
    wpr = (L + 31) // 32
    w = torch.zeros(wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_w_dev(w.data_ptr(), 1, wpr, L, k, dk.data_ptr(), dc64.data_ptr(), 32, 1)
    assert (m, nd, inst) == (1, 1, 3) and int(dc64[0]) == 3
    # host operators on empty inputs
    e8 = np.empty(0, np.uint8); off0 = np.zeros(1, np.int64)
    assert len(rfx.ReverseComplementKmerBinaryExtraction(e8, off0, 31)) == 0
    k_, c_, d_ = rfx.KmerCounting_and_CoverageFilter(np.empty(0, np.uint64), 1)
    assert len(k_) == 0 and d_ == 0
    assert len(rfx.ReverseComplementKmerBinaryExtractionFromDataset64(e8, off0, 63)) == 0


@pytest.mark.gpu
def test_wide_owner_buckets_and_sharded_count(rfx, torch_mod):
    """k = 63 multi-GPU support: every two-word k-mer lands in the bucket of its owner
    (mulhi(wide_hash, n_owners)), the per-owner counts are the global count restricted to the owner, and
    the sharded driver on a one-rank RCCL group equals the fused device count."""
    torch = torch_mod
    import torch.distributed as dist
    from reflexiv_amd import dist as rd
    from tests.test_dist_gloo import wide_hash, mulhi_owner
    seed, G, n_reads, L, k, owners = 17, 60_000, 30_000, 150, 63, 4
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read_w(L, k) * n_reads
    out = torch.empty(2 * N, dtype=torch.int64, device="cuda")
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    h = rfx.bucket_wide_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, out.data_ptr(), N, doff.data_ptr())
    assert h[0] == 0 and h[-1] == N and np.all(np.diff(h) > 0)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon_w(bases, off, k)
    got = out.cpu().numpy().view(np.uint64).reshape(N, 2)
    assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], km[np.lexsort((km[:, 1], km[:, 0]))])
    wk, wc, wd = O.count_filter_w(km, k, 2)
    own_w = mulhi_owner(wide_hash(wk[:, 0], wk[:, 1]), owners)
    tot_d = 0
    for o in range(owners):
        part = got[h[o]:h[o + 1]]
        assert np.all(mulhi_owner(wide_hash(part[:, 0], part[:, 1]), owners) == o)
        n_o = int(h[o + 1] - h[o])
        dk = torch.empty(2 * n_o, dtype=torch.int64, device="cuda"); dc = torch.empty(n_o, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m, d = rfx.count_wide_elems_dev(out[2 * int(h[o]):].data_ptr(), n_o, k, dk.data_ptr(), dc.data_ptr(), n_o, 2)
        tot_d += d
        assert np.array_equal(dk[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2), wk[own_w == o])
        assert np.array_equal(dc[:m].cpu().numpy(), wc[own_w == o])
    assert tot_d == wd
    # the sharded driver through RCCL with one rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rfx.use_stream(torch.cuda.current_stream().cuda_stream)
        reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=k)
        for wide_records, chunks in ((True, 3), (False, 3), (True, 1)):       # 32-byte records / 16-byte k-mers
            eng = rd.HipEngine(rfx, wide_records=wide_records)
            eng.force_exchange = True
            keys, counts, tot = rd.sharded_count(eng, reads, 2, 10_000_000, 0, chunks=chunks)
            assert eng.width == (4 if wide_records else 2)
            assert tot == [N, wd, len(wk)]
            assert np.array_equal(keys.cpu().numpy().view(np.uint64).reshape(-1, 2), wk) and np.array_equal(counts.cpu().numpy(), wc)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("k,owners", [(31, 3), (31, 16), (25, 8), (17, 2)])
def test_combine_bucket_merge_pairs(rfx, torch_mod, k, owners):
    """The exchange after a local combine (reduceByKey's map-side combine): two halves of a read set are
    combined separately into (k-mer, count) pairs, the pairs are bucketed by owner, and every owner's merge of
    both halves' buckets equals the global count / filter restricted to that owner."""
    torch = torch_mod
    from tests.test_dist_gloo import owner_of
    seed, G, n_reads, L, min_cov = 31, 40_000, 24_000, 150, 3
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon(bases, off, k)
    halves, bucketed = [], []
    for a, b in ((0, n_reads // 3), (n_reads // 3, n_reads)):
        nk = rfx.kmers_per_read(L, k) * (b - a)
        cap = nk + (9 << 20)
        scratch = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
        out = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
        doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        # a short buffer reports the need
        with pytest.raises(Exception) as ei:
            rfx.combine_reads_dev(dw[a * wpr:].data_ptr(), b - a, wpr, L, k, owners, scratch.data_ptr(), out.data_ptr(), 16,
                                  doff.data_ptr())
        assert 16 < ei.value.need <= cap
        m, h, inst = rfx.combine_reads_dev(dw[a * wpr:].data_ptr(), b - a, wpr, L, k, owners, scratch.data_ptr(),
                                           out.data_ptr(), cap, doff.data_ptr())
        assert inst == nk
        wk, wc, wd = O.count_filter(km[a * (L - k + 1): b * (L - k + 1)], 1)
        assert m == wd and h[0] == 0 and h[-1] == m and np.array_equal(doff.cpu().numpy(), h)
        bp = out[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)
        for o in range(owners):
            assert np.all(owner_of(bp[h[o]:h[o + 1], 0], owners) == o)
        order = np.argsort(bp[:, 0], kind="stable")
        assert np.array_equal(bp[order, 0], wk) and np.array_equal(bp[order, 1], wc.astype(np.uint64))
        # the grouping step alone, on pairs with holes (count 0) in between
        holes = torch.zeros(4 * m + 2, dtype=torch.int64, device="cuda")
        holes[2:2 + 4 * m].view(-1, 4)[:, :2] = out[:2 * m].view(-1, 2)
        out2 = torch.empty(2 * m, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        h2 = rfx.bucket_pairs_by_owner_dev(holes.data_ptr(), 2 * m + 1, owners, out2.data_ptr(), doff.data_ptr())
        assert np.array_equal(h2, h)
        bp2 = out2.cpu().numpy().view(np.uint64).reshape(m, 2)
        assert np.array_equal(bp2[np.argsort(bp2[:, 0], kind="stable")], bp[order])
        bucketed.append((out, h))
    gk, gc, gd = O.count_filter(km, min_cov)
    own = owner_of(gk, owners)
    own_all = owner_of(np.unique(km), owners)
    for o in range(owners):
        recv = torch.cat([out[2 * int(h[o]):2 * int(h[o + 1])] for out, h in bucketed])
        n = int(recv.numel()) // 2
        dk = torch.empty(max(1, n), dtype=torch.int64, device="cuda"); dc = torch.empty(max(1, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, d = rfx.merge_pairs_dev(recv.data_ptr(), n, k, dk.data_ptr(), dc.data_ptr(), max(1, n), min_cov)
        assert d == int(np.sum(own_all == o))
        assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), gk[own == o])
        assert np.array_equal(dc[:m].cpu().numpy(), gc[own == o])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [33, 40, 47, 63])
def test_wide_record_path_equals_oracle(rfx, torch_mod, k, monkeypatch):
    """RFX_WIDE_RECORDS=1: k = 33..63 through 32-byte super-k-mer records (minimiser of the central 31/30
    bases, record levels, expanding leaves) -- same counts as the oracle and as the default element path."""
    torch = torch_mod
    seed, G, n_reads, L = 77, 30_000, 20_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read_w(L, k) * n_reads
    res = []
    for flag in ("0", "1", "sweep"):
        # "sweep": the record path with its level 1 in one sweep (k_sk_onesweep<W, true>), forced at this size
        monkeypatch.setenv("RFX_WIDE_RECORDS", "0" if flag == "0" else "1")
        monkeypatch.setenv("RFX_SK_ONESWEEP", "2" if flag == "sweep" else "0")
        if flag == "sweep":
            monkeypatch.setenv("RFX_LEVEL_BITS", "9,1")        # (two levels: this size's single level takes the two-pass form)
        dk = torch.empty(2 * N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m, d, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 2)
        res.append((m, d, dk[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2), dc[:m].cpu().numpy()))
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 2)
    for m, d, kk, cc in res:
        assert (m, d) == (len(wk), wd)
        assert np.array_equal(kk, wk) and np.array_equal(cc, wc)


@pytest.mark.gpu
def test_combine_and_merge_edge_cases(rfx, torch_mod):
    """empty read set, reads too short for a k-mer, no pairs to group or merge, and a merge whose partial
    counts of one k-mer arrive in many pieces (sums, then the coverage filter on the SUM)."""
    torch = torch_mod
    k, L = 31, 100
    wpr = (L + 31) // 32
    cap = 9 << 20
    scratch = torch.empty(2 * cap, dtype=torch.int64, device="cuda"); out = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    doff = torch.empty(5, dtype=torch.int64, device="cuda")
    dw = torch.zeros(4 * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, h, inst = rfx.combine_reads_dev(dw.data_ptr(), 0, wpr, L, k, 4, scratch.data_ptr(), out.data_ptr(), cap, doff.data_ptr())
    assert (m, inst) == (0, 0) and not h.any()
    m, h, inst = rfx.combine_reads_dev(dw.data_ptr(), 4, wpr, 20, k, 4, scratch.data_ptr(), out.data_ptr(), cap, doff.data_ptr())
    assert (m, inst) == (0, 0) and not h.any()                       # 20-base reads hold no 31-mer
    # four poly-A reads: one k-mer, count 4 * 70
    m, h, inst = rfx.combine_reads_dev(dw.data_ptr(), 4, wpr, L, k, 4, scratch.data_ptr(), out.data_ptr(), cap, doff.data_ptr())
    assert (m, inst) == (1, 4 * (L - k + 1))
    assert out[:2].cpu().tolist() == [0, 4 * (L - k + 1)]
    h2 = rfx.bucket_pairs_by_owner_dev(out.data_ptr(), 0, 4, scratch.data_ptr(), doff.data_ptr())
    assert not h2.any()
    dk = torch.empty(16, dtype=torch.int64, device="cuda"); dc = torch.empty(16, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert rfx.merge_pairs_dev(out.data_ptr(), 0, k, dk.data_ptr(), dc.data_ptr(), 16, 2) == (0, 0)
    # 3 k-mers in 40 000 pieces: counts 1 each -> sums 20000, 19999, 1; the filter sees the sums
    n = 40_000
    keys = torch.tensor([5, 9], dtype=torch.int64).repeat(n // 2); keys[-1] = 77
    pairs = torch.stack([keys, torch.ones(n, dtype=torch.int64)], dim=1).reshape(-1).cuda()
    torch.cuda.synchronize()
    m, d = rfx.merge_pairs_dev(pairs.data_ptr(), n, k, dk.data_ptr(), dc.data_ptr(), 16, 2, 19_999)
    assert (m, d) == (1, 3) and dk[:1].cpu().tolist() == [9] and dc[:1].cpu().tolist() == [19_999]
    m, d = rfx.merge_pairs_dev(pairs.data_ptr(), n, k, dk.data_ptr(), dc.data_ptr(), 16, 1)
    assert (m, d) == (3, 3) and dk[:3].cpu().tolist() == [5, 9, 77] and dc[:3].cpu().tolist() == [20_000, 19_999, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [31, 17])
def test_combine_and_merge_through_heavy_leaves(rfx, torch_mod, k, monkeypatch):
    """The pair forms with every leaf forced through the heavy-leaf path (slices, partial counts sorted and
    reduced): the combine emits pairs from k_reduce_partials, the merge sums weighted partials."""
    torch = torch_mod
    monkeypatch.setenv("RFX_HEAVY", "40,16,64" if k == 31 else "300,128,64")
    seed, G, n_reads, L, owners = 11 + k, 20_000, 30_000, 150, 2
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    N = rfx.kmers_per_read(L, k) * n_reads
    cap = N + (9 << 20)
    scratch = torch.empty(2 * cap, dtype=torch.int64, device="cuda"); out = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, h, inst = rfx.combine_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, scratch.data_ptr(), out.data_ptr(), cap, doff.data_ptr())
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    km = O.extract_canon(bases, off, k)
    wk, wc, wd = O.count_filter(km, 1)
    bp = out[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)
    order = np.argsort(bp[:, 0], kind="stable")
    assert m == wd and np.array_equal(bp[order, 0], wk) and np.array_equal(bp[order, 1], wc.astype(np.uint64))
    # merge everything (both owners' buckets, twice over: every partial count doubled)
    twice = torch.cat([out[:2 * m], out[:2 * m]])
    dk = torch.empty(2 * m, dtype=torch.int64, device="cuda"); dc = torch.empty(2 * m, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    mm, d = rfx.merge_pairs_dev(twice.data_ptr(), 2 * m, k, dk.data_ptr(), dc.data_ptr(), 2 * m, 6)
    keep = 2 * wc >= 6
    assert d == wd and mm == int(keep.sum())
    assert np.array_equal(dk[:mm].cpu().numpy().view(np.uint64), wk[keep]) and np.array_equal(dc[:mm].cpu().numpy(), 2 * wc[keep])


@pytest.mark.gpu
@pytest.mark.parametrize("k,owners", [(63, 4), (63, 1), (47, 64), (31, 4), (29, 1)])
def test_records_do_not_depend_on_the_run_descriptors(rfx, torch_mod, k, owners, monkeypatch):
    """level 1 of the record paths: the scatter that takes its runs from the histogram pass's descriptors
    (the default) and the one that recomputes them (RFX_SK_DESC=0) write the same MULTISET of records, bit for
    bit, every time (round 4: a 24-byte record layout passed every count test most of the time and failed this
    one always -- tools/attempts/)."""
    torch = torch_mod
    seed, G, n_reads, L = 31 + k, 30_000, 4000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    wide = k > 32
    fn = rfx.bucket_wide_records_by_owner_dev if wide else rfx.bucket_records_by_owner_dev
    width = 4 if wide else 2

    def records():
        need, h = fn(dw.data_ptr(), n_reads, wpr, L, k, owners, 0, 0, doff.data_ptr())
        out = torch.full((width * need,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        nrec, h = fn(dw.data_ptr(), n_reads, wpr, L, k, owners, out.data_ptr(), need, doff.data_ptr())
        assert nrec == need
        r = out.cpu().numpy().view(np.uint64).reshape(nrec, width)
        return r[np.lexsort(tuple(r[:, i] for i in reversed(range(width))))], h

    monkeypatch.setenv("RFX_SK_DESC", "0")
    ref, href = records()
    assert not (ref == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1).any()        # every slot was written
    for mode in ("0", "1", "1", "1"):
        monkeypatch.setenv("RFX_SK_DESC", mode)
        got, h = records()
        assert np.array_equal(h, href) and np.array_equal(got, ref), mode


@pytest.mark.gpu
@pytest.mark.parametrize("k,bits", [(31, "6,5"), (63, "6,5"), (33, "4,4")])
def test_first_level_of_received_records_without_a_histogram_under_skew(rfx, torch_mod, k, bits, monkeypatch):
    """the claim form of the receiver's first level (forced at this size) on low-complexity input -- poly-A, a dinucleotide
    repeat, one read many times: one bucket takes most of the records, every tile claims from the same cursor -- against the
    exact form and the oracle."""
    torch = torch_mod
    L = 150
    rng = np.random.default_rng(100 + k)
    one = "".join(rng.choice(list("ACGT"), size=L))
    reads = ["A" * L] * 9000 + ["AC" * (L // 2)] * 4000 + [one] * 4000 + ["T" * L] * 2000 + \
            ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(3000)]
    rng.shuffle(reads)
    n = len(reads)
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    off = np.arange(n + 1, dtype=np.int64) * L
    wpr = (L + 31) // 32
    db = torch.from_numpy(bases.copy()).cuda(); do = torch.from_numpy(off).cuda()
    dw = torch.empty(n * wpr, dtype=torch.int64, device="cuda")
    doff = torch.empty(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.encode_reads_dev(db.data_ptr(), do.data_ptr(), n, wpr, dw.data_ptr())
    wide = k > 32
    fn = rfx.bucket_wide_records_by_owner_dev if wide else rfx.bucket_records_by_owner_dev
    width = 4 if wide else 2
    need, _ = fn(dw.data_ptr(), n, wpr, L, k, 1, 0, 0, doff.data_ptr())
    out = torch.empty(width * need, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nrec, h = fn(dw.data_ptr(), n, wpr, L, k, 1, out.data_ptr(), need, doff.data_ptr())
    N = (rfx.kmers_per_read_w(L, k) if wide else L - k + 1) * n
    W = 2 if wide else 1
    if wide:
        wk, wc, wd = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 2)
    else:
        wk, wc, wd = O.count_filter(O.extract_canon(bases, off, k), 2)
    for mode in ("0", "2"):
        monkeypatch.setenv("RFX_REC_ONESWEEP", mode)
        monkeypatch.setenv("RFX_LEVEL_BITS", bits)
        dk = torch.empty(W * N, dtype=torch.int64, device="cuda")
        dc = torch.empty(N, dtype=torch.int64 if wide else torch.int32, device="cuda")
        torch.cuda.synchronize()
        if wide:
            m, d = rfx.count_wide_records_dev(out.data_ptr(), nrec, 0, k, dk.data_ptr(), dc.data_ptr(), N, 2)
        else:
            m, d = rfx.count_records_dev(out.data_ptr(), nrec, N, k, dk.data_ptr(), dc.data_ptr(), N, 2)
        assert (m, d) == (len(wk), wd), mode
        assert np.array_equal(dk[:W * m].cpu().numpy().view(np.uint64).reshape(m, W), np.asarray(wk).view(np.uint64).reshape(len(wk), W))
        assert np.array_equal(dc[:m].cpu().numpy().astype(np.int64), np.asarray(wc).astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [31, 63])
def test_children_that_outgrow_their_regions_spill_and_are_moved(rfx, torch_mod, k, monkeypatch):
    """the last radix level in one sweep (k_rec_l2sweep) sizes every child's region from a 1/16 sample; a child that outgrows
    it spills into a list and k_l2_spill_fix moves it, region and spill, behind the regions.  RFX_L2_SQUEEZE shrinks the
    regions: at 100 % (nothing or little spills), from 84 % down (children spill and are moved), 30 % (more than the list takes: the
    exact form runs after all) the survivors are those of the exact form, key for key."""
    torch = torch_mod
    seed, G, n_reads, L = 3 + k, 1_500_000, 2_400_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    wide = k > 32
    N = (rfx.kmers_per_read_w(L, k) if wide else L - k + 1) * n_reads
    cap = N // 8
    W = 2 if wide else 1
    monkeypatch.setenv("RFX_LEVEL_BITS", "8,6")
    res = []
    stats = []
    for sweep, squeeze in [("0", "100"), ("1", "100")] + [("1", str(q)) for q in range(84, 59, -3)] + [("1", "30")]:
        monkeypatch.setenv("RFX_L2_ONESWEEP", sweep)
        monkeypatch.setenv("RFX_L2_SQUEEZE", squeeze)
        dk = torch.empty(W * cap, dtype=torch.int64, device="cuda")
        dc = torch.empty(cap, dtype=torch.int64 if wide else torch.int32, device="cuda")
        torch.cuda.synchronize()
        if wide:
            m, d, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
        else:
            m, d, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
        h = hashlib.sha256(dk[:W * m].cpu().numpy().tobytes() + dc[:m].cpu().numpy().tobytes()).hexdigest()
        res.append((m, d, inst, h))
        t = rfx.count_timing()
        stats.append((t.get("stat_l2_spilled", (0, 0))[1], t.get("stat_l2_void", (0, 0))[1]))
    assert res[0][2] == N and res[0][0] > 1_000_000
    assert all(r == res[0] for r in res), res
    assert stats[0] == (0, 0) and stats[1][1] == 0, stats
    assert any(sp > 0 and void == 0 for sp, void in stats), stats  # somewhere on the way down children spilled and were moved; the sweep stood
    assert stats[-1][1] == 1, stats                                # too many: the exact form after all


@pytest.mark.gpu
@pytest.mark.parametrize("k,bits", [(31, "6,5"), (31, "9,4"), (29, "4,6"), (63, "6,5"), (47, "8,4"), (63, "5,3,4")])
def test_first_level_of_received_records_in_one_sweep(rfx, torch_mod, k, bits, monkeypatch):
    """what a rank receives from the exchange is counted with its FIRST level without a histogram pass (k_rec_claim_scatter:
    regions from a sample, a contiguous range claimed per tile and digit) -- forced at this size -- and gives the counts of the
    exact form and of the oracle; both
    record kinds, first levels of 4..9 bits, two and three levels."""
    torch = torch_mod
    seed, G, n_reads, L = 7 + k, 40_000, 30_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    doff = torch.empty(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    wide = k > 32
    fn = rfx.bucket_wide_records_by_owner_dev if wide else rfx.bucket_records_by_owner_dev
    width = 4 if wide else 2
    need, _ = fn(dw.data_ptr(), n_reads, wpr, L, k, 1, 0, 0, doff.data_ptr())
    out = torch.empty(width * need, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nrec, h = fn(dw.data_ptr(), n_reads, wpr, L, k, 1, out.data_ptr(), need, doff.data_ptr())
    N = (rfx.kmers_per_read_w(L, k) if wide else L - k + 1) * n_reads
    cap = N // 2
    W = 2 if wide else 1
    res = []
    for mode in ("0", "2"):
        monkeypatch.setenv("RFX_REC_ONESWEEP", mode)
        monkeypatch.setenv("RFX_LEVEL_BITS", bits)
        dk = torch.empty(W * cap, dtype=torch.int64, device="cuda")
        dc = torch.empty(cap, dtype=torch.int64 if wide else torch.int32, device="cuda")
        torch.cuda.synchronize()
        if wide:
            m, d = rfx.count_wide_records_dev(out.data_ptr(), nrec, 0, k, dk.data_ptr(), dc.data_ptr(), cap, 2)
        else:
            m, d = rfx.count_records_dev(out.data_ptr(), nrec, N, k, dk.data_ptr(), dc.data_ptr(), cap, 2)
        res.append((m, d, dk[:W * m].cpu().numpy().view(np.uint64).reshape(m, W), dc[:m].cpu().numpy().astype(np.int64)))
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    if wide:
        wk, wc, wd = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 2)
    else:
        wk, wc, wd = O.count_filter(O.extract_canon(bases, off, k), 2)
        wk = wk.reshape(-1, 1)
    for m, d, kk, cc in res:
        assert (m, d) == (len(wk), wd)
        assert np.array_equal(kk, wk.view(np.uint64).reshape(len(wk), W)) and np.array_equal(cc, wc.astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("k,owners", [(63, 4), (40, 3), (33, 8)])
def test_wide_record_owner_buckets(rfx, torch_mod, k, owners):
    """k = 33..63 multi-GPU support, record form: the 32-byte records grouped by owner, a too-small buffer
    reports the need, and the per-owner counts of the records are disjoint and add up to the global count."""
    torch = torch_mod
    seed, G, n_reads, L = 19 + k, 50_000, 24_000, 150
    dg, dw, wpr = make_reads_dev(rfx, torch, seed, G, n_reads, L)
    doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    need, h = rfx.bucket_wide_records_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, 0, 0, doff.data_ptr())
    assert h is None and need > n_reads
    out = torch.empty(4 * need, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nrec, h = rfx.bucket_wide_records_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, out.data_ptr(), need, doff.data_ptr())
    assert nrec == need and h[0] == 0 and h[-1] == nrec and np.all(np.diff(h) > 0)
    recs = out.cpu().numpy().view(np.uint64).reshape(nrec, 4)
    nwin = ((recs[:, 3] >> np.uint64(32)) & np.uint64(15)).astype(np.int64) + 1
    N = rfx.kmers_per_read_w(L, k) * n_reads
    assert int(nwin.sum()) == N                                   # every window is in exactly one record
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L)
    wk, wc, wd = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 2)
    got_k, got_c, tot_d = [], [], 0
    for o in range(owners):
        n_o = int(h[o + 1] - h[o])
        cap = 6 * n_o + 16
        dk = torch.empty(2 * cap, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m, d = rfx.count_wide_records_dev(out[4 * int(h[o]):].data_ptr(), n_o, 0, k, dk.data_ptr(), dc.data_ptr(), cap, 2)
        tot_d += d
        kk = dk[:2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)
        assert m == 0 or np.all((kk[1:, 0] > kk[:-1, 0]) | ((kk[1:, 0] == kk[:-1, 0]) & (kk[1:, 1] > kk[:-1, 1])))
        got_k.append(kk); got_c.append(dc[:m].cpu().numpy())
    allk = np.concatenate(got_k); allc = np.concatenate(got_c)
    order = np.lexsort((allk[:, 1], allk[:, 0]))
    assert tot_d == wd and np.array_equal(allk[order], wk) and np.array_equal(allc[order], wc)


@pytest.mark.parametrize("k,env", [
    (31, {"RFX_LEAF_TARGET": "131072", "RFX_PRESPLIT": "1500"}),      # leaves 8 x the default size, tiny pre-split threshold
    (31, {"RFX_LEAF_TARGET": "131072", "RFX_PRESPLIT": "1000000"}),   # no pre-split: every full leaf finds out by overflowing
    (29, {"RFX_LEAF_TARGET": "65536", "RFX_PRESPLIT": "700"}),
    (63, {"RFX_LEAF_TARGET": "131072", "RFX_WIDE_PRESPLIT": "300"}),
    (63, {"RFX_LEAF_TARGET": "131072", "RFX_WIDE_PRESPLIT": "1000000"}),
    (47, {"RFX_LEAF_TARGET": "65536", "RFX_WIDE_PRESPLIT": "150"}),
])
def test_leaf_tables_under_forced_split_passes(rfx, torch_mod, k, env):
    """Many hash-selected table passes per workgroup (the `0081c2e` race -- the adaptive pre-split
    threshold moved after the closing barrier -- showed only in a variant build).  Leaves several times the default size
    on a distinct-heavy read set (shallow coverage, 2 % substitutions) overflow their 4096-slot tables: with a tiny
    pre-split threshold every leaf starts in many parts and the threshold adapts on every overflow; with a huge one every
    full leaf is abandoned and re-streamed in 2, 4, .. parts.  Counts and survivors against the oracle, and the
    statistics show that the machinery really ran."""
    import os
    torch = torch_mod
    seed, G, n_reads, L, min_cov = 4242 + k, 3_000_000, 400_000, 150, 2
    wpr = (L + 31) // 32
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    err = int(0.02 * (1 << 32))
    rfx.synth_genome_dev(seed, G, dg.data_ptr())
    rfx.synth_reads_dev(seed, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr(), err)
    rfx.sync()
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, n_reads, L, err)
    old = {key: os.environ.get(key) for key in env}
    os.environ.update(env)
    try:
        if k <= 31:
            N = rfx.kmers_per_read(L, k) * n_reads
            dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, min_cov)
            stats = rfx.count_timing()
            O.set_threads(O.host_cores())
            wk, wc, wd, wi = O.count_reads_omp(bases, off, k, min_cov)
            assert (inst, nd, m) == (wi, wd, len(wk))
            assert np.array_equal(dk[:m].cpu().numpy().view(np.uint64), wk)
            assert np.array_equal(dc[:m].cpu().numpy(), wc)
        else:
            W = k // 32 + 1
            N = rfx.kmers_per_read_w(L, k) * n_reads
            cap = N // 4
            dk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, min_cov)
            stats = rfx.count_timing()
            O.set_threads(O.host_cores())
            wk, wc, wd, wi = O.count_reads_omp(bases, off, k, min_cov)
            assert (inst, nd, m) == (wi, wd, len(wc)) and wi == N
            assert np.array_equal(dk[:m * W].cpu().numpy().view(np.uint64).reshape(m, W), wk)
            assert np.array_equal(dc[:m].cpu().numpy(), np.asarray(wc, np.int64))
    finally:
        for key, v in old.items():
            if v is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = v
    leaves, passes, over = (stats[s][1] for s in ("stat_leaves", "stat_passes", "stat_overflows"))
    assert leaves > 0 and passes >= 3 * leaves, stats                 # several table passes per leaf on average
    if int(env.get("RFX_PRESPLIT", env.get("RFX_WIDE_PRESPLIT"))) >= 1000000:
        assert over >= leaves // 2, stats                             # and the overflow path when nothing is pre-split
