"""The multi-GPU count stage behind the C ABI (rfx_comm_*, rfx_dev_sharded_count, rfx_dev_gather_shards: RCCL bound inside
libreflexiv_hip.so) on a ONE-rank communicator -- the exchange, the generations pipeline, the 512 MiB rounds and the
rank's own bucket through ncclSend / ncclRecv all run on the GPU with real RCCL; results against the fused one-GPU count
(itself pinned to the oracle) and, at the per-GPU share of BASELINE configs 3 and 4 (6.25 Gbp, -cover 38), against the
oracle's pin tests/golden/c3_share.json (made by tests/golden/make_c2_full.py --reads 41666668 --cover 38)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = os.path.join(HERE, "golden", "c3_share.json")


class _Shared:
    """ONE context + one-rank communicator for the whole module (the RFX_COMM_* / RFX_SK_* knobs are read per call): RCCL
    does not take kindly to many communicators made and destroyed in one process."""
    rfx = None


@pytest.fixture(scope="module", autouse=True)
def _close_shared():
    yield
    if _Shared.rfx is not None:
        _Shared.rfx.close()
        _Shared.rfx = None


class _Knobs:
    def __init__(self, rfx, env):
        self.rfx, self.env, self.old = rfx, env, {}

    def __getattr__(self, name):
        return getattr(self.rfx, name)

    def close(self):                                            # (the tests' `finally: rfx.close()`: put the knobs back)
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make_comm(env):
    import reflexiv_amd
    if _Shared.rfx is None:
        _Shared.rfx = reflexiv_amd.Reflexiv()
        _Shared.rfx.comm_init(reflexiv_amd.Reflexiv.comm_unique_id(), 0, 1)
    h = _Knobs(_Shared.rfx, env)
    h.old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    return h


def reads_on_device(rfx, seed, G, n_reads, L, err=21474836):
    import torch
    wpr = (L + 31) // 32
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(seed, G, dg.data_ptr())
    rfx.synth_reads_dev(seed, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr(), err)
    rfx.sync()
    return dw, wpr


@pytest.mark.parametrize("env", [{}, {"RFX_COMM_SELF_VIA_RCCL": "1"},
                                 {"RFX_COMM_SELF_VIA_RCCL": "1", "RFX_COMM_LIMIT_BYTES": "1048576"},
                                 # the sender's bucketing by level 1's one sweep (an owner's bucket in pieces), forced at this size
                                 {"RFX_COMM_SELF_VIA_RCCL": "1", "RFX_SK_ONESWEEP": "2", "RFX_COMM_LIMIT_BYTES": "262144"},
                                 {"RFX_SK_ONESWEEP": "2", "RFX_COMM_VIRTUAL_WORLD": "8"},
                                 {"RFX_SK_ONESWEEP": "2", "RFX_COMM_VIRTUAL_WORLD": "3"}])
def test_sharded_count_behind_the_c_abi_one_rank(env):
    import torch
    rfx = make_comm(env)
    try:
        seed, G, n_reads, L = 31, 300_000, 150_000, 150
        dw, wpr = reads_on_device(rfx, seed, G, n_reads, L)
        for k, gens in ((31, 1), (31, 4), (25, 3), (63, 1), (63, 4), (40, 2)):
            wide = k > 32
            W = 2 if wide else 1
            nk = (rfx.kmers_per_read_w if wide else rfx.kmers_per_read)(L, k)
            N = nk * n_reads
            cap = N // 2
            cdt = torch.int64 if wide else torch.int32
            dk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=cdt, device="cuda")
            sk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); sc = torch.empty(cap, dtype=cdt, device="cuda")
            torch.cuda.synchronize()
            if wide:
                m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
            else:
                m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
            ms, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, sk.data_ptr(), sc.data_ptr(), cap, 3,
                                            generations=gens)
            assert (ms, tot) == (m, [inst, nd, m]), (k, gens)
            assert torch.equal(sk[:m * W], dk[:m * W]) and torch.equal(sc[:m], dc[:m]), (k, gens)
            assert rfx.comm_bytes_bucketed() > 0
            # a capacity that is too small is reported with the need, and the call can be repeated
            import reflexiv_amd
            with pytest.raises(reflexiv_amd.RfxError) as ei:
                rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, sk.data_ptr(), sc.data_ptr(), m // 2, 3, generations=gens)
            assert ei.value.need == m
            # gather to root (one rank: the shard itself), count() of the stop rule
            gk = torch.empty_like(sk); gc = torch.empty_like(sc)
            torch.cuda.synchronize()
            assert rfx.gather_shards_dev(sk.data_ptr(), sc.data_ptr(), m, W, 8 if wide else 4, 0, gk.data_ptr(), gc.data_ptr(), cap) == m
            assert torch.equal(gk[:m * W], sk[:m * W]) and torch.equal(gc[:m], sc[:m])
        assert rfx.comm_all_reduce([5, -7, 1 << 40]) == [5, -7, 1 << 40]
        assert rfx.comm_all_reduce([3, 9], "max") == [3, 9]
    finally:
        rfx.close()


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/c3_share.json not generated")
@pytest.mark.parametrize("k,env", [(31, {"RFX_COMM_SELF_VIA_RCCL": "1"}), (63, {"RFX_COMM_SELF_VIA_RCCL": "1"}),
                                   # as a rank of 8 (what `bench.py --force-dist` times): 32 (generation, owner) bins, the sender's
                                   # bucketing by level 1's one sweep with 16 sub-bins per bucket and the gaps closed
                                   (31, {"RFX_COMM_VIRTUAL_WORLD": "8"}), (63, {"RFX_COMM_VIRTUAL_WORLD": "8"})])
def test_multi_gpu_branch_at_the_config3_share_matches_the_oracle_pin(k, env):
    """The code path `bench.py --gpus 8` runs on every rank (records in 4 generations through RCCL, count per generation,
    merge), rehearsed on one rank at the per-GPU share of configs 3 / 4: 41,666,668 reads, -cover 38.  Counts, sha256 of
    the survivors, the extend trace and sha256 of the contig text against the ORACLE's pin."""
    import torch
    import reflexiv_amd
    pin = json.load(open(PIN))
    rec, w = pin[f"k{k}"], pin["workload"]
    rfx = make_comm(env)
    try:
        n_reads, L, cover, P = w["reads"], w["read_len"], w["cover"], w["partitions"]
        dw, wpr = reads_on_device(rfx, w["seed"], w["genome"], n_reads, L, w["err_per_2_32"])
        wide = k > 32
        W = 2 if wide else 1
        cap = 1 << 24
        dk = torch.empty(cap * W, dtype=torch.int64, device="cuda")
        dc = torch.empty(cap, dtype=torch.int64 if wide else torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, cover, generations=4)
        assert tot == [rec["n_instances"], rec["n_distinct"], rec["n_kept"]] and m == rec["n_kept"]
        keys = dk[:m * W].cpu().numpy().view(np.uint64)
        counts = dc[:m].cpu().numpy()
        assert hashlib.sha256(keys.tobytes()).hexdigest() == rec["sha256_keys"]
        assert hashlib.sha256(counts.tobytes()).hexdigest() == rec["sha256_counts"]
        prm = reflexiv_amd.default_params(k=k, min_cov=cover, partitions=P)
        if wide:
            aw = (k - 1) // 31 + 1
            ak = torch.empty(m * aw, dtype=torch.int64, device="cuda"); ac = torch.empty(m, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            m2 = rfx.counter_to_asm_dev(dk.data_ptr(), dc.data_ptr(), m, k, ak.data_ptr(), ac.data_ptr(), cover)
            text, nc, trace = rfx.assemble_w_dev(ak.data_ptr(), ac.data_ptr(), m2, prm)
        else:
            text, nc, trace = rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), m, prm)
        assert trace == rec["trace"] and nc == rec["n_contigs"]
        assert hashlib.sha256(text.encode()).hexdigest() == rec["sha256_contig_text"]
    finally:
        rfx.close()


PIN_C5 = os.path.join(HERE, "golden", "c5_share.json")


@pytest.mark.skipif(not os.path.exists(PIN_C5), reason="tests/golden/c5_share.json not generated")
def test_config5_real_per_gpu_share_as_a_rank_of_8_matches_the_oracle_pin():
    """BASELINE config 5's REAL per-GPU share -- 18.75 Gbp = 125,000,000 PE150 reads of a 400 Mbp genome (150 Gbp at 48x over 8
    GPUs), -cover 2, 1.5e10 k-mer instances, 2.5e9 distinct, 4.5e8 kept -- through rfx_dev_sharded_count as a rank of 8 (32
    (generation, owner) bins, records in 4 generations through RCCL), against the ORACLE's count-stage pin
    (tests/golden/make_c2_full.py --reads 125000000 --genome 400000000 --cover 2 --ks 31 --passes 16 --count-only; the oracle's
    extend stage does not hold 9e8 records in the build container's memory, so the pin stops at the survivors)."""
    import torch
    pin = json.load(open(PIN_C5))
    rec, w = pin["k31"], pin["workload"]
    rfx = make_comm({"RFX_COMM_SELF_VIA_RCCL": "1", "RFX_COMM_VIRTUAL_WORLD": "8"})
    try:
        n_reads, L, cover = w["reads"], w["read_len"], w["cover"]
        dw, wpr = reads_on_device(rfx, w["seed"], w["genome"], n_reads, L, w["err_per_2_32"])
        cap = int(rec["n_kept"] * 1.02) + 4096
        dk = torch.empty(cap, dtype=torch.int64, device="cuda")
        dc = torch.empty(cap, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, 31, dk.data_ptr(), dc.data_ptr(), cap, cover, generations=4)
        assert tot == [rec["n_instances"], rec["n_distinct"], rec["n_kept"]] and m == rec["n_kept"]
        hk, hc = hashlib.sha256(), hashlib.sha256()
        step = 1 << 26                                          # (the survivors come down in pieces: 5.4 GB in all)
        for a in range(0, m, step):
            hk.update(dk[a:min(m, a + step)].cpu().numpy().view(np.uint64).tobytes())
            hc.update(dc[a:min(m, a + step)].cpu().numpy().tobytes())
        assert hk.hexdigest() == rec["sha256_keys"] and hc.hexdigest() == rec["sha256_counts"]
        del dw, dk, dc
        rfx.trim()                                              # (35 GB of send / count workspace back before the next test)
    finally:
        rfx.close()


def test_sharded_assemble_reads_one_rank_matches_the_documented_example(golden_dir):
    """rfx_sharded_assemble_reads (host ASCII reads -> encode -> sharded count over RCCL -> gather -> driver) on a one-rank
    communicator: the documented example (k = 31, -cover 3, P = 4, RDD twin) gives `>Contig-4558-0` and the oracle's text;
    a ragged read set (lengths 40..150) gives what the oracle gives."""
    import reflexiv_amd
    from oracle import oracle as O
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    rfx = make_comm({"RFX_COMM_SELF_VIA_RCCL": "1"})
    try:
        prm = reflexiv_amd.default_params(min_cov=3, partitions=4, twin=reflexiv_amd.TWIN_RDD)
        text, nc, trace, tot = rfx.sharded_assemble_reads(ex["bases"], ex["read_off"], prm, generations=4)
        km = O.extract_canon(ex["bases"], ex["read_off"], 31)
        wk, wc, wd = O.count_filter(km, 3)
        otext, onc, otrace, _ = O.assemble_from_counts(wk, wc, O.default_params(min_cov=3, partitions=4, twin=O.TWIN_RDD))
        assert tot == [len(km), wd, len(wk)]
        assert (text, nc, trace) == (otext, onc, otrace) and text.startswith(">Contig-4558-0\n")
        # ragged reads
        rng = np.random.default_rng(8)
        g = O.synth_genome(5, 60_000)
        bases, off = O.synth_reads(5, g, 60_000, 0, 30_000, 150)
        lens = rng.integers(40, 151, 30_000)
        keep = np.concatenate([np.arange(off[i], off[i] + lens[i]) for i in range(len(lens))])
        rb = bases[keep]
        roff = np.zeros(len(lens) + 1, np.int64); roff[1:] = np.cumsum(lens)
        prm = reflexiv_amd.default_params(min_cov=2, partitions=3, min_contig=200)
        text, nc, trace, tot = rfx.sharded_assemble_reads(rb, roff, prm, generations=2)
        km = O.extract_canon(rb, roff, 31)
        wk, wc, wd = O.count_filter(km, 2)
        otext, onc, otrace, _ = O.assemble_from_counts(wk, wc, O.default_params(min_cov=2, partitions=3, min_contig=200))
        assert tot == [len(km), wd, len(wk)] and (text, nc, trace) == (otext, onc, otrace)
    finally:
        rfx.close()


def test_contexts_and_communicators_made_and_destroyed_in_one_process(golden_dir):
    """What a Spark executor does job after job: a fresh context + communicator, some work, both destroyed -- six times in
    this process, ending with the call that aborted in round 3 (gpurun_out/s3_both.log: SIGABRT inside
    rfx_sharded_assemble_reads, the ninth test of this module when every test made its own context).  The cause was in
    this library, not in RCCL: stopped kernel timers lived in a thread_local list, the k > 31 merge of
    rfx_dev_sharded_count stopped one without ever collecting it, the context died with its events, and the NEXT context's
    first collect() handed the destroyed events to hipEventElapsedTime.  Timers now belong to their context
    (rfx_ctx::timers_pending); this test walks exactly that sequence."""
    import torch
    import reflexiv_amd
    from oracle import oracle as O
    os.environ["RFX_BACKTRACE"] = "1"                      # a host crash inside the library names its frames
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    try:
        for round_ in range(6):
            rfx = reflexiv_amd.Reflexiv()
            rfx.comm_init(reflexiv_amd.Reflexiv.comm_unique_id(), 0, 1)
            try:
                if round_ < 5:
                    # k = 63 in 4 generations: the merge of the generations' shards is order_wide2, whose "sort" timer nobody collected
                    seed, G, n_reads, L = 40 + round_, 200_000, 100_000 + 20_000 * round_, 150
                    dw, wpr = reads_on_device(rfx, seed, G, n_reads, L)
                    k = 63 if round_ % 2 == 0 else 40
                    cap = rfx.kmers_per_read_w(L, k) * n_reads // 2
                    sk = torch.empty(cap * 2, dtype=torch.int64, device="cuda"); sc = torch.empty(cap, dtype=torch.int64, device="cuda")
                    dk = torch.empty(cap * 2, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
                    torch.cuda.synchronize()
                    ms, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, sk.data_ptr(), sc.data_ptr(), cap, 3, generations=4)
                    m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 3)
                    assert (ms, tot) == (m, [inst, nd, m]) and torch.equal(sk[:m * 2], dk[:m * 2]) and torch.equal(sc[:m], dc[:m])
                    assert rfx.workspace_bytes() > 0
                    rfx.trim()
                    assert rfx.workspace_bytes() == 0
                else:
                    prm = reflexiv_amd.default_params(min_cov=3, partitions=4, twin=reflexiv_amd.TWIN_RDD)
                    text, nc, trace, tot = rfx.sharded_assemble_reads(ex["bases"], ex["read_off"], prm, generations=4)
                    assert nc == 2 and text.startswith(">Contig-4558-0\n")
            finally:
                rfx.close()
    finally:
        os.environ.pop("RFX_BACKTRACE", None)


def test_sharded_extend_behind_the_c_abi_one_rank_real_rccl(golden_dir):
    """rfx_dev_sharded_assemble on a ONE-rank communicator of the real RCCL (the rank's own pieces through ncclSend / ncclRecv,
    the sample and carry all-gathers, the per-pass all-reduce all run): the ORACLE's golden contigs and pass-by-pass record
    counts of the documented example (both twins, P = 4 and 8) and of the planted bubble / repeat fixture, with the whole loop
    sharded (gather_below = 0), handed over to the one-GPU driver in the middle, and gathered at once; k = 63 against the
    oracle's driver."""
    import torch
    import reflexiv_amd
    from oracle import oracle as O
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    pl = np.load(os.path.join(golden_dir, "planted.npz"))
    rfx = make_comm({"RFX_COMM_SELF_VIA_RCCL": "1", "RFX_COMM_LIMIT_BYTES": "65536"})
    try:
        rng = np.random.default_rng(9)
        o = rng.permutation(len(ex["keys_cov3"]))                    # a shard comes in any order
        dk = torch.from_numpy(ex["keys_cov3"][o].view(np.int64).copy()).cuda()
        dc = torch.from_numpy(ex["counts_cov3"][o].astype(np.int32)).cuda()
        torch.cuda.synchronize()
        for P, twin, tn in ((4, reflexiv_amd.TWIN_DS, "ds"), (8, reflexiv_amd.TWIN_RDD, "rdd"), (4, reflexiv_amd.TWIN_RDD, "rdd")):
            for gb in (0, 700, -1):
                prm = reflexiv_amd.default_params(min_cov=3, partitions=P, twin=twin)
                text, nc, trace = rfx.sharded_assemble_dev(dk.data_ptr(), dc.data_ptr(), len(o), prm, gather_below=gb)
                assert text == str(ex[f"contigs_{tn}_P{P}"]), (P, tn, gb)
                assert trace == [int(x) for x in ex[f"trace_{tn}_P{P}"]], (P, tn, gb)
        pk = torch.from_numpy(pl["k31_keys"].view(np.int64).copy()).cuda()
        pc = torch.from_numpy(pl["k31_counts"].astype(np.int32)).cuda()
        torch.cuda.synchronize()
        prm = reflexiv_amd.default_params(k=31, min_cov=2, partitions=4, twin=reflexiv_amd.TWIN_DS, min_contig=100)
        text, nc, trace = rfx.sharded_assemble_dev(pk.data_ptr(), pc.data_ptr(), len(pl["k31_keys"]), prm, gather_below=0)
        assert text == str(pl["k31_ds_contigs"]) and trace == [int(x) for x in pl["k31_ds_trace"]]
        # a text buffer that is too short: RFX_E_CAP with the length needed, the retry gives the same text
        text2, nc2, trace2 = rfx.sharded_assemble_dev(pk.data_ptr(), pc.data_ptr(), len(pl["k31_keys"]), prm, gather_below=0, text_cap=100)
        assert rfx.text_retries == 1 and (text2, nc2, trace2) == (text, nc, trace)
        # k = 63 (two-word keys, three-word k-mers; the from-counts extras make the driver gather before iteration 18)
        g = O.synth_genome(11, 50_000)
        bases, off = O.synth_reads(11, g, 50_000, 0, 20_000, 150)
        wk, wc, _, _ = O.count_reads_omp(bases, off, 63, 3)
        ak = O.counter_to_asm_w(wk, 63)
        for extras in (1, 0):
            oprm = O.default_params(k=63, min_cov=3, partitions=4, min_contig=200)
            oprm.extras = extras
            otext, onc, otrace, _ = O.assemble_from_counts(ak, wc.astype(np.int32), oprm)
            prm = reflexiv_amd.default_params(k=63, min_cov=3, partitions=4, min_contig=200)
            prm.extras = extras
            perm = rng.permutation(len(wc))
            dk63 = torch.from_numpy(ak[perm].view(np.int64).copy()).cuda()
            dc63 = torch.from_numpy(wc[perm].astype(np.int32)).cuda()
            torch.cuda.synchronize()
            text, nc, trace = rfx.sharded_assemble_dev(dk63.data_ptr(), dc63.data_ptr(), len(wc), prm, gather_below=0)
            assert (text, nc, trace) == (otext, onc, otrace), extras
    finally:
        rfx.close()
