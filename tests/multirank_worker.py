"""One rank of tests/test_gpu_multirank.py: `python multirank_worker.py RANK WORLD WORKDIR` -- every rank is a process of
its own with its own context and communicator on cuda:0 (RCCL replaced by tests/fake_rccl through RFX_RCCL_LIB, because
RCCL refuses two ranks on one device).  Rank 0 checks the shards against the fused one-GPU count of ALL ranks' reads and
against the oracle, and writes WORKDIR/ok.

`python multirank_worker.py threads WORLD WORKDIR`: the same ranks as THREADS of this one process ("one process or thread
per GPU", include/reflexiv_hip.h) -- how a world of 8, the node size of BASELINE configs 3-5, runs on a test box whose
process guard allows six GPU processes."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    if sys.argv[1] == "threads":
        return main_threads(int(sys.argv[2]), sys.argv[3])
    run_rank(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])


def main_threads(world, work):
    import threading
    import traceback
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")                                # one thread makes the process's HIP context
    failed = []

    def body(r):
        try:
            run_rank(r, world, work, cases=((31, 16), (63, 2)), n_reads=30_000, extend=((31, 4, 0), (63, 3, 0)))
        except BaseException:                                    # noqa: BLE001 -- reported below, with the rank
            failed.append((r, traceback.format_exc()))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r, tb in failed:
        print(f"--- rank {r} of {world} (thread) failed:\n{tb}", flush=True)
    sys.exit(1 if failed else 0)


def sharded_extend_cases(rfx, rank, world, dw, da, n_reads, wpr, L, cover, configs):
    """The extend stage range-sharded over the ranks (rfx_dev_sharded_assemble: every sortByKey as a shuffle of whole records)
    against the ONE-GPU driver on the fused list of all ranks' reads, on rank 0: text, contigs and the pass-by-pass record
    counts must be identical for the same logical partition count P, whatever the number of ranks -- P below, equal to and
    not a multiple of the world (partitions that go on on the next rank), the whole loop sharded (gather_below = 0), a
    hand-over to rank 0 in the middle, and the immediate gather of a small set."""
    import torch
    import reflexiv_amd
    for k, P, gather_below in configs:
        wide = k > 32
        W = 2 if wide else 1
        nk = (rfx.kmers_per_read_w if wide else rfx.kmers_per_read)(L, k)
        cap = nk * n_reads
        cdt = torch.int64 if wide else torch.int32
        sk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); sc = torch.empty(cap, dtype=cdt, device="cuda")
        torch.cuda.synchronize()
        ms, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, sk.data_ptr(), sc.data_ptr(), cap, cover, generations=2)
        prm = reflexiv_amd.default_params(k=k, min_cov=cover, partitions=P, min_contig=300)
        if wide:                                                 # the counter's layout -> the assembler's (KmerBinarizer), shard by shard
            aw = (k - 1) // 31 + 1
            ak = torch.empty(max(1, ms) * aw, dtype=torch.int64, device="cuda"); ac = torch.empty(max(1, ms), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            m2 = rfx.counter_to_asm_dev(sk.data_ptr(), sc.data_ptr(), ms, k, ak.data_ptr(), ac.data_ptr(), cover)
            text, nc, trace = rfx.sharded_assemble_dev(ak.data_ptr(), ac.data_ptr(), m2, prm, gather_below=gather_below)
        else:
            text, nc, trace = rfx.sharded_assemble_dev(sk.data_ptr(), sc.data_ptr(), ms, prm, gather_below=gather_below)
        if rank == 0:
            fk = torch.empty(cap * world * W, dtype=torch.int64, device="cuda"); fc = torch.empty(cap * world, dtype=cdt, device="cuda")
            torch.cuda.synchronize()
            if wide:
                m, nd, inst = rfx.count_reads_w_dev(da.data_ptr(), world * n_reads, wpr, L, k, fk.data_ptr(), fc.data_ptr(), cap * world, cover)
                fa = torch.empty(max(1, m) * aw, dtype=torch.int64, device="cuda"); fcc = torch.empty(max(1, m), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                mm = rfx.counter_to_asm_dev(fk.data_ptr(), fc.data_ptr(), m, k, fa.data_ptr(), fcc.data_ptr(), cover)
                wtext, wnc, wtrace = rfx.assemble_w_dev(fa.data_ptr(), fcc.data_ptr(), mm, prm)
            else:
                m, nd, inst = rfx.count_reads_dev(da.data_ptr(), world * n_reads, wpr, L, k, fk.data_ptr(), fc.data_ptr(), cap * world, cover)
                wtext, wnc, wtrace = rfx.assemble_dev(fk.data_ptr(), fc.data_ptr(), m, prm)
            assert tot[2] == m and wnc > 0, (k, P, gather_below, tot, m, wnc)
            assert trace == wtrace, ("sharded extend: record counts per pass", k, P, gather_below, trace, wtrace)
            assert (text, nc) == (wtext, wnc), ("sharded extend: contigs", k, P, gather_below, nc, wnc)
        else:
            assert text == "" and nc == 0


def run_rank(rank, world, work, cases=((31, 1), (31, 4), (25, 3), (63, 2), (47, 4), (15, 1), (67, 1), (95, 2)), n_reads=60_000,
             extend=((31, 4, 0), (31, 8, 0), (31, 3, 2000), (31, 8, -1), (63, 4, 0), (63, 8, 3000))):
    import torch
    import reflexiv_amd
    from reflexiv_amd import Reflexiv

    torch.cuda.set_device(0)
    idf = os.path.join(work, "id")
    if rank == 0:
        uid = Reflexiv.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 300, "rank 0 never published the id"
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    rfx = Reflexiv(0)
    rfx.comm_init(uid, rank, world)

    seed, G, L = 77, 200_000, 150                              # n_reads per rank
    wpr = (L + 31) // 32
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(seed, G, dg.data_ptr())
    rfx.synth_reads_dev(seed, dg.data_ptr(), G, rank * n_reads, n_reads, L, wpr, dw.data_ptr())
    rfx.sync()
    if rank == 0:
        da = torch.empty(world * n_reads * wpr, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        rfx.synth_reads_dev(seed, dg.data_ptr(), G, 0, world * n_reads, L, wpr, da.data_ptr())
        rfx.sync()

    assert rfx.comm_all_reduce([rank + 1, -rank, 1 << 40]) == [world * (world + 1) // 2, -world * (world - 1) // 2, world << 40]
    assert rfx.comm_all_reduce([rank, 7 - rank], "max") == [world - 1, 7]

    cover = 3
    want = None
    if rank == 0:
        from oracle import oracle as OR
        og = OR.synth_genome(seed, G)
        ob, oo = OR.synth_reads(seed, og, G, 0, world * n_reads, L)
        okm = OR.extract_canon(ob, oo, 31)
        wk31, wc31, wd31 = OR.count_filter(okm, cover)
        want31 = [len(okm), wd31, len(wk31)]
        del ob, oo, okm
    for k, gens in cases:
        want = want31 if (rank == 0 and k == 31) else None
        wide = k > 32
        W = k // 32 + 1 if wide else 1                         # (k = 15, 67, 95: outside the record path, the k-mers themselves travel)
        nk = (rfx.kmers_per_read_w if wide else rfx.kmers_per_read)(L, k)
        cap = nk * n_reads
        cdt = torch.int64 if wide else torch.int32
        sk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); sc = torch.empty(cap, dtype=cdt, device="cuda")
        torch.cuda.synchronize()
        ms, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, sk.data_ptr(), sc.data_ptr(), cap, cover, generations=gens)
        assert tot[0] == nk * n_reads * world
        # the shard is ascending (the order contract of every count-stage output)
        if ms > 1 and not wide:
            s = sk[:ms]
            assert bool((s[1:] > s[:-1]).all()), (k, gens, "shard not ascending")
        allm = rfx.comm_all_reduce([ms])[0]
        assert allm == tot[2]
        gk = torch.empty(max(1, allm) * W, dtype=torch.int64, device="cuda") if rank == 0 else sk[:0]
        gc = torch.empty(max(1, allm), dtype=cdt, device="cuda") if rank == 0 else sc[:0]
        torch.cuda.synchronize()
        got = rfx.gather_shards_dev(sk.data_ptr(), sc.data_ptr(), ms, W, 8 if wide else 4, 0, gk.data_ptr(), gc.data_ptr(), allm)
        # a root buffer that is too small fails on root only, before anybody sends
        if allm > 1:
            try:
                rfx.gather_shards_dev(sk.data_ptr(), sc.data_ptr(), ms, W, 8 if wide else 4, 0, gk.data_ptr(), gc.data_ptr(), allm - 1)
                assert rank != 0
            except reflexiv_amd.RfxError as e:
                assert rank == 0 and e.need == allm
        if rank == 0:
            assert got == allm
            fk = torch.empty(cap * world * W, dtype=torch.int64, device="cuda"); fc = torch.empty(cap * world, dtype=cdt, device="cuda")
            torch.cuda.synchronize()
            if wide and W > 2:
                m, nd, inst = rfx.count_reads_w_dev(da.data_ptr(), world * n_reads, wpr, L, k, fk.data_ptr(), fc.data_ptr(), cap * world, cover)
                hk = gk[:got * W].cpu().numpy().view(np.uint64).reshape(-1, W); hc = gc[:got].cpu().numpy()
                o = np.lexsort([hk[:, w] for w in range(W - 1, -1, -1)])      # ascending by (word 0, word 1, ...)
                rk = torch.from_numpy(hk[o].reshape(-1).view(np.int64).copy()).cuda(); rc = torch.from_numpy(hc[o].copy()).cuda()
            elif wide:
                m, nd, inst = rfx.count_reads_w_dev(da.data_ptr(), world * n_reads, wpr, L, k, fk.data_ptr(), fc.data_ptr(), cap * world, cover)
                rfx.order_kmers_w_dev(gk.data_ptr(), gc.data_ptr(), got, k)
                rfx.sync()
                rk, rc = gk[:got * W], gc[:got]
            else:
                m, nd, inst = rfx.count_reads_dev(da.data_ptr(), world * n_reads, wpr, L, k, fk.data_ptr(), fc.data_ptr(), cap * world, cover)
                tk = torch.empty_like(gk); tv = torch.empty_like(gc)
                rfx.sort_pairs_dev(gk.data_ptr(), gc.data_ptr(), got, 2 * k, tk.data_ptr(), tv.data_ptr())
                rfx.sync()
                rk, rc = gk[:got], gc[:got]
            if want is not None and not wide:
                # the truth first, so that a mismatch names the side that is wrong
                assert [inst, nd, m] == want, ("fused count vs oracle", k, gens, [inst, nd, m], want)
            assert tot == [inst, nd, m], ("sharded totals vs fused", k, gens, tot, [inst, nd, m])
            assert torch.equal(rk, fk[:m * W]) and torch.equal(rc, fc[:m]), (k, gens)
            # every k-mer lives on exactly one rank: the gathered list has no repeats (it equals the fused list) -- done above

    sharded_extend_cases(rfx, rank, world, dw, da if rank == 0 else None, n_reads, wpr, L, cover, extend)

    # host ASCII reads of any length -> contig text on rank 0: the documented example dealt round the ranks
    from oracle import oracle as O
    ex = np.load(os.path.join(HERE, "golden", "example.npz"))
    bases, off = ex["bases"], ex["read_off"]
    nr = len(off) - 1
    mine = np.arange(rank, nr, world)
    mb = np.concatenate([bases[off[i]:off[i + 1]] for i in mine])
    mo = np.zeros(len(mine) + 1, np.int64)
    mo[1:] = np.cumsum(off[mine + 1] - off[mine])
    prm = reflexiv_amd.default_params(min_cov=3, partitions=4, twin=reflexiv_amd.TWIN_RDD)
    text, nc, trace, tot = rfx.sharded_assemble_reads(mb, mo, prm, generations=2)
    # ... and with every sortByKey of the extend stage as a range shuffle over the ranks, to the end of the loop
    text_s, nc_s, trace_s, tot_s = rfx.sharded_assemble_reads(mb, mo, prm, generations=2, gather_below=0)
    assert (text_s, nc_s, trace_s, tot_s) == (text, nc, trace, tot)
    if rank == 0:
        km = O.extract_canon(bases, off, 31)
        wk, wc, wd = O.count_filter(km, 3)
        otext, onc, otrace, _ = O.assemble_from_counts(wk, wc, O.default_params(min_cov=3, partitions=4, twin=O.TWIN_RDD))
        assert tot == [len(km), wd, len(wk)]
        assert (text, nc, trace) == (otext, onc, otrace) and text.startswith(">Contig-4558-0\n")
    else:
        assert text == "" and nc == 0
    if world > 1:
        #  rank 0 brings NO reads and a text buffer that is too short: RFX_E_CAP must come back on every rank,
        # with the length rank 0 needs, so that all of them repeat the collective together (rank 0 alone used to wait for
        # ever in the first all-reduce); an empty rank also sends nothing and receives its whole shard (receive buffer grown
        # by agreement)
        if rank == 0:
            mb2, mo2 = np.zeros(0, np.uint8), np.zeros(1, np.int64)
        else:
            mine = np.arange(rank - 1, nr, world - 1)
            mb2 = np.concatenate([bases[off[i]:off[i + 1]] for i in mine])
            mo2 = np.zeros(len(mine) + 1, np.int64)
            mo2[1:] = np.cumsum(off[mine + 1] - off[mine])
        text2, nc2, trace2, tot2 = rfx.sharded_assemble_reads(mb2, mo2, prm, generations=2, text_cap=1000)
        assert rfx.text_retries == 1, rfx.text_retries          # on EVERY rank
        if rank == 0:
            assert tot2 == tot and (text2, nc2, trace2) == (otext, onc, otrace)
        else:
            assert text2 == "" and nc2 == 0
    rfx.comm_all_reduce([1])                                   # nobody leaves while a peer still reads its files
    rfx.close()
    with open(os.path.join(work, f"ok{rank}"), "w") as f:
        f.write("ok\n")


if __name__ == "__main__":
    main()
