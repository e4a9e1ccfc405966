"""world_size-2 gloo rehearsal of the multi-GPU count path (reflexiv_amd/dist.py).

The exchange logic (owner buckets -> all-to-all(v) -> local count -> scalar all-reduce) is
the product code; the local compute is replaced, in this CPU test only, by an engine built
on the oracle that buckets with the same owner function as the HIP kernel
(owner = mulhi(kmer_hash(kmer), world), reflexiv_amd/csrc/rfx_kmer.hip k_owner_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O

M = np.uint64(0xFFFFFFFFFFFFFFFF)


def kmer_hash(x):
    """reflexiv_amd/csrc/rfx_device.h kmer_hash"""
    with np.errstate(over="ignore"):
        h = x.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return h ^ (h >> np.uint64(32))


def owner_of(kmers, n):
    h = kmer_hash(kmers)
    n = np.uint64(n)
    lo, hi = h & np.uint64(0xFFFFFFFF), h >> np.uint64(32)
    return ((hi * n + ((lo * n) >> np.uint64(32))) >> np.uint64(32)).astype(np.int64)


def wide_hash(hi, lo):
    """reflexiv_amd/csrc/rfx_kmer.hip wide_hash (two-word k-mers, k = 33..63)"""
    with np.errstate(over="ignore"):
        x = hi.astype(np.uint64) ^ (lo.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        x = x * np.uint64(0xD6E8FEB86659FD93)
    return x ^ (x >> np.uint64(32))


def mulhi_owner(h, n):
    n = np.uint64(n)
    lo, hi = h & np.uint64(0xFFFFFFFF), h >> np.uint64(32)
    return ((hi * n + ((lo * n) >> np.uint64(32))) >> np.uint64(32)).astype(np.int64)


class OracleWideEngine:
    """CPU stand-in for HipEngine on the k > 31 path (tests only): 16-byte elements {word0, word1}."""
    width = 2

    def bucket_by_owner(self, reads, n_owners):
        self.k = reads["k"]
        km = O.extract_canon_w(reads["bases"], reads["read_off"], reads["k"])        # [n, 2]
        self.n_instances = len(km)
        own = mulhi_owner(wide_hash(km[:, 0], km[:, 1]), n_owners)
        order = np.argsort(own, kind="stable")
        off = np.zeros(n_owners + 1, np.int64)
        np.cumsum(np.bincount(own, minlength=n_owners), out=off[1:])
        return torch.from_numpy(np.ascontiguousarray(km[order]).reshape(-1).view(np.int64)), torch.from_numpy(off)

    def count_kmers(self, kmers, min_cov, max_cov, twin):
        k2, c, d = O.count_filter_w(kmers.numpy().view(np.uint64).reshape(-1, 2), self.k, min_cov, max_cov)
        return torch.from_numpy(k2.reshape(-1).view(np.int64)), torch.from_numpy(c), d

    def split_reads(self, reads, chunks):
        return OracleEngine.split_reads(self, reads, chunks)


class OracleEngine:
    """CPU stand-in for HipEngine (tests only)."""

    def bucket_by_owner(self, reads, n_owners):
        km = O.extract_canon(reads["bases"], reads["read_off"], reads["k"])
        own = owner_of(km, n_owners)
        order = np.argsort(own, kind="stable")
        off = np.zeros(n_owners + 1, np.int64)
        np.cumsum(np.bincount(own, minlength=n_owners), out=off[1:])
        return torch.from_numpy(km[order].view(np.int64)), torch.from_numpy(off)

    def count_kmers(self, kmers, min_cov, max_cov, twin):
        k, c, d = O.count_filter(kmers.numpy().view(np.uint64), min_cov, max_cov, twin)
        return torch.from_numpy(k.view(np.int64)), torch.from_numpy(c), d

    def merge_sorted(self, keys_l, counts_l):
        k = np.concatenate([x.numpy() for x in keys_l]).view(np.uint64); c = np.concatenate([x.numpy() for x in counts_l])
        o = np.argsort(k, kind="stable")
        return torch.from_numpy(k[o].view(np.int64)), torch.from_numpy(c[o])

    def split_reads(self, reads, chunks):
        off = reads["read_off"]
        n = len(off) - 1
        cuts = [n * c // chunks for c in range(chunks + 1)]
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                out.append(dict(bases=reads["bases"][off[a]:off[b]], read_off=off[a:b + 1] - off[a], k=reads["k"]))
        return out


class OracleCombineEngine(OracleEngine):
    """CPU stand-in for HipEngine(combine=True) (tests only): local count first, 16-byte {k-mer, count}
    pairs cross the exchange, the owner sums the partial counts."""
    width = 2

    def bucket_by_owner(self, reads, n_owners):
        km = O.extract_canon(reads["bases"], reads["read_off"], reads["k"])
        self.n_instances = len(km)
        k, c, _ = O.count_filter(km, 1)
        own = owner_of(k, n_owners)
        order = np.argsort(own, kind="stable")
        off = np.zeros(n_owners + 1, np.int64)
        np.cumsum(np.bincount(own, minlength=n_owners), out=off[1:])
        pairs = np.stack([k[order], c[order].astype(np.uint64)], axis=1).reshape(-1)
        return torch.from_numpy(pairs.view(np.int64)), torch.from_numpy(off)

    def count_kmers(self, pairs, min_cov, max_cov, twin):
        p = pairs.numpy().view(np.uint64).reshape(-1, 2)
        keys, inv = np.unique(p[:, 0], return_inverse=True)
        sums = np.bincount(inv, weights=p[:, 1].astype(np.float64), minlength=len(keys)).astype(np.int64)
        keep = np.ones(len(keys), bool) if (twin == O.TWIN_RDD and min_cov <= 1) else (sums >= min_cov) & (sums <= max_cov)
        return torch.from_numpy(keys[keep].view(np.int64)), torch.from_numpy(sums[keep].astype(np.int32)), len(keys)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, seed, G, per_rank, L, k, min_cov, q, chunks=1, limit_bytes=None, combine=False, generations=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reflexiv_amd import dist as rd
        if limit_bytes:
            rd.A2A_LIMIT_BYTES = limit_bytes          # force the message-size rounds
        g = O.synth_genome(seed, G)
        bases, off = O.synth_reads(seed, g, G, rank * per_rank, per_rank, L)    # this rank's read shard
        reads = dict(bases=bases, read_off=off, k=k)
        engine = OracleCombineEngine() if combine else OracleEngine()
        keys, counts, tot = rd.sharded_count(engine, reads, min_cov, 10_000_000, O.TWIN_DS, chunks=chunks, generations=generations)
        allk, allc = rd.gather_survivors(keys, counts)
        if rank == 0:
            q.put(("root", allk.numpy().view(np.uint64).copy(), allc.numpy().copy(), None))
        else:
            assert allk is None
        q.put((rank, keys.numpy().view(np.uint64).copy(), counts.numpy().copy(), tot))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks,limit_bytes,combine", [(2, 1, None, False), (3, 1, None, False), (2, 4, None, False),
                                                              (3, 3, None, False), (2, 1, 40_000, False), (3, 2, 24_000, False),
                                                              (2, 1, None, True), (3, 3, None, True), (2, 2, 24_000, True),
                                                              (2, -4, None, False), (3, -3, None, False), (2, -2, 9_000, False)])
def test_sharded_count_equals_global_count(world, chunks, limit_bytes, combine):
    """chunks < 0: -chunks GENERATIONS of the hash space instead (bucketed once by (generation, owner), the
    all-to-alls launched back to back, generation g counted while the later ones travel)."""
    generations = 1
    if chunks < 0:
        generations, chunks = -chunks, 1
    seed, G, per_rank, L, k, min_cov = 42, 20_000, 1500, 100, 31, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, G, per_rank, L, k, min_cov, q, chunks, limit_bytes, combine,
                                               generations))
             for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    root = [t for t in got if t[0] == "root"][0]
    res = sorted([t for t in got if t[0] != "root"], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, per_rank * world, L)
    km = O.extract_canon(bases, off, k)
    wk, wc, wd = O.count_filter(km, min_cov)
    # global scalars agree on every rank
    for _, _, _, tot in res:
        assert tot == [len(km), wd, len(wk)]
    # shards are disjoint, each ascending, owner-consistent, and their union is the global answer
    allk = np.concatenate([r[1] for r in res]); allc = np.concatenate([r[2] for r in res])
    for rank, kk, cc, _ in res:
        assert np.all(kk[1:] > kk[:-1])
        assert np.all(owner_of(kk, world * generations) % world == rank)       # bin g * world + o of an owner function over G * world bins
    order = np.argsort(allk, kind="stable")
    assert np.array_equal(allk[order], wk) and np.array_equal(allc[order], wc)
    # the root's gathered list is the same multiset, in rank order
    assert np.array_equal(root[1], allk) and np.array_equal(root[2], allc)


def test_owner_function_is_balanced_and_total():
    rng = np.random.default_rng(1)
    km = rng.integers(0, 1 << 62, 200_000, dtype=np.uint64)
    for n in (1, 2, 4, 8):
        own = owner_of(km, n)
        assert own.min() >= 0 and own.max() == n - 1
        cnt = np.bincount(own, minlength=n)
        assert cnt.min() > 0.9 * len(km) / n


# ------------------------------------------------------------------ range-sharded extend stage

class OracleOps:
    """CPU stand-in for reflexiv_amd.dist.HipOps (tests only): the oracle's operators."""

    def make(self, key, marker, ext_off, ext, left, right):
        return O.Records(np.ascontiguousarray(key, np.uint64), np.ascontiguousarray(marker, np.int32),
                         np.ascontiguousarray(ext_off, np.int64), np.ascontiguousarray(ext, np.uint64),
                         np.ascontiguousarray(left, np.int32), np.ascontiguousarray(right, np.int32))

    def sort_pairs(self, keys, counts):
        o = np.argsort(keys, kind="stable")
        return keys[o], counts[o]

    def rc_expand(self, keys, counts, k):
        return O.rc_expand_subkmer(keys, counts, k)

    def sort(self, r):
        return O.sort_records(r)

    def fork_forward(self, r, ps, k, min_err, twin):
        return O.fork_filter_forward(r, ps, k, min_err, twin)

    def reflect(self, r, k):
        return O.reflect_from_forward(r, k)

    def fork_reflected(self, r, ps, k, min_err, twin):
        return O.fork_filter_reflected(r, ps, k, min_err, twin)

    def random_reflection(self, r, ps, k):
        return O.random_reflection(r, ps, k)

    def extend_pass(self, r, ps, k, twin, stage):
        return O.extend_pass(r, ps, k, twin)

    def contigs_text(self, r, k, min_contig, twin):
        return O.contigs_text(r, k, min_contig, twin)


def _asm_worker(rank, world, port, keys, counts, prm_kw, q, limit_bytes=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reflexiv_amd import dist as rd
        if limit_bytes:
            rd.A2A_LIMIT_BYTES = limit_bytes
        mine = owner_of(keys, world) == rank                     # the hash shard the count stage leaves on this rank
        prm = O.default_params(**prm_kw)
        trace = []
        text, nc = rd.sharded_assemble(OracleOps(), keys[mine], counts[mine], prm, trace=trace)
        if rank == 0:
            q.put((text, nc, trace))
        else:
            assert text is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_sharded_assemble(world, keys, counts, prm_kw, limit_bytes=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_asm_worker, args=(r, world, port, keys, counts, prm_kw, q, limit_bytes)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,P,twin", [(2, 4, "ds"), (2, 4, "rdd"), (4, 4, "ds"), (2, 8, "ds")])
def test_sharded_extend_reproduces_the_example(golden_dir, world, P, twin):
    """Range-sharded extend stage on 2 / 4 ranks == the single-process driver for the same logical
    partition count (documented example: two contigs of 4558 bp at P = 4)."""
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    tw = O.TWIN_DS if twin == "ds" else O.TWIN_RDD
    text, nc, trace = _run_sharded_assemble(world, ex["keys_cov3"], ex["counts_cov3"],
                                            dict(min_cov=3, partitions=P, twin=tw))
    assert text == str(ex[f"contigs_{twin}_P{P}"])
    assert trace == [int(x) for x in ex[f"trace_{twin}_P{P}"]]


def test_sharded_extend_with_bubbles_and_repeat(golden_dir):
    """planted fixture (SNP bubble + repeat: fork-marked records, left/right >= 0 branches)."""
    pl = np.load(os.path.join(golden_dir, "planted.npz"))
    text, nc, trace = _run_sharded_assemble(2, pl["k31_keys"], pl["k31_counts"],
                                            dict(k=31, min_cov=2, partitions=4, twin=O.TWIN_DS, min_contig=100),
                                            limit_bytes=4096)          # record exchanges in several rounds
    assert text == str(pl["k31_ds_contigs"])
    assert trace == [int(x) for x in pl["k31_ds_trace"]]


def test_sharded_extend_world1_without_process_group(golden_dir):
    from reflexiv_amd import dist as rd
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    prm = O.default_params(min_cov=3, partitions=4, twin=O.TWIN_DS)
    rng = np.random.default_rng(3)
    o = rng.permutation(len(ex["keys_cov3"]))                   # any input order
    text, nc = rd.sharded_assemble(OracleOps(), ex["keys_cov3"][o], ex["counts_cov3"][o], prm)
    assert text == str(ex["contigs_ds_P4"])


def _wide_worker(rank, world, port, seed, G, per_rank, L, k, min_cov, q, chunks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reflexiv_amd import dist as rd
        g = O.synth_genome(seed, G)
        bases, off = O.synth_reads(seed, g, G, rank * per_rank, per_rank, L)
        reads = dict(bases=bases, read_off=off, k=k)
        keys, counts, tot = rd.sharded_count(OracleWideEngine(), reads, min_cov, 10_000_000, 0, chunks=chunks)
        q.put((rank, keys.numpy().view(np.uint64).reshape(-1, 2).copy(), counts.numpy().copy(), tot))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 2)])
def test_sharded_wide_count_equals_global_count(world, chunks):
    """k = 63 (two-word k-mers): owner = mulhi(wide_hash, world); 16-byte elements cross the exchange."""
    seed, G, per_rank, L, k, min_cov = 7, 15_000, 900, 120, 63, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, seed, G, per_rank, L, k, min_cov, q, chunks))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = O.synth_genome(seed, G)
    bases, off = O.synth_reads(seed, g, G, 0, per_rank * world, L)
    km = O.extract_canon_w(bases, off, k)
    wk, wc, wd = O.count_filter_w(km, k, min_cov)
    for rank, kk, cc, tot in res:
        assert tot == [len(km), wd, len(wk)]
        assert np.all(mulhi_owner(wide_hash(kk[:, 0], kk[:, 1]), world) == rank)
    allk = np.concatenate([r[1] for r in res]); allc = np.concatenate([r[2] for r in res])
    order = np.lexsort((allk[:, 1], allk[:, 0]))
    assert np.array_equal(allk[order], wk) and np.array_equal(allc[order], wc)


# ------------------------------------------------------------------ the same stage, records resident on the device
# (reflexiv_amd.dist.sharded_assemble_dev): here the "device" is the CPU and the operators are the oracle's

class OracleTOps:
    """CPU stand-in for reflexiv_amd.dist.HipDevOps (tests only): the oracle's operators on tensor record sets."""

    def _rec(self, r):
        return O.Records(r.key[:r.n].numpy().view(np.uint64).copy(), r.marker[:r.n].numpy().copy(),
                         r.ext_off[:r.n + 1].numpy().copy(), r.ext[:r.words].numpy().view(np.uint64).copy(),
                         r.left[:r.n].numpy().copy(), r.right[:r.n].numpy().copy())

    def _t(self, o):
        from reflexiv_amd.dist import TRecs
        tt = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt).copy())
        return TRecs(tt(o.key, np.int64), tt(o.marker, np.int32), tt(o.ext_off, np.int64), tt(o.ext, np.int64),
                     tt(o.left, np.int32), tt(o.right, np.int32), o.n, int(o.ext_off[o.n]), 1)

    def make(self, key, marker, ext_off, ext, left, right, kw=1):
        from reflexiv_amd.dist import TRecs
        return TRecs(key.contiguous(), marker.contiguous(), ext_off.contiguous(), ext.contiguous(), left.contiguous(),
                     right.contiguous(), int(marker.numel()), int(ext.numel()), kw)

    def sort_pairs(self, keys, counts, k):
        kk = keys.numpy().view(np.uint64)
        o = np.argsort(kk, kind="stable")
        return torch.from_numpy(kk[o].view(np.int64).copy()), torch.from_numpy(counts.numpy()[o].copy())

    def rc_expand(self, keys, counts, k):
        return self._t(O.rc_expand_subkmer(keys.numpy().view(np.uint64), counts.numpy(), k))

    def sort(self, r, k):
        return self._t(O.sort_records(self._rec(r)))

    def fork_forward(self, r, ps, k, min_err, twin):
        o, ops = O.fork_filter_forward(self._rec(r), ps.numpy(), k, min_err, twin)
        return self._t(o), torch.from_numpy(ops)

    def fork_reflected(self, r, ps, k, min_err, twin):
        o, ops = O.fork_filter_reflected(self._rec(r), ps.numpy(), k, min_err, twin)
        return self._t(o), torch.from_numpy(ops)

    def reflect(self, r, k):
        return self._t(O.reflect_from_forward(self._rec(r), k))

    def random_reflection(self, r, ps, k):
        return self._t(O.random_reflection(self._rec(r), ps.numpy(), k))

    def extend_pass(self, r, ps, k, twin, stage):
        o, ops = O.extend_pass(self._rec(r), ps.numpy(), k, twin)
        return self._t(o), torch.from_numpy(ops)

    def lower_bound(self, sorted_keys, values, upper):
        return torch.from_numpy(np.searchsorted(sorted_keys.numpy().view(np.uint64), values.numpy().view(np.uint64),
                                                side="right" if upper else "left").astype(np.int64))

    def contigs_text(self, r, k, min_contig, twin):
        return O.contigs_text(self._rec(r), k, min_contig, twin)


def _asm_dev_worker(rank, world, port, keys, counts, prm_kw, q, limit_bytes=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reflexiv_amd import dist as rd
        if limit_bytes:
            rd.A2A_LIMIT_BYTES = limit_bytes
        mine = owner_of(keys, world) == rank
        prm = O.default_params(**prm_kw)
        trace = []
        text, nc = rd.sharded_assemble_dev(OracleTOps(), torch.from_numpy(keys[mine].view(np.int64).copy()),
                                           torch.from_numpy(counts[mine].astype(np.int32)), prm, trace=trace)
        if rank == 0:
            q.put((text, nc, trace))
        else:
            assert text is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_sharded_assemble_dev(world, keys, counts, prm_kw, limit_bytes=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_asm_dev_worker, args=(r, world, port, keys, counts, prm_kw, q, limit_bytes)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,P,twin", [(2, 4, "ds"), (4, 4, "rdd"), (2, 8, "ds")])
def test_device_resident_sharded_extend_reproduces_the_example(golden_dir, world, P, twin):
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    tw = O.TWIN_DS if twin == "ds" else O.TWIN_RDD
    text, nc, trace = _run_sharded_assemble_dev(world, ex["keys_cov3"], ex["counts_cov3"], dict(min_cov=3, partitions=P, twin=tw))
    assert text == str(ex[f"contigs_{twin}_P{P}"])
    assert trace == [int(x) for x in ex[f"trace_{twin}_P{P}"]]


def test_device_resident_sharded_extend_with_bubbles_and_repeat(golden_dir):
    pl = np.load(os.path.join(golden_dir, "planted.npz"))
    text, nc, trace = _run_sharded_assemble_dev(2, pl["k31_keys"], pl["k31_counts"],
                                                dict(k=31, min_cov=2, partitions=4, twin=O.TWIN_DS, min_contig=100),
                                                limit_bytes=4096)          # record exchanges in several rounds
    assert text == str(pl["k31_ds_contigs"])
    assert trace == [int(x) for x in pl["k31_ds_trace"]]


def test_device_resident_sharded_extend_world1_and_splitter_rounds(golden_dir):
    """no process group: the one-rank path; and the eight-bits-per-round splitter search against the exact boundaries"""
    from reflexiv_amd import dist as rd
    ex = np.load(os.path.join(golden_dir, "example.npz"))
    prm = O.default_params(min_cov=3, partitions=4, twin=O.TWIN_DS)
    rng = np.random.default_rng(3)
    o = rng.permutation(len(ex["keys_cov3"]))
    text, nc = rd.sharded_assemble_dev(OracleTOps(), torch.from_numpy(ex["keys_cov3"][o].view(np.int64).copy()),
                                       torch.from_numpy(ex["counts_cov3"][o].astype(np.int32)), prm)
    assert text == str(ex["contigs_ds_P4"])
    for bits, n, P in ((60, 5000, 7), (62, 100, 4), (13, 3000, 16), (60, 3, 5)):
        keys = np.sort(rng.integers(0, 1 << bits, n, dtype=np.uint64))
        keys[n // 3:n // 3 + n // 10] = keys[n // 3]                       # a long run of equal keys across a boundary
        keys = np.sort(keys)
        v, incl, ng = rd.splitters_t(OracleTOps(), torch.from_numpy(keys.view(np.int64).copy()), P, bits)
        wv, wincl, _ = rd._splitters_local(keys, P)
        assert ng == n and np.array_equal(v, wv) and np.array_equal(incl, wincl)
