"""Radix-sharded k-mer counting across the GPUs of one node.

Stands in for the hash shuffle of `reduceByKey` (P/ReflexivMain.java:155;
`groupBy("value").count()` P/ReflexivDSMain.java:207-209): rank r owns the k-mers whose hash
falls in shard r of the k-mer space (owner = mulhi(kmer_hash(kmer), world)), so after ONE
all-to-all(v) every rank holds every instance of its k-mers and counts / filters locally.
One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI) moves the bytes.

The local compute is an `engine`: `HipEngine` (the product: HIP kernels through the C ABI)
or, in the CPU test-suite only, a stand-in injected by the test.  This module never
computes on the CPU itself.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HipEngine:
    """Local compute on one MI355X through libreflexiv_hip.so.

    For k = 28..31 the unit that crosses the exchange is the 16-byte super-k-mer record (two int64
    per record, ~2.6 B per k-mer instance); otherwise the 8-byte canonical k-mer."""

    def __init__(self, rfx, records: bool = True):
        self.rfx = rfx
        self.records = records
        self.k = None
        self.width = 1            # int64 words per exchanged unit

    def _use_records(self, k):
        return self.records and 28 <= k <= 31

    def bucket_by_owner(self, reads, n_owners):
        """reads = dict(words=int64 cuda tensor, n_reads, wpr, read_len, k) ->
        (kmers int64[N] grouped by owner, owner_off int64[n_owners+1] on the host)."""
        self.k = reads["k"]
        n = self.rfx.kmers_per_read(reads["read_len"], reads["k"]) * reads["n_reads"]
        self.n_instances = n
        if self._use_records(reads["k"]):
            self.width = 2
            doff = torch.empty(n_owners + 1, dtype=torch.int64, device=reads["words"].device)
            torch.cuda.current_stream().synchronize()
            args = (reads["words"].data_ptr(), reads["n_reads"], reads["wpr"], reads["read_len"], reads["k"], n_owners)
            nrec, _ = self.rfx.bucket_records_by_owner_dev(*args, 0, 0, doff.data_ptr())
            out = torch.empty(2 * max(1, nrec), dtype=torch.int64, device=reads["words"].device)
            torch.cuda.current_stream().synchronize()
            nrec, h = self.rfx.bucket_records_by_owner_dev(*args, out.data_ptr(), nrec, doff.data_ptr())
            return out[:2 * nrec], torch.from_numpy(h.copy())
        self.width = 1
        out = torch.empty(max(1, n), dtype=torch.int64, device=reads["words"].device)
        doff = torch.empty(n_owners + 1, dtype=torch.int64, device=reads["words"].device)
        torch.cuda.current_stream().synchronize()
        h = self.rfx.bucket_by_owner_dev(reads["words"].data_ptr(), reads["n_reads"], reads["wpr"],
                                         reads["read_len"], reads["k"], n_owners, out.data_ptr(), n,
                                         doff.data_ptr())
        return out[:n], torch.from_numpy(h.copy())

    def count_kmers(self, kmers, min_cov, max_cov, twin):
        from ._lib import RfxError, RFX_E_CAP
        n = int(kmers.numel()) // self.width
        recs = self.width == 2
        n_inst = n * 6 if recs else n
        cap = max(1 << 20, n_inst // 8)         # survivors are few; grow on RFX_E_CAP
        while True:
            keys = torch.empty(cap, dtype=torch.int64, device=kmers.device)
            counts = torch.empty(cap, dtype=torch.int32, device=kmers.device)
            torch.cuda.current_stream().synchronize()
            try:
                if recs:
                    m, d = self.rfx.count_records_dev(kmers.data_ptr(), n, 0, self.k, keys.data_ptr(),
                                                      counts.data_ptr(), cap, min_cov, max_cov, twin)
                else:
                    m, d = self.rfx.count_kmers_dev(kmers.data_ptr(), n, keys.data_ptr(), counts.data_ptr(), cap,
                                                    min_cov, max_cov, twin)
                return keys[:m], counts[:m], d
            except RfxError as e:
                if e.status != RFX_E_CAP or cap >= 16 * n_inst:
                    raise
                del keys, counts
                cap = cap * 4


def exchange_by_owner(kmers: torch.Tensor, owner_off: torch.Tensor, group=None, width: int = 1) -> torch.Tensor:
    """all-to-all(v): send bucket o of `kmers` to rank o, return the concatenation of what
    every rank sent to this one (C2 of SURVEY.md 2.4).  `width` int64 words per unit (2 for
    super-k-mer records); owner_off counts units."""
    world = dist.get_world_size(group)
    send_counts = (owner_off[1:] - owner_off[:-1]).to(torch.int64) * width
    assert send_counts.numel() == world
    recv_counts = torch.empty(world, dtype=torch.int64)
    sc = send_counts.to(kmers.device) if kmers.is_cuda else send_counts
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.cpu()
    recv = torch.empty(int(recv_counts.sum()), dtype=kmers.dtype, device=kmers.device)
    dist.all_to_all_single(recv, kmers, output_split_sizes=[int(x) for x in recv_counts],
                           input_split_sizes=[int(x) for x in send_counts], group=group)
    return recv


def sharded_count(engine, reads, min_cov, max_cov, twin, group=None):
    """-> (keys, counts) of this rank's shard (ascending), global (instances, distinct, survivors)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    kmers, owner_off = engine.bucket_by_owner(reads, world)
    width = getattr(engine, "width", 1)
    recv = exchange_by_owner(kmers, owner_off, group, width) if world > 1 else kmers
    keys, counts, distinct = engine.count_kmers(recv, min_cov, max_cov, twin)
    n_inst = getattr(engine, "n_instances", None)
    tot = torch.tensor([int(kmers.numel()) if n_inst is None or width == 1 else int(n_inst), int(distinct),
                        int(keys.numel())], dtype=torch.int64,
                       device=keys.device)
    if world > 1:
        dist.all_reduce(tot, group=group)            # C4-style scalar reduce
    return keys, counts, [int(x) for x in tot.cpu()]


def gather_survivors(keys: torch.Tensor, counts: torch.Tensor, group=None, root: int = 0):
    """Collect every rank's (kmer, count) shard on `root` (all-gather of padded shards; the
    filtered list is tiny next to the instances: D' << N).  Shards are hash ranges, so the
    concatenation is NOT in k-mer order -- the caller sorts it (rfx sort_pairs) before the
    extend stage, which runs on one GPU this round."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return keys, counts
    n = torch.tensor([int(keys.numel())], dtype=torch.int64, device=keys.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    m = max(1, max(sizes))
    pk = torch.zeros(m, dtype=keys.dtype, device=keys.device); pk[:keys.numel()] = keys
    pc = torch.zeros(m, dtype=counts.dtype, device=counts.device); pc[:counts.numel()] = counts
    gk = [torch.empty_like(pk) for _ in range(world)]
    gc = [torch.empty_like(pc) for _ in range(world)]
    dist.all_gather(gk, pk, group=group)
    dist.all_gather(gc, pc, group=group)
    if dist.get_rank(group) != root:
        return None, None
    return (torch.cat([gk[r][:sizes[r]] for r in range(world)]),
            torch.cat([gc[r][:sizes[r]] for r in range(world)]))
