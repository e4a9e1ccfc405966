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

import os

import torch
import torch.distributed as dist


class HipEngine:
    """Local compute on one MI355X through libreflexiv_hip.so.

    For k = 21..31 the unit that crosses the exchange is the 16-byte super-k-mer record (two int64
    per record, ~2.6 B per k-mer instance); otherwise the 8-byte canonical k-mer."""

    def __init__(self, rfx, records: bool = True, combine: bool = False, wide_records: bool = True):
        self.rfx = rfx
        self.records = records
        self.wide_records = wide_records      # k = 33..63: ship 32-byte super-k-mer records instead of 16-byte k-mers
        self.combine = combine    # k <= 31: count locally first, ship (k-mer, partial count) pairs
        self.k = None
        self.width = 1            # int64 words per exchanged unit
        self.timing = {}          # kernel family -> [ms, launches] over every library call since the last reset

    def hint_instances(self, n):
        self._inst_acc = int(n)

    def generations_ok(self, reads):
        return reads["k"] <= 31 and not self.combine

    def merge_sorted(self, keys_l, counts_l):
        """ascending shards of disjoint k-mer sets -> one ascending shard (k <= 31)"""
        keys, counts = torch.cat(keys_l), torch.cat(counts_l)
        m = int(keys.numel())
        if len(keys_l) > 1 and m > 1:
            tk, tc = torch.empty_like(keys), torch.empty_like(counts)
            torch.cuda.current_stream().synchronize()
            self.rfx.sort_pairs_dev(keys.data_ptr(), counts.data_ptr(), m, 2 * self.k, tk.data_ptr(), tc.data_ptr())
            self.rfx.sync()
        return keys, counts

    def _take_hint(self):
        h, self._inst_acc = int(getattr(self, "_inst_acc", 0) or 0), 0
        return h

    def _acc_timing(self):
        for name, (ms, ln) in self.rfx.count_timing().items():
            a = self.timing.setdefault(name, [0.0, 0])
            a[0] += ms; a[1] += ln

    # ---- combine form (reduceByKey's map-side combine, P/ReflexivMain.java:155): the unit that crosses the
    # exchange is a 16-byte {k-mer, local count} pair, one per DISTINCT k-mer of the rank's reads -- at high
    # coverage several times fewer bytes than one unit per instance (5 Gbp of a 4.6 Mbp genome: 1.4 B per
    # instance against 2.6 for records), and the local count runs while the previous chunk's pairs travel.
    def _combine_by_owner(self, reads, n_owners):
        from ._lib import RfxError, RFX_E_CAP
        dev = reads["words"].device
        nk = self.rfx.kmers_per_read(reads["read_len"], reads["k"])
        n = nk * reads["n_reads"]
        self.n_instances, self.width = n, 2
        if getattr(self, "_pairs_per_read", None) is None:
            self._pairs_per_read = nk / 4.0 + 1.0
        cap = int(self._pairs_per_read * reads["n_reads"]) + (1 << 20)
        doff = torch.empty(n_owners + 1, dtype=torch.int64, device=dev)
        while True:
            scratch = torch.empty(2 * cap, dtype=torch.int64, device=dev)
            out = torch.empty(2 * cap, dtype=torch.int64, device=dev)      # (same sizes every step: no allocator churn)
            torch.cuda.current_stream().synchronize()
            try:
                m, h, _ = self.rfx.combine_reads_dev(reads["words"].data_ptr(), reads["n_reads"], reads["wpr"],
                                                     reads["read_len"], reads["k"], n_owners, scratch.data_ptr(),
                                                     out.data_ptr(), cap, doff.data_ptr())
                self._acc_timing()
                return out[:2 * m], torch.from_numpy(h.copy())
            except RfxError as e:
                if e.status != RFX_E_CAP:
                    raise
                del scratch, out
                cap = int(1.05 * getattr(e, "need", 2 * cap)) + (1 << 20)
                self._pairs_per_read = cap / max(1, reads["n_reads"])       # only ever grows

    def _merge_pairs(self, pairs, min_cov, max_cov, twin):
        from ._lib import RfxError, RFX_E_CAP
        n = int(pairs.numel()) // 2
        cap = max(1 << 20, n // 8)
        while True:
            keys = torch.empty(cap, dtype=torch.int64, device=pairs.device)
            counts = torch.empty(cap, dtype=torch.int32, device=pairs.device)
            torch.cuda.current_stream().synchronize()
            try:
                m, d = self.rfx.merge_pairs_dev(pairs.data_ptr(), n, self.k, keys.data_ptr(), counts.data_ptr(), cap,
                                                min_cov, max_cov, twin)
                self._acc_timing()
                return keys[:m], counts[:m], d
            except RfxError as e:
                if e.status != RFX_E_CAP or cap >= n:
                    raise
                del keys, counts
                cap = min(n, cap * 4)

    def _use_records(self, k):
        import os
        # (RFX_SUPERKMER=0 is the library's ablation knob for the record path: follow it)
        return self.records and 21 <= k <= 31 and os.environ.get("RFX_SUPERKMER", "1") != "0"

    def bucket_by_owner(self, reads, n_owners):
        """reads = dict(words=int64 cuda tensor, n_reads, wpr, read_len, k) ->
        (kmers int64[N] grouped by owner, owner_off int64[n_owners+1] on the host)."""
        self.k = reads["k"]
        if reads["k"] > 31 and self.wide_records:       # k = 33..63: 32-byte super-k-mer records (~5 B per instance)
            dev = reads["words"].device
            nk = self.rfx.kmers_per_read_w(reads["read_len"], reads["k"])
            self.n_instances, self.width = nk * reads["n_reads"], 4
            self._inst_acc = getattr(self, "_inst_acc", 0) + self.n_instances
            if getattr(self, "_wrec_per_read", None) is None:
                self._wrec_per_read = nk / 5.0 + 1.0
            doff = torch.empty(n_owners + 1, dtype=torch.int64, device=dev)
            args = (reads["words"].data_ptr(), reads["n_reads"], reads["wpr"], reads["read_len"], reads["k"], n_owners)
            cap = int(self._wrec_per_read * reads["n_reads"]) + 4096
            while True:
                out = torch.empty(4 * max(1, cap), dtype=torch.int64, device=dev)
                torch.cuda.current_stream().synchronize()
                nrec, h = self.rfx.bucket_wide_records_by_owner_dev(*args, out.data_ptr(), cap, doff.data_ptr())
                if h is not None:
                    self._acc_timing()
                    return out[:4 * nrec], torch.from_numpy(h.copy())
                del out
                cap = nrec
                self._wrec_per_read = 1.03 * nrec / max(1, reads["n_reads"])     # only ever grows
        if reads["k"] > 31:                             # two-word k-mers (k = 33..63): 16-byte elements
            n = self.rfx.kmers_per_read_w(reads["read_len"], reads["k"]) * reads["n_reads"]
            self.n_instances, self.width = n, 2
            out = torch.empty(2 * max(1, n), dtype=torch.int64, device=reads["words"].device)
            doff = torch.empty(n_owners + 1, dtype=torch.int64, device=reads["words"].device)
            torch.cuda.current_stream().synchronize()
            h = self.rfx.bucket_wide_by_owner_dev(reads["words"].data_ptr(), reads["n_reads"], reads["wpr"],
                                                  reads["read_len"], reads["k"], n_owners, out.data_ptr(), n, doff.data_ptr())
            return out[:2 * n], torch.from_numpy(h.copy())
        if self.combine:
            return self._combine_by_owner(reads, n_owners)
        n = self.rfx.kmers_per_read(reads["read_len"], reads["k"]) * reads["n_reads"]
        self.n_instances = n
        if self._use_records(reads["k"]):
            self.width = 2
            doff = torch.empty(n_owners + 1, dtype=torch.int64, device=reads["words"].device)
            torch.cuda.current_stream().synchronize()
            args = (reads["words"].data_ptr(), reads["n_reads"], reads["wpr"], reads["read_len"], reads["k"], n_owners)
            # capacity from the records-per-read ratio of the previous call (one histogram pass instead of
            # two); the call reports the need if the guess is short
            ratio = self._records_per_read(reads)
            cap = int(ratio * reads["n_reads"]) + 4096
            out = torch.empty(2 * max(1, cap), dtype=torch.int64, device=reads["words"].device)
            torch.cuda.current_stream().synchronize()
            nrec, h = self.rfx.bucket_records_by_owner_dev(*args, out.data_ptr() if cap else 0, cap, doff.data_ptr())
            if h is None or nrec > cap:
                out = torch.empty(2 * max(1, nrec), dtype=torch.int64, device=reads["words"].device)
                torch.cuda.current_stream().synchronize()
                nrec, h = self.rfx.bucket_records_by_owner_dev(*args, out.data_ptr(), nrec, doff.data_ptr())
            self._acc_timing()
            if nrec > cap:                             # only ever grows: buffer sizes stay the same from step to step
                self._rec_per_read = 1.03 * nrec / max(1, reads["n_reads"])
            return out[:2 * nrec], torch.from_numpy(h.copy())
        self.width = 1
        out = torch.empty(max(1, n), dtype=torch.int64, device=reads["words"].device)
        doff = torch.empty(n_owners + 1, dtype=torch.int64, device=reads["words"].device)
        torch.cuda.current_stream().synchronize()
        h = self.rfx.bucket_by_owner_dev(reads["words"].data_ptr(), reads["n_reads"], reads["wpr"],
                                         reads["read_len"], reads["k"], n_owners, out.data_ptr(), n,
                                         doff.data_ptr())
        return out[:n], torch.from_numpy(h.copy())

    def estimate_units(self, reads):
        """records (k-mers) this rank's reads produce, from the ratio seen in the previous call; 0 = unknown"""
        if reads["k"] > 31 and self.wide_records:
            nk = self.rfx.kmers_per_read_w(reads["read_len"], reads["k"])
            return int((getattr(self, "_wrec_per_read", None) or nk / 5.0 + 1.0) * reads["n_reads"])
        if reads["k"] > 31:
            return self.rfx.kmers_per_read_w(reads["read_len"], reads["k"]) * reads["n_reads"]
        if self.combine:
            nk = self.rfx.kmers_per_read(reads["read_len"], reads["k"])
            return int((getattr(self, "_pairs_per_read", None) or nk / 4.0 + 1.0) * reads["n_reads"])
        if self._use_records(reads["k"]):
            return int(self._records_per_read(reads) * reads["n_reads"])
        return self.rfx.kmers_per_read(reads["read_len"], reads["k"]) * reads["n_reads"]

    def _records_per_read(self, reads):
        """capacity planning: super-k-mer records a read yields -- one per ~6.2 windows on real-looking
        sequence; start at one per 5 and grow only if a call reports more (identical buffer sizes in
        every step keep torch's caching allocator from re-allocating)"""
        if getattr(self, "_rec_per_read", None) is None:
            self._rec_per_read = self.rfx.kmers_per_read(reads["read_len"], reads["k"]) / 5.0 + 1.0
        return self._rec_per_read

    def split_reads(self, reads, chunks):
        """the packed read set as `chunks` contiguous slices (views, nothing is copied)"""
        n, wpr = reads["n_reads"], reads["wpr"]
        cuts = [n * c // chunks for c in range(chunks + 1)]
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                sub = dict(reads)
                sub["words"] = reads["words"][a * wpr:b * wpr]
                sub["n_reads"] = b - a
                out.append(sub)
        return out

    def count_kmers(self, kmers, min_cov, max_cov, twin):
        from ._lib import RfxError, RFX_E_CAP
        if self.k is not None and self.k > 31:          # -> keys int64[2m] (two words per k-mer), counts int64[m]
            recs = self.width == 4
            hint = self._take_hint()        # (by symmetry a rank receives about as many instances as its own reads hold)
            n = int(kmers.numel()) // self.width
            cap = max(1 << 20, (6 * n if recs else n) // 8)
            while True:
                keys = torch.empty(2 * cap, dtype=torch.int64, device=kmers.device)
                counts = torch.empty(cap, dtype=torch.int64, device=kmers.device)
                torch.cuda.current_stream().synchronize()
                try:
                    if recs:
                        m, d = self.rfx.count_wide_records_dev(kmers.data_ptr(), n, hint, self.k, keys.data_ptr(),
                                                               counts.data_ptr(), cap, min_cov, max_cov)
                    else:
                        m, d = self.rfx.count_wide_elems_dev(kmers.data_ptr(), n, self.k, keys.data_ptr(), counts.data_ptr(),
                                                             cap, min_cov, max_cov)
                    self._acc_timing()
                    return keys[:2 * m], counts[:m], d
                except RfxError as e:
                    if e.status != RFX_E_CAP or cap >= 16 * n:
                        raise
                    del keys, counts
                    cap = cap * 4
        if self.combine:
            return self._merge_pairs(kmers, min_cov, max_cov, twin)
        n = int(kmers.numel()) // self.width
        recs = self.width == 2
        hint31 = self._take_hint() if recs else 0
        n_inst = n * 6 if recs else n
        cap = max(1 << 20, n_inst // 8)         # survivors are few; grow on RFX_E_CAP
        while True:
            keys = torch.empty(cap, dtype=torch.int64, device=kmers.device)
            counts = torch.empty(cap, dtype=torch.int32, device=kmers.device)
            torch.cuda.current_stream().synchronize()
            try:
                if recs:
                    m, d = self.rfx.count_records_dev(kmers.data_ptr(), n, hint31, self.k, keys.data_ptr(),
                                                      counts.data_ptr(), cap, min_cov, max_cov, twin)
                else:
                    m, d = self.rfx.count_kmers_dev(kmers.data_ptr(), n, keys.data_ptr(), counts.data_ptr(), cap,
                                                    min_cov, max_cov, twin)
                self._acc_timing()
                return keys[:m], counts[:m], d
            except RfxError as e:
                if e.status != RFX_E_CAP or cap >= 16 * n_inst:
                    raise
                del keys, counts
                cap = cap * 4


# RCCL 2.26 (the one in this torch build) returns corrupt data from all_to_all_single when a per-peer
# message exceeds 1 GiB (tools/dbg_dist.py on a one-rank group: intact at exactly 2^27 int64, the
# second half wrong just above).  No single call below ever moves more than A2A_LIMIT_BYTES per
# peer: larger buckets go in rounds through staging buffers.
A2A_LIMIT_BYTES = 1 << 29      # per peer per call (512 MiB)

# One-rank groups only: hand the rank's own bucket over by a device copy instead of through RCCL.  OFF by default so
# that every test and every real run goes through dist.all_to_all_single (the 512 MiB rounds and the async handles
# included); only the bench's one-rank rehearsal (`bench.py --force-dist`) turns it on, and its numbers are then
# labelled exchange-free.
LOCAL_SHORTCUT = os.environ.get("RFX_DIST_LOCAL_SHORTCUT", "0") == "1"


def _alltoallv(send: torch.Tensor, send_counts, group=None, async_op: bool = False, limit: int = None, out=None,
               recv_counts=None, rounds: int = None):
    """all-to-all(v) of `send` (bucket d = send_counts[d] elements, buckets back to back) ->
    (recv, [work handles]); recv holds source 0's bucket, then source 1's, ...  Per-peer messages are
    capped at `limit` elements per call (default: A2A_LIMIT_BYTES).  recv_counts / rounds: already
    agreed by the caller (no handshake, no host synchronisation before the data moves)."""
    limit = limit or max(1, A2A_LIMIT_BYTES // send.element_size())
    world = dist.get_world_size(group)
    send_counts = [int(x) for x in send_counts]
    if recv_counts is None:
        sc = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=group)
        recv_counts = [int(x) for x in rc.cpu()]
    else:
        recv_counts = [int(x) for x in recv_counts]
    if rounds is None:
        mx = torch.tensor([max(send_counts + recv_counts + [0])], dtype=torch.int64, device=send.device)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        rounds = max(1, -(-int(mx.item()) // limit))
    need = sum(recv_counts)
    if world == 1 and LOCAL_SHORTCUT:
        # one rank (the rehearsal of the multi-GPU branch on a single GPU): the only bucket is this rank's own, which on
        # N ranks is the 1/N that never crosses a link -- hand it over as it is (or by one device copy into the caller's
        # buffer) instead of through RCCL's self-copy, which moves it at a sixth of the HBM rate
        if out is None or out.numel() < need:
            return send[:need], []
        out[:need].copy_(send[:need])
        return out[:need], []
    # `out` (optional): a caller-owned buffer whose head receives the data when it is large enough
    recv = out[:need] if out is not None and out.numel() >= need else torch.empty(need, dtype=send.dtype, device=send.device)
    if rounds == 1:
        w = dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group,
                                   async_op=async_op)
        return recv, ([w] if async_op else [])
    soff = [0]
    for c in send_counts:
        soff.append(soff[-1] + c)
    roff = [0]
    for c in recv_counts:
        roff.append(roff[-1] + c)
    for j in range(rounds):
        ins = [max(0, min(limit, send_counts[d] - j * limit)) for d in range(world)]
        outs = [max(0, min(limit, recv_counts[d] - j * limit)) for d in range(world)]
        stage_in = torch.cat([send[soff[d] + j * limit: soff[d] + j * limit + ins[d]] for d in range(world)])
        stage_out = torch.empty(sum(outs), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(stage_out, stage_in, output_split_sizes=outs, input_split_sizes=ins, group=group)
        o = 0
        for d in range(world):
            recv[roff[d] + j * limit: roff[d] + j * limit + outs[d]] = stage_out[o:o + outs[d]]
            o += outs[d]
    return recv, []


def exchange_by_owner(kmers: torch.Tensor, owner_off: torch.Tensor, group=None, width: int = 1) -> torch.Tensor:
    """all-to-all(v): send bucket o of `kmers` to rank o, return the concatenation of what
    every rank sent to this one (C2 of SURVEY.md 2.4).  `width` int64 words per unit (2 for
    super-k-mer records); owner_off counts units."""
    send_counts = (owner_off[1:] - owner_off[:-1]).to(torch.int64) * width
    assert send_counts.numel() == dist.get_world_size(group)
    recv, _ = _alltoallv(kmers, send_counts.tolist(), group)
    return recv


def exchange_by_owner_async(kmers: torch.Tensor, owner_off: torch.Tensor, group=None, width: int = 1, out=None):
    """exchange_by_owner, but the data all-to-all(v) is only LAUNCHED when it fits one call: ->
    (recv buffer, work handles).  The caller keeps bucketing the next chunk of reads while the bytes
    move (xGMI is the slowest stage of the multi-GPU step), and waits on the handles before it
    reads `recv`."""
    send_counts = (owner_off[1:] - owner_off[:-1]).to(torch.int64) * width
    return _alltoallv(kmers, send_counts.tolist(), group, async_op=True, out=out)


def _sharded_count_generations(engine, reads, min_cov, max_cov, twin, group, world, G):
    """The hash space in G generations: the reads are bucketed ONCE by (generation, owner) -- bin g*world + o
    of an owner function over G*world bins -- the G all-to-alls are launched back to back, and generation g
    is counted while g+1.. are still travelling (a k-mer lives in exactly one generation, so the G counts
    are independent).  The pipeline for a link-bound exchange: after the first generation has arrived the
    step is bound by max(exchange, count) instead of their sum."""
    km, off = engine.bucket_by_owner(reads, world * G)
    width = getattr(engine, "width", 1)
    engine.bucketed_bytes = getattr(engine, "bucketed_bytes", 0) + int(km.numel()) * km.element_size()
    n_inst = getattr(engine, "n_instances", None)
    off = off.to(torch.int64)
    per = (off[1:] - off[:-1]).reshape(G, world)                      # [g, dest] units
    dev = km.device
    sc = (per.t().contiguous() * width).to(dev)                       # [dest, g] words: G numbers for every peer
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc.view(-1), sc.view(-1), group=group)                    # one handshake for all generations
    rc = rc.cpu().reshape(world, G)                                    # [src, g] words coming from every peer
    sc = sc.cpu()
    limit = max(1, A2A_LIMIT_BYTES // km.element_size())
    mx = torch.tensor([int(max(int(sc.max()) if sc.numel() else 0, int(rc.max()) if rc.numel() else 0))],
                      dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(mx.item()) // limit))
    recvs, works = [], []
    for g in range(G):
        a, b = int(off[g * world]) * width, int(off[(g + 1) * world]) * width
        r, w = _alltoallv(km[a:b], sc[:, g].tolist(), group, async_op=True, recv_counts=rc[:, g].tolist(), rounds=rounds)
        recvs.append(r); works.append(w)
    keys_l, counts_l, distinct = [], [], 0
    for g in range(G):
        for w in works[g]:
            w.wait()
        if hasattr(engine, "hint_instances") and n_inst:
            engine.hint_instances(int(n_inst) // G)
        k_, c_, d_ = engine.count_kmers(recvs[g], min_cov, max_cov, twin)
        keys_l.append(k_); counts_l.append(c_); distinct += int(d_)
        recvs[g] = None
    keys, counts = engine.merge_sorted(keys_l, counts_l)
    tot = torch.tensor([int(km.numel()) if n_inst is None or width == 1 else int(n_inst), distinct, int(counts.numel())],
                       dtype=torch.int64, device=keys.device)
    if world > 1:
        dist.all_reduce(tot, group=group)
    return keys, counts, [int(x) for x in tot.cpu()]


def _sharded_count_capi(engine, rfx, reads, min_cov, max_cov, twin, generations):
    from ._lib import RfxError, RFX_E_CAP
    k = reads["k"]
    wide = k > 32
    W = 2 if wide else 1
    dev = reads["words"].device
    nk = rfx.kmers_per_read_w(reads["read_len"], k) if wide else rfx.kmers_per_read(reads["read_len"], k)
    cap = max(1 << 20, nk * reads["n_reads"] // 8)
    while True:
        keys = torch.empty(cap * W, dtype=torch.int64, device=dev)
        counts = torch.empty(cap, dtype=torch.int64 if wide else torch.int32, device=dev)
        torch.cuda.current_stream().synchronize()
        try:
            m, tot = rfx.sharded_count_dev(reads["words"].data_ptr(), reads["n_reads"], reads["wpr"], reads["read_len"], k,
                                           keys.data_ptr(), counts.data_ptr(), cap, min_cov, max_cov, twin,
                                           generations=max(1, generations))
            break
        except RfxError as e:
            if e.status != RFX_E_CAP:
                raise
            cap = max(2 * cap, rfx.comm_all_reduce([int(e.need)], "max")[0])       # (RFX_E_CAP arrives on every rank at once)
    engine.bucketed_bytes = getattr(engine, "bucketed_bytes", 0) + rfx.comm_bytes_bucketed()
    engine.k = k
    if hasattr(engine, "_acc_timing"):
        engine._acc_timing()
    return keys[:m * W], counts[:m], tot


def sharded_count(engine, reads, min_cov, max_cov, twin, group=None, chunks: int = 1, generations: int = 1):
    """-> (keys, counts) of this rank's shard (ascending), global (instances, distinct, survivors).
    chunks > 1 (and an engine that can split its reads): the reads are bucketed chunk by chunk and
    chunk c travels while chunk c+1 is bucketed; counting starts when everything has arrived.
    generations > 1 (engines with `merge_sorted`): see _sharded_count_generations."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    width = None
    exchange = world > 1 or (dist.is_initialized() and bool(getattr(engine, "force_exchange", False)))   # (tests: 1-rank RCCL)
    rfx = getattr(engine, "rfx", None)
    if (exchange and rfx is not None and getattr(rfx, "comm", None) and not getattr(engine, "combine", False)
            and getattr(rfx, "comm_world", 0) == world and (21 <= reads["k"] <= 31 or 33 <= reads["k"] <= 63)):
        # the exchange lives behind the C ABI (rfx_dev_sharded_count: RCCL send / recv inside libreflexiv_hip.so); this module
        # only allocates the outputs.  Engines without a communicator (the gloo stand-ins of the CPU tests, the pairs / k-mer
        # exchange units) take the torch.distributed forms below.
        return _sharded_count_capi(engine, rfx, reads, min_cov, max_cov, twin, generations)
    if (exchange and generations > 1 and hasattr(engine, "merge_sorted") and world * generations <= 64
            and getattr(engine, "generations_ok", lambda r: True)(reads)):
        return _sharded_count_generations(engine, reads, min_cov, max_cov, twin, group, world, generations)
    if exchange and chunks > 1 and hasattr(engine, "split_reads"):
        parts, sent, n_inst_total, pending = [], 0, 0, []
        subs = engine.split_reads(reads, chunks)
        # one receive buffer for all chunks when the engine can estimate the volume (by symmetry a
        # rank receives about as much as it produces); a chunk that does not fit is joined by a copy
        est = engine.estimate_units(reads) if hasattr(engine, "estimate_units") else 0
        big, pos = None, 0
        for sub in subs:
            km, off = engine.bucket_by_owner(sub, world)
            width = getattr(engine, "width", 1)
            if big is None and est:
                big = torch.empty(int(est * width * 1.1) + 4096, dtype=km.dtype, device=km.device)
            sent += int(km.numel())
            engine.bucketed_bytes = getattr(engine, "bucketed_bytes", 0) + int(km.numel()) * km.element_size()
            n_inst_total += int(getattr(engine, "n_instances", 0) or 0)
            recv_c, works = exchange_by_owner_async(km, off, group, width, out=None if big is None else big[pos:])
            if big is not None and recv_c.numel() and recv_c.data_ptr() == big[pos:].data_ptr():
                pos += recv_c.numel()
            elif recv_c.numel():
                parts.append(recv_c)
            pending.append((works, km))                    # km stays alive until its send has completed
        for works, _ in pending:
            for w in works:
                w.wait()
        head = [big[:pos]] if big is not None and pos else []
        recv = (head + parts)[0] if len(head + parts) == 1 else torch.cat(head + parts) if head + parts else km[:0]
        del pending, parts
        kmers_numel, n_inst = sent, (n_inst_total if width != 1 else None)
    else:
        kmers, owner_off = engine.bucket_by_owner(reads, world)
        engine.bucketed_bytes = getattr(engine, "bucketed_bytes", 0) + int(kmers.numel()) * kmers.element_size()
        width = getattr(engine, "width", 1)
        recv = exchange_by_owner(kmers, owner_off, group, width) if exchange else kmers
        kmers_numel, n_inst = int(kmers.numel()), getattr(engine, "n_instances", None)
    keys, counts, distinct = engine.count_kmers(recv, min_cov, max_cov, twin)
    tot = torch.tensor([kmers_numel if n_inst is None or width == 1 else int(n_inst), int(distinct),
                        int(counts.numel())], dtype=torch.int64,
                       device=keys.device)
    if world > 1:
        dist.all_reduce(tot, group=group)            # C4-style scalar reduce
    return keys, counts, [int(x) for x in tot.cpu()]


def gather_survivors(keys: torch.Tensor, counts: torch.Tensor, group=None, root: int = 0, words: int = 1):
    """Collect every rank's (kmer, count) shard on `root` (all-gather of padded shards; the
    filtered list is tiny next to the instances: D' << N).  Shards are hash ranges, so the
    concatenation is NOT in k-mer order -- the caller sorts it (rfx sort_pairs) before the
    extend stage, which runs on one GPU this round."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return keys, counts
    keys = keys.reshape(-1)                      # `words` words per k-mer (k > 31: the counter's AoS layout)
    n = torch.tensor([int(counts.numel())], dtype=torch.int64, device=keys.device)
    assert keys.numel() == counts.numel() * words
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    m = max(1, max(sizes))
    pk = torch.zeros(m * words, dtype=keys.dtype, device=keys.device); pk[:keys.numel()] = keys
    pc = torch.zeros(m, dtype=counts.dtype, device=counts.device); pc[:counts.numel()] = counts
    gk = [torch.empty_like(pk) for _ in range(world)]
    gc = [torch.empty_like(pc) for _ in range(world)]
    dist.all_gather(gk, pk, group=group)
    dist.all_gather(gc, pc, group=group)
    if dist.get_rank(group) != root:
        return None, None
    return (torch.cat([gk[r][:sizes[r] * words] for r in range(world)]),
            torch.cat([gc[r][:sizes[r]] for r in range(world)]))


# ----------------------------------------------------------------------------------------------
# Range-sharded extend stage: every sort of the reference's loop
# (sortByKey, P/ReflexivMain.java:179,191,211,235,247,286) becomes
#   local stable sort -> exact splitters at the LOGICAL partition boundaries (order contract B.0:
#   floor(p*n/P) moved forward past equal keys) by a bitwise distributed selection (one small
#   all-reduce per key bit) -> ONE all-to-all(v) of whole records -> local stable sort.
# Rank r owns the logical partitions [r*P/W, (r+1)*P/W): equal keys never straddle ranks, every
# per-partition operator (fork filters, random reflection, extend pass) runs locally with its own
# partition starts, and the result does not depend on the number of ranks.  Records travel as the
# flat arrays of the reference layout; ties keep (source rank, local position) order, which is the
# global arrival order because ranks hold consecutive partitions.
#
# The local operators come from an `ops` object: `HipOps` (the C ABI operators of
# libreflexiv_hip.so) or, in the CPU test-suite only, a stand-in injected by the test.

import numpy as np


class HipOps:
    """The per-operator C ABI (include/reflexiv_hip.h) behind the names the sharded driver uses."""

    def __init__(self, rfx):
        self.rfx = rfx

    def make(self, key, marker, ext_off, ext, left, right):
        from .api import Records
        return Records(np.ascontiguousarray(key, np.uint64), np.ascontiguousarray(marker, np.int32),
                       np.ascontiguousarray(ext_off, np.int64), np.ascontiguousarray(ext, np.uint64),
                       np.ascontiguousarray(left, np.int32), np.ascontiguousarray(right, np.int32))

    def sort_pairs(self, keys, counts):
        import torch as _t
        dk = _t.from_numpy(keys.view(np.int64)).cuda(); dc = _t.from_numpy(counts.astype(np.int32)).cuda()
        tk, tc = _t.empty_like(dk), _t.empty_like(dc)
        _t.cuda.synchronize()
        self.rfx.sort_pairs_dev(dk.data_ptr(), dc.data_ptr(), int(dk.numel()), 64, tk.data_ptr(), tc.data_ptr())
        self.rfx.sync()
        return dk.cpu().numpy().view(np.uint64), dc.cpu().numpy()

    def rc_expand(self, keys, counts, k):
        return self.rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(keys, counts, k)

    def sort(self, r):
        return self.rfx.sortByKey(r, 1)[0]

    def fork_forward(self, r, ps, k, min_err, twin):
        return self.rfx.FilterForkSubKmer(r, ps, k, min_err, twin)

    def reflect(self, r, k):
        return self.rfx.ReflectedSubKmerExtractionFromForward(r, k)

    def fork_reflected(self, r, ps, k, min_err, twin):
        return self.rfx.FilterForkReflectedSubKmer(r, ps, k, min_err, twin)

    def random_reflection(self, r, ps, k):
        return self.rfx.kmerRandomReflection(r, ps, k)

    def extend_pass(self, r, ps, k, twin, stage):
        return self.rfx.ExtendReflexivKmer(r, ps, k, twin, stage)

    def contigs_text(self, r, k, min_contig, twin):
        return self.rfx.KmerToContig(r, k, min_contig, twin)


def _coll_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _allreduce_i64(a: np.ndarray, group=None) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(_coll_device())
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()


def _alltoall_var(chunks, dtype, group=None):
    """chunks[d] = 1-D numpy array for rank d -> list of what every rank sent here (in rank order)."""
    world = dist.get_world_size(group)
    dev = _coll_device()
    counts = [len(c) for c in chunks]
    send = np.ascontiguousarray(np.concatenate(chunks) if world else np.empty(0, dtype))
    st = torch.from_numpy(send.view(np.int64 if send.dtype.itemsize == 8 else np.int32)).to(dev)
    # the receive counts come back inside _alltoallv; recover the cuts from a second tiny exchange
    sc = torch.tensor(counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    rcl = [int(x) for x in rc.cpu()]
    rt, _ = _alltoallv(st, counts, group)
    out = rt.cpu().numpy().view(dtype)
    cuts = np.cumsum([0] + rcl)
    return [out[cuts[i]:cuts[i + 1]] for i in range(world)]


def splitters(sorted_keys: np.ndarray, P: int, key_bits: int, group=None):
    """Boundary of every logical partition p = 1..P-1 over the GLOBAL sorted order of all ranks' keys:
    (v[p-1], incl[p-1]) -- partition p starts at the first key >= v (incl) or > v (the run of v began
    before floor(p*n/P) and is not split).  Exact; key_bits + 2 all-reduces of P-1 integers."""
    n_glob = int(_allreduce_i64(np.array([len(sorted_keys)]), group)[0])
    q = np.array([(p * n_glob) // P for p in range(1, P)], np.int64)
    v = np.zeros(P - 1, np.uint64)
    if n_glob == 0 or P == 1:
        return v, np.ones(P - 1, bool), n_glob
    for bit in range(key_bits - 1, -1, -1):
        t = v | np.uint64(1 << bit)
        c = _allreduce_i64(np.searchsorted(sorted_keys, t, side="left"), group)
        v = np.where(c <= q, t, v)           # largest v with count(keys < v) <= q  ==  key at global rank q
    c_lt = _allreduce_i64(np.searchsorted(sorted_keys, v, side="left"), group)
    return v, c_lt == q, n_glob


def _local_bounds(sorted_keys, v, incl):
    """index in the local sorted keys where each partition p = 1..P-1 starts"""
    lo = np.searchsorted(sorted_keys, v, side="left")
    hi = np.searchsorted(sorted_keys, v, side="right")
    return np.where(incl, lo, hi).astype(np.int64)


def sort_exchange(ops, r, P: int, key_bits: int, group=None):
    """Global stable sort by key of the ranks' record sets (in rank-major arrival order) ->
    (this rank's logical partitions, sorted; their local partition starts [P/W + 1])."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert P % world == 0, "the logical partition count must be a multiple of the number of ranks"
    per = P // world
    r = ops.sort(r)
    if world == 1:
        if P == 1:
            return r, np.array([0, r.n], np.int64)
        v, incl, _ = _splitters_local(r.key, P)
        return r, np.concatenate([[0], _local_bounds(r.key, v, incl), [r.n]]).astype(np.int64)
    v, incl, _ = splitters(r.key, P, key_bits, group)
    b = np.concatenate([[0], _local_bounds(r.key, v, incl), [r.n]]).astype(np.int64)     # partition p = [b[p], b[p+1])
    rcut = b[::per]                                                                        # rank d = [rcut[d], rcut[d+1])
    wcut = r.ext_off[rcut]
    lens = (r.ext_off[1:] - r.ext_off[:-1]).astype(np.int64)
    parts = {}
    for name, arr, cut, dt in (("key", r.key, rcut, np.uint64), ("marker", r.marker, rcut, np.int32),
                               ("len", lens, rcut, np.int64), ("ext", r.ext, wcut, np.uint64),
                               ("left", r.left, rcut, np.int32), ("right", r.right, rcut, np.int32)):
        got = _alltoall_var([arr[cut[d]:cut[d + 1]] for d in range(world)], dt, group)
        parts[name] = np.concatenate(got) if got else np.empty(0, dt)
    ext_off = np.concatenate([[0], np.cumsum(parts["len"])]).astype(np.int64)
    merged = ops.sort(ops.make(parts["key"], parts["marker"], ext_off, parts["ext"], parts["left"], parts["right"]))
    p0 = rank * per
    lb = _local_bounds(merged.key, v, incl)                                                # all P-1 boundaries, local indices
    full = np.concatenate([[0], lb, [merged.n]]).astype(np.int64)
    ps = full[p0:p0 + per + 1].copy()
    ps[0], ps[-1] = 0, merged.n
    return merged, ps


def _splitters_local(sorted_keys, P):
    n = len(sorted_keys)
    q = np.array([(p * n) // P for p in range(1, P)], np.int64)
    if n == 0:
        return np.zeros(P - 1, np.uint64), np.ones(P - 1, bool), 0
    v = sorted_keys[q]
    return v, np.searchsorted(sorted_keys, v, side="left") == q, n


def sharded_assemble(ops, keys: np.ndarray, counts: np.ndarray, prm, group=None, root: int = 0, trace=None):
    """Extend stage on range-sharded records.  keys/counts: this rank's shard of the filtered
    (k-mer, count) list in any order (e.g. the hash shard sharded_count() leaves).  Returns
    (contig text, n_contigs) on `root`, (None, None) elsewhere; identical to the single-GPU driver
    for the same prm.partitions (P/ReflexivMain.java:168-316)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    k, twin, P = prm.k, prm.twin, max(1, prm.partitions)
    keys = np.ascontiguousarray(keys, np.uint64); counts = np.ascontiguousarray(counts, np.int32)
    # the count stage's order contract: ascending canonical k-mer -> range-shard the survivors first
    if world > 1:
        keys, counts = ops.sort_pairs(keys, counts)
        v, incl, _ = splitters(keys, world, 2 * k, group)
        cut = np.concatenate([[0], _local_bounds(keys, v, incl), [len(keys)]]).astype(np.int64)
        gk = _alltoall_var([keys[cut[d]:cut[d + 1]] for d in range(world)], np.uint64, group)
        gc = _alltoall_var([counts[cut[d]:cut[d + 1]] for d in range(world)], np.int32, group)
        keys, counts = ops.sort_pairs(np.concatenate(gk), np.concatenate(gc))
    else:
        keys, counts = ops.sort_pairs(keys, counts)
    kb = 2 * (k - 1)
    r = ops.rc_expand(keys, counts, k)
    r, ps = sort_exchange(ops, r, P, kb, group)
    r, ps = ops.fork_forward(r, ps, k, prm.min_error_cov, twin)
    r = ops.reflect(r, k)
    r, ps = sort_exchange(ops, r, P, kb, group)
    r, ps = ops.fork_reflected(r, ps, k, prm.min_error_cov, twin)
    r = ops.random_reflection(r, ps, k)

    def count(rr):
        return int(_allreduce_i64(np.array([rr.n]), group)[0]) if world > 1 else rr.n

    def one_pass(rr, stage):
        rr, pst = sort_exchange(ops, rr, P, kb, group)
        rr, _ = ops.extend_pass(rr, pst, k, twin, stage)
        if trace is not None:
            trace.append(count(rr))
        return rr
    r = one_pass(r, 0)
    it = 0
    for _ in range(3):
        it += 1
        r = one_pass(r, 0)
    it += 1
    r = one_pass(r, 1)
    last = 0
    while it <= prm.max_iter:
        it += 1
        if it >= prm.min_iter and it % 3 == 0:
            c = count(r)
            if c == last:
                break
            last = c
        r = one_pass(r, 2)
    # the surviving records are few: contig ids follow the global record order, so the text is made on root
    if world > 1:
        lens = (r.ext_off[1:] - r.ext_off[:-1]).astype(np.int64)
        fields = {}
        for name, arr, dt in (("key", r.key, np.uint64), ("marker", r.marker, np.int32), ("len", lens, np.int64),
                              ("ext", r.ext, np.uint64), ("left", r.left, np.int32), ("right", r.right, np.int32)):
            got = _alltoall_var([arr if d == root else arr[:0] for d in range(world)], dt, group)
            fields[name] = np.concatenate(got)
        if rank != root:
            return None, None
        off = np.concatenate([[0], np.cumsum(fields["len"])]).astype(np.int64)
        r = ops.make(fields["key"], fields["marker"], off, fields["ext"], fields["left"], fields["right"])
    return ops.contigs_text(r, k, prm.min_contig, twin)


# ----------------------------------------------------------------------------------------------
# The same range-sharded extend stage with the records RESIDENT in HBM: record sets are torch tensors on the rank's
# GPU, every operator is a device-level C-ABI call (rfx_dev_*, include/reflexiv_hip.h), the splitter search is
# rfx_dev_lower_bound on the locally sorted keys (eight key bits per round: 255 candidates per boundary, one
# all-reduce of the counts), and the shuffle is all_to_all_single on the device tensors -- nothing but a few
# boundary values and counts ever visits the host.  What C5-scale inputs need (6e9 records do not fit one GPU and
# cannot be staged through numpy).  The driver below is written against an `ops` object and plain tensor slicing, so
# the CPU test-suite runs it on 2 / 4 gloo ranks with an oracle-backed stand-in (tests/test_dist_gloo.py) and the GPU
# suite runs it with the real operators on a one-rank RCCL group.

from dataclasses import dataclass


@dataclass
class TRecs:
    """a record set as tensors on one device (reference layout; key: int64 bit patterns, kw words per record)"""
    key: torch.Tensor
    marker: torch.Tensor
    ext_off: torch.Tensor
    ext: torch.Tensor
    left: torch.Tensor
    right: torch.Tensor
    n: int
    words: int
    kw: int = 1


class HipDevOps:
    """device-level record operators of libreflexiv_hip.so on TRecs (every tensor on the context's GPU)"""

    def __init__(self, rfx):
        self.rfx = rfx
        import ctypes as C
        from ._lib import CRecords
        self.C, self.CRecords = C, CRecords

    def _c(self, r: TRecs):
        c = self.CRecords()
        c.n, c.need_words, c.key_words = r.n, r.words, r.kw
        c.key, c.marker, c.ext_off = r.key.data_ptr(), r.marker.data_ptr(), r.ext_off.data_ptr()
        c.ext, c.left, c.right = r.ext.data_ptr(), r.left.data_ptr(), r.right.data_ptr()
        c.cap_n, c.cap_words = int(r.marker.numel()), int(r.ext.numel())
        return c

    def _empty(self, dev, cap_n, cap_words, kw):
        cap_n, cap_words = max(1, cap_n), max(1, cap_words)
        i64, i32 = torch.int64, torch.int32
        return TRecs(torch.empty(cap_n * kw, dtype=i64, device=dev), torch.empty(cap_n, dtype=i32, device=dev),
                     torch.empty(cap_n + 1, dtype=i64, device=dev), torch.empty(cap_words, dtype=i64, device=dev),
                     torch.empty(cap_n, dtype=i32, device=dev), torch.empty(cap_n, dtype=i32, device=dev), 0, 0, kw)

    def _trim(self, o: TRecs, c):
        n, w = int(c.n), int(c.need_words)
        return TRecs(o.key[:n * o.kw], o.marker[:n], o.ext_off[:n + 1], o.ext[:w], o.left[:n], o.right[:n], n, w, o.kw)

    def _run(self, name, fn, *args):
        torch.cuda.current_stream().synchronize()
        self.rfx._check(fn(self.rfx.ctx, *args), name)

    def make(self, key, marker, ext_off, ext, left, right, kw=1):
        n = int(marker.numel())
        return TRecs(key.contiguous(), marker.contiguous(), ext_off.contiguous(), ext.contiguous(), left.contiguous(),
                     right.contiguous(), n, int(ext.numel()), kw)

    def sort_pairs(self, keys, counts, k):
        keys, counts = keys.contiguous().clone(), counts.contiguous().clone()
        tk, tc = torch.empty_like(keys), torch.empty_like(counts)
        torch.cuda.current_stream().synchronize()
        self.rfx.sort_pairs_dev(keys.data_ptr(), counts.data_ptr(), int(keys.numel()), 2 * k, tk.data_ptr(), tc.data_ptr())
        self.rfx.sync()
        return keys, counts

    def rc_expand(self, keys, counts, k):
        from .api import sub_words
        n = int(counts.numel())
        o = self._empty(keys.device, 2 * n, 2 * n, sub_words(k))
        c = self._c(o)
        self._run("rfx_dev_rc_expand_subkmer", self.rfx.L.rfx_dev_rc_expand_subkmer, self.C.c_void_p(keys.data_ptr()),
                  self.C.c_void_p(counts.data_ptr()), self.C.c_int64(n), k, self.C.byref(c))
        return self._trim(o, c)

    def sort(self, r: TRecs, k):
        o = self._empty(r.marker.device, r.n, r.words, r.kw)
        ci, co = self._c(r), self._c(o)
        self._run("rfx_dev_sort_records", self.rfx.L.rfx_dev_sort_records, self.C.byref(ci), 1, k, self.C.byref(co), None)
        return self._trim(o, co)

    def _with_ps(self, name, fn, r, ps, pre, post):
        P = int(ps.numel()) - 1
        o = self._empty(r.marker.device, r.n, r.words, r.kw)
        ops = torch.empty(P + 1, dtype=torch.int64, device=r.marker.device)
        ci, co = self._c(r), self._c(o)
        self._run(name, fn, *pre, self.C.byref(ci), self.C.c_void_p(ps.data_ptr()), P, *post, self.C.byref(co),
                  self.C.c_void_p(ops.data_ptr()))
        return self._trim(o, co), ops

    def fork_forward(self, r, ps, k, min_err, twin):
        return self._with_ps("rfx_dev_fork_filter", self.rfx.L.rfx_dev_fork_filter, r, ps, (0,), (k, min_err, twin))

    def fork_reflected(self, r, ps, k, min_err, twin):
        return self._with_ps("rfx_dev_fork_filter", self.rfx.L.rfx_dev_fork_filter, r, ps, (1,), (k, min_err, twin))

    def reflect(self, r, k):
        o = self._empty(r.marker.device, r.n, r.words, r.kw)
        ci, co = self._c(r), self._c(o)
        self._run("rfx_dev_reflect_from_forward", self.rfx.L.rfx_dev_reflect_from_forward, self.C.byref(ci), k, self.C.byref(co))
        return self._trim(o, co)

    def random_reflection(self, r, ps, k):
        P = int(ps.numel()) - 1
        o = self._empty(r.marker.device, r.n, r.words, r.kw)
        ci, co = self._c(r), self._c(o)
        self._run("rfx_dev_random_reflection", self.rfx.L.rfx_dev_random_reflection, self.C.byref(ci),
                  self.C.c_void_p(ps.data_ptr()), P, k, self.C.byref(co))
        return self._trim(o, co)

    def extend_pass(self, r, ps, k, twin, stage):
        return self._with_ps("rfx_dev_extend_pass", self.rfx.L.rfx_dev_extend_pass, r, ps, (), (k, twin, stage, 2))

    def lower_bound(self, sorted_keys, values, upper):
        out = torch.empty(int(values.numel()), dtype=torch.int64, device=sorted_keys.device)
        self._run("rfx_dev_lower_bound", self.rfx.L.rfx_dev_lower_bound, self.C.c_void_p(sorted_keys.data_ptr()),
                  self.C.c_int64(int(sorted_keys.numel())), self.C.c_void_p(values.data_ptr()), self.C.c_int64(int(values.numel())),
                  1 if upper else 0, self.C.c_void_p(out.data_ptr()))
        return out

    def contigs_text(self, r: TRecs, k, min_contig, twin):
        from .api import Records
        h = Records(r.key.cpu().numpy().view(np.uint64), r.marker.cpu().numpy(), r.ext_off.cpu().numpy(),
                    r.ext.cpu().numpy().view(np.uint64), r.left.cpu().numpy(), r.right.cpu().numpy())
        return self.rfx.KmerToContig(h, k, min_contig, twin)


def _allreduce_t(t: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized():
        dist.all_reduce(t, group=group)
    return t


def splitters_t(ops, sorted_keys: torch.Tensor, P: int, key_bits: int, group=None):
    """splitters() on a device tensor of locally sorted one-word keys: eight key bits per round -- 255 candidate
    values per boundary, their local positions by ops.lower_bound, one all-reduce of the counts.  -> (v uint64[P-1],
    incl bool[P-1], global count)"""
    dev = sorted_keys.device
    n_glob = int(_allreduce_t(torch.tensor([int(sorted_keys.numel())], dtype=torch.int64, device=dev), group).item())
    q = np.array([(p * n_glob) // P for p in range(1, P)], np.int64)
    v = np.zeros(P - 1, np.uint64)
    if n_glob == 0 or P == 1:
        return v, np.ones(P - 1, bool), n_glob
    levels = (key_bits + 7) // 8
    steps = np.arange(1, 256, dtype=np.uint64)
    for lvl in range(levels - 1, -1, -1):
        shift = np.uint64(8 * lvl)
        cand = (v[:, None] | (steps[None, :] << shift)).reshape(-1)                 # ascending per boundary
        ct = torch.from_numpy(cand.view(np.int64).copy()).to(dev)
        cnt = _allreduce_t(ops.lower_bound(sorted_keys, ct, False), group).cpu().numpy().reshape(P - 1, 255)
        sel = (cnt <= q[:, None]).sum(axis=1).astype(np.uint64)                      # the largest candidate that still fits
        v = v | (sel << shift)
    vt = torch.from_numpy(v.view(np.int64).copy()).to(dev)
    c_lt = _allreduce_t(ops.lower_bound(sorted_keys, vt, False), group).cpu().numpy()
    return v, c_lt == q, n_glob


def _local_bounds_t(ops, sorted_keys, v, incl):
    dev = sorted_keys.device
    vt = torch.from_numpy(v.view(np.int64).copy()).to(dev)
    lo = ops.lower_bound(sorted_keys, vt, False).cpu().numpy()
    hi = ops.lower_bound(sorted_keys, vt, True).cpu().numpy()
    return np.where(incl, lo, hi).astype(np.int64)


def _exchange_t(t: torch.Tensor, cuts, group=None):
    """slices [cuts[d], cuts[d+1]) of t to rank d -> what every rank sent here, in rank order (one tensor)"""
    counts = [int(cuts[d + 1] - cuts[d]) for d in range(len(cuts) - 1)]
    recv, _ = _alltoallv(t[int(cuts[0]):int(cuts[-1])].contiguous(), counts, group)
    return recv


def sort_exchange_t(ops, r: TRecs, P: int, k: int, group=None, force_exchange: bool = False):
    """sort_exchange() on device-resident records (one-word keys) -> (this rank's logical partitions sorted by key,
    their partition starts as a device tensor)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert P % world == 0, "the logical partition count must be a multiple of the number of ranks"
    assert r.kw == 1, "the range shuffle handles one-word keys (k <= 32)"
    per = P // world
    dev = r.marker.device
    r = ops.sort(r, k)
    v, incl, _ = splitters_t(ops, r.key[:r.n], P, 2 * (k - 1), group)
    b = np.concatenate([[0], _local_bounds_t(ops, r.key[:r.n], v, incl), [r.n]]).astype(np.int64)
    if world == 1 and not force_exchange:
        return r, torch.from_numpy(b).to(dev)
    rcut = b[::per]
    wcut = r.ext_off[torch.from_numpy(rcut).to(dev)].cpu().numpy()
    lens = (r.ext_off[1:r.n + 1] - r.ext_off[:r.n])
    key = _exchange_t(r.key, rcut, group)
    marker = _exchange_t(r.marker, rcut, group)
    left = _exchange_t(r.left, rcut, group)
    right = _exchange_t(r.right, rcut, group)
    lens = _exchange_t(lens, rcut, group)
    ext = _exchange_t(r.ext, wcut, group)
    ext_off = torch.zeros(int(lens.numel()) + 1, dtype=torch.int64, device=dev)
    if lens.numel():
        ext_off[1:] = torch.cumsum(lens, 0)
    merged = ops.sort(ops.make(key, marker, ext_off, ext, left, right), k)
    p0 = rank * per
    full = np.concatenate([[0], _local_bounds_t(ops, merged.key[:merged.n], v, incl), [merged.n]]).astype(np.int64)
    ps = full[p0:p0 + per + 1].copy()
    ps[0], ps[-1] = 0, merged.n
    return merged, torch.from_numpy(ps).to(dev)


def sharded_assemble_dev(ops, keys: torch.Tensor, counts: torch.Tensor, prm, group=None, root: int = 0, trace=None,
                         force_exchange: bool = False):
    """sharded_assemble() with the records resident on the ranks' devices.  keys (int64 bit patterns) / counts (int32):
    this rank's shard of the filtered (k-mer, count) list, any order, on its device.  k <= 31.  Returns (contig text,
    n_contigs) on `root`, (None, None) elsewhere; identical to the single-GPU driver for the same prm.partitions."""
    rfx = getattr(ops, "rfx", None)
    if rfx is not None and getattr(rfx, "comm", None):
        # the engine's context carries an RCCL communicator: the whole driver runs behind the C ABI
        # (rfx_dev_sharded_assemble, reflexiv_amd/csrc/rfx_shard.hip); this module is only its caller
        torch.cuda.current_stream().synchronize()
        k_, c_ = keys.contiguous(), counts.contiguous()
        text, nc, tr = rfx.sharded_assemble_dev(k_.data_ptr(), c_.data_ptr(), int(c_.numel()), prm, gather_below=0 if force_exchange else -1)
        if trace is not None:
            trace.extend(tr)
        return (text, nc) if rfx.comm_rank == root == 0 else (None, None)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    k, twin, P = prm.k, prm.twin, max(1, prm.partitions)
    dev = keys.device
    exchange = world > 1 or force_exchange
    # the count stage's order contract: ascending canonical k-mer -> range-shard the survivors first
    keys, counts = ops.sort_pairs(keys, counts, k)
    if exchange:
        v, incl, _ = splitters_t(ops, keys, world, 2 * k, group)
        cut = np.concatenate([[0], _local_bounds_t(ops, keys, v, incl), [int(keys.numel())]]).astype(np.int64)
        keys, counts = _exchange_t(keys, cut, group), _exchange_t(counts, cut, group)
        keys, counts = ops.sort_pairs(keys, counts, k)
    r = ops.rc_expand(keys, counts, k)
    r, ps = sort_exchange_t(ops, r, P, k, group, force_exchange)
    r, ps = ops.fork_forward(r, ps, k, prm.min_error_cov, twin)
    r = ops.reflect(r, k)
    r, ps = sort_exchange_t(ops, r, P, k, group, force_exchange)
    r, ps = ops.fork_reflected(r, ps, k, prm.min_error_cov, twin)
    r = ops.random_reflection(r, ps, k)

    def count(rr):
        return int(_allreduce_t(torch.tensor([rr.n], dtype=torch.int64, device=dev), group).item())

    def one_pass(rr, stage):
        rr, pst = sort_exchange_t(ops, rr, P, k, group, force_exchange)
        rr, _ = ops.extend_pass(rr, pst, k, twin, stage)
        if trace is not None:
            trace.append(count(rr))
        return rr
    r = one_pass(r, 0)
    it = 0
    for _ in range(3):
        it += 1
        r = one_pass(r, 0)
    it += 1
    r = one_pass(r, 1)
    last = 0
    while it <= prm.max_iter:
        it += 1
        if it >= prm.min_iter and it % 3 == 0:
            c = count(r)
            if c == last:
                break
            last = c
        r = one_pass(r, 2)
    # the surviving records are few: contig ids follow the global record order, so the text is made on root
    if world > 1:
        lens = (r.ext_off[1:r.n + 1] - r.ext_off[:r.n])

        def to_root(t, n):
            cuts = np.array([0] * (root + 1) + [n] * (world - root), np.int64)
            return _exchange_t(t, cuts, group)
        key, marker = to_root(r.key, r.n), to_root(r.marker, r.n)
        left, right = to_root(r.left, r.n), to_root(r.right, r.n)
        lens, ext = to_root(lens, r.n), to_root(r.ext, r.words)
        if rank != root:
            return None, None
        off = torch.zeros(int(lens.numel()) + 1, dtype=torch.int64, device=dev)
        if lens.numel():
            off[1:] = torch.cumsum(lens, 0)
        r = ops.make(key, marker, off, ext, left, right)
    return ops.contigs_text(r, k, prm.min_contig, twin)
