"""Where a multi-GPU count step spends its time, rehearsed on ONE rank (the exchange degenerates to a
device copy): wall time of every engine call and of the exchange, synchronised, for both exchange forms.
    python tools/prof_dist.py [--gbp 5] [--k 31] [--chunks 4] [--steps 3]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import reflexiv_amd
from reflexiv_amd import dist as rd

ap = argparse.ArgumentParser()
ap.add_argument("--gbp", type=float, default=5.0); ap.add_argument("--k", type=int, default=31)
ap.add_argument("--chunks", type=int, default=4); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--genome", type=int, default=4_640_000); ap.add_argument("--cover", type=int, default=30)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rfx = reflexiv_amd.Reflexiv(0)
rfx.use_stream(torch.cuda.current_stream().cuda_stream)
L = 150; wpr = (L + 31) // 32
n_reads = int(round(a.gbp * 1e9 / L)); n_reads += n_reads & 1
dg = torch.empty((a.genome + 31) // 32, dtype=torch.int64, device="cuda")
dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, a.genome, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), a.genome, 0, n_reads, L, wpr, dw.data_ptr())
rfx.sync()
reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=a.k)


def timed(obj, name, acc):
    f = getattr(obj, name)
    def g(*x, **kw):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*x, **kw)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    setattr(obj, name, g)


_xa = rd.exchange_by_owner_async
xacc = {}
def _timed_exchange(*x, **kw):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = _xa(*x, **kw)
    xacc["launch"] = xacc.get("launch", 0.0) + (time.perf_counter() - t) * 1e3
    torch.cuda.synchronize()
    xacc["exchange"] = xacc.get("exchange", 0.0) + (time.perf_counter() - t) * 1e3
    return r
rd.exchange_by_owner_async = _timed_exchange

for combine in (False, True, "gen"):
    eng = rd.HipEngine(rfx, combine=combine is True); eng.force_exchange = True
    acc = {}
    timed(eng, "bucket_by_owner", acc); timed(eng, "count_kmers", acc)
    for name in ("combine_reads_dev", "bucket_pairs_by_owner_dev", "merge_pairs_dev", "bucket_records_by_owner_dev", "count_records_dev"):
        timed(rfx, name, acc)
    for step in range(a.steps):
        acc.clear(); xacc.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keys, counts, tot = rd.sharded_count(eng, reads, a.cover, 10_000_000, 0, chunks=a.chunks, generations=4 if combine == 'gen' else 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(f"{'rec-gen' if combine == 'gen' else 'pairs  ' if combine else 'records'} step {step}: {dt:7.1f} ms  tot={tot}  " +
              "  ".join(f"{k}={v:.1f}" for k, v in sorted(list(acc.items()) + list(xacc.items()))), flush=True)
    del eng, keys, counts
    torch.cuda.empty_cache()
dist.destroy_process_group()
