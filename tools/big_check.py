"""Scale check: the three multi-GPU forms on ONE rank against the fused count at --gbp (totals must agree)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import reflexiv_amd
from reflexiv_amd import dist as rd
ap = argparse.ArgumentParser(); ap.add_argument("--gbp", type=float, default=10.0); ap.add_argument("--k", type=int, default=31)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rfx = reflexiv_amd.Reflexiv(0); rfx.use_stream(torch.cuda.current_stream().cuda_stream)
L = 150; wpr = 5; n_reads = int(a.gbp * 1e9 / L) // 2 * 2; G = 4_640_000
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=a.k)
N = rfx.kmers_per_read(L, a.k) * n_reads
cap = N // 8
dk = torch.empty(cap, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, a.k, dk.data_ptr(), dc.data_ptr(), cap, 30)
print(f"fused: instances {inst} distinct {nd} kept {m}  {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
ref = (dk[:m].clone(), dc[:m].clone()); del dk, dc
for name, kw, skw in (("records x4 chunks", {}, dict(chunks=4)), ("pairs x4 chunks", dict(combine=True), dict(chunks=4)),
                      ("records x4 generations", {}, dict(generations=4))):
    eng = rd.HipEngine(rfx, **kw); eng.force_exchange = True
    torch.cuda.synchronize(); t = time.perf_counter()
    keys, counts, tot = rd.sharded_count(eng, reads, 30, 10_000_000, 0, **skw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    o = torch.argsort(keys)
    ok = tot == [inst, nd, m] and torch.equal(keys[o], ref[0]) and torch.equal(counts[o], ref[1])
    print(f"{name}: tot {tot}  {dt:.1f} ms  {'OK' if ok else 'MISMATCH'}", flush=True)
    del eng, keys, counts; torch.cuda.empty_cache()
dist.destroy_process_group()
