import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
import reflexiv_amd
from reflexiv_amd import dist as rd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29572")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rfx = reflexiv_amd.Reflexiv(0); rfx.use_stream(torch.cuda.current_stream().cuda_stream)
L = 150; wpr = 5; n_reads = 33333334; G = 4_640_000
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
reads = dict(words=dw, n_reads=n_reads, wpr=wpr, read_len=L, k=31)
for chunks in (1, 2, 4, 8):
    eng = rd.HipEngine(rfx, combine=True); eng.force_exchange = True
    for it in range(2):
        eng.bucketed_bytes = 0
        torch.cuda.synchronize(); t = time.perf_counter()
        rd.sharded_count(eng, reads, 30, 10_000_000, 0, chunks=chunks)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print(f"chunks={chunks}: pairs bytes {eng.bucketed_bytes/1e9:.2f} GB = {eng.bucketed_bytes/4e9:.2f} B/inst, step {dt:.1f} ms", flush=True)
    del eng; torch.cuda.empty_cache()
dist.destroy_process_group()
