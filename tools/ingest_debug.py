import sys, os, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
n_reads, L, G = 33333334, 150, 4640000
wpr = 5
dev = "cuda"
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device=dev); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
nuc = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
host = torch.empty(n_reads * L, dtype=torch.uint8).pin_memory()
wv = dw.view(n_reads, wpr)
step_r = max(1, (1 << 28) // L)
for a in range(0, n_reads, step_r):
    b = min(n_reads, a + step_r)
    idx = torch.arange(L, device=dev)
    w = wv[a:b][:, idx // 32]
    code = (w >> (62 - 2 * (idx % 32))) & 3
    host[a * L:b * L].copy_(nuc[code].reshape(-1), non_blocking=True)
    del w, code
torch.cuda.synchronize()
del dw
roff = (torch.arange(n_reads + 1, dtype=torch.int64) * L).pin_memory().numpy()
prm = reflexiv_amd.default_params(k=31, min_cov=30, partitions=8)
for it in range(4):
    t = time.perf_counter()
    text, nc, tr, kept = rfx.assemble_reads_ptr(host.data_ptr(), n_reads * L, roff, prm)
    print("call", it, round((time.perf_counter() - t) * 1e3, 1), "ms", {a: round(b[0], 2) for a, b in rfx.count_timing().items()}, flush=True)
