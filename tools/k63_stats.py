import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
n_reads, L, G = 33333334, 150, 4640000
wpr = 5
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
for k in (63, 47, 33):
    cap = 1 << 24
    dk = torch.empty(cap * 2, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for it in range(3):
        t = time.perf_counter()
        m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 30)
        dt = time.perf_counter() - t
    st = rfx.count_timing()
    print(k, "ms", round(dt * 1e3, 2), {a: (round(b[0], 2), b[1]) for a, b in st.items()}, "inst", inst, "distinct", nd, flush=True)
