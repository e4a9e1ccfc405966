"""k = 63 count with level-bit plans and pre-split thresholds, one process (for tools/ab_k63.sh: one run per build)"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
n_reads, L, G, wpr = 33333334, 150, 4640000, 5
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
cap = 1 << 24
dk = torch.empty(cap * 2, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for env in ({},):
    old = {a: os.environ.get(a) for a in env}
    os.environ.update(env)
    best = None
    for _ in range(3):
        m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, 63, dk.data_ptr(), dc.data_ptr(), cap, 30)
        st = rfx.count_timing()
        tot = sum(v[0] for a, v in st.items() if not a.startswith("stat_"))
        best = tot if best is None else min(best, tot)
    print(env, "kernels", round(best, 2), {a: round(b[0], 2) for a, b in st.items() if not a.startswith("stat_")}, "kept", m, "distinct", nd,
          "passes", st["stat_passes"][1], "overflows", st["stat_overflows"][1], flush=True)
    for a, v in old.items():
        if v is None: os.environ.pop(a, None)
        else: os.environ[a] = v
