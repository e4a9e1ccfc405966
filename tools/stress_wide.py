"""stress: k = 33..63 record path vs element path, many sizes / k in a row (intermittent failures of the
lock-and-publish table protocol would show as a differing count)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
L = 150; wpr = 5
bad = 0
it = 0
for rep in range(6):
    for k, n_reads, G in ((63, 400_000, 300_000), (47, 900_000, 200_000), (33, 250_000, 100_000), (63, 2_000_000, 1_000_000), (55, 60_000, 20_000)):
        it += 1
        dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        rfx.synth_genome_dev(100 + it, G, dg.data_ptr()); rfx.synth_reads_dev(100 + it, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
        N = rfx.kmers_per_read_w(L, k) * n_reads
        res = []
        for flag in ("0", "1"):
            os.environ["RFX_WIDE_RECORDS"] = flag
            dk = torch.empty(2 * N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            m, d, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 1)
            res.append((m, d, dk[:2 * m].clone(), dc[:m].clone()))
            del dk, dc
        ok = res[0][0] == res[1][0] and res[0][1] == res[1][1] and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3]) \
            and int(res[0][3].sum()) == N
        bad += not ok
        print(f"it {it} k={k} reads={n_reads}: distinct {res[0][1]} / {res[1][1]}  {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
