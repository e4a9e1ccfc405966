"""debug: combine output vs the fused count (min_cov 1) with a garbage-filled scratch buffer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
L, k = 150, 31
wpr = (L + 31) // 32
for n_reads, G in ((24_000, 40_000), (100_000, 200_000), (1_000_000, 1_000_000), (300_000, 200_000), (2_000_000, 500_000)) * 12:
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(7, G, dg.data_ptr()); rfx.synth_reads_dev(7, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
    N = rfx.kmers_per_read(L, k) * n_reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m0, nd, _ = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, 1)
    cap = N + (9 << 20)
    scratch = torch.full((2 * cap,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
    out = torch.empty(2 * cap, dtype=torch.int64, device="cuda"); doff = torch.empty(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    m, h, inst = rfx.combine_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, 1, scratch.data_ptr(), out.data_ptr(), cap, doff.data_ptr())
    sc = scratch.view(-1, 2)
    bad = (sc[:, 0] == 0x5A5A5A5A5A5A5A5A)
    ext = int(torch.nonzero(~bad).max().item()) + 1 if (~bad).any() else 0
    nbad = int(bad[:ext].sum().item())
    first = torch.nonzero(bad[:ext])[:5].flatten().tolist()
    print(f"n_reads={n_reads}: fused distinct={m0} combine pairs={m} extent~{ext} untouched-inside-extent={nbad} first={first}", flush=True)
    if nbad:
        idx = torch.nonzero(bad[:ext]).flatten()
        runs = torch.nonzero(idx[1:] != idx[:-1] + 1).flatten()
        print("  runs:", len(runs) + 1, "first run start/len:", int(idx[0]), int(runs[0]) + 1 if len(runs) else len(idx), " start%16384 =", int(idx[0]) % 16384)
    bp = out[:2 * m].view(-1, 2)
    keys, cnt = bp[:, 0], bp[:, 1]
    ks, order = torch.sort(keys)
    dup = torch.nonzero(ks[1:] == ks[:-1]).flatten()
    print("  duplicate keys:", len(dup), " sum of counts:", int(cnt.sum()), "instances:", N, " fused sum:", int(dc[:m0].sum()))
    if len(dup):
        for d in dup[:5].tolist():
            i, j = int(order[d]), int(order[d + 1])
            print("   key", hex(int(ks[d])), "counts", int(cnt[i]), int(cnt[j]), "positions", i, j)
    fk = dk[:m0]
    missing = ~torch.isin(fk, ks); extra = ~torch.isin(ks, fk)
    print("  missing from combine:", int(missing.sum()), " not in fused:", int(extra.sum()), [hex(int(x)) for x in ks[extra][:5]])
    # positions in scratch of a duplicated key
    if len(dup):
        key = ks[dup[0]]
        pos = torch.nonzero(sc[:, 0] == key).flatten().tolist()
        print("   scratch positions of the first duplicate:", pos, [p % 16384 for p in pos], [p // 16384 for p in pos])
