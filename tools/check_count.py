#!/usr/bin/env python3
"""Debug helper: fused GPU count vs the oracle on synthetic reads (sizes the oracle finishes in ~10 s)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--genome", type=int, default=4_640_000)
    ap.add_argument("--cover", type=int, default=3)
    a = ap.parse_args()
    import torch
    import reflexiv_amd
    from oracle import oracle as O
    rfx = reflexiv_amd.Reflexiv(0)
    L, k, wpr = 150, 31, 5
    dg = torch.empty((a.genome + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(a.reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(1, a.genome, dg.data_ptr())
    rfx.synth_reads_dev(1, dg.data_ptr(), a.genome, 0, a.reads, L, wpr, dw.data_ptr())
    rfx.sync()
    N = 120 * a.reads
    dk = torch.empty(N, dtype=torch.int64, device="cuda"); dc = torch.empty(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), a.reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), N, a.cover)
    g = O.synth_genome(1, a.genome)
    bases, off = O.synth_reads(1, g, a.genome, 0, a.reads, L)
    km = O.extract_canon(bases, off, k)
    wk, wc, wd = O.count_filter(km, a.cover)
    gk = dk[:m].cpu().numpy().view(np.uint64); gc = dc[:m].cpu().numpy()
    print("gpu", m, nd, "oracle", len(wk), wd, "equal", m == len(wk) and np.array_equal(gk, wk) and np.array_equal(gc, wc))
    if m != len(wk) or not np.array_equal(gk, wk):
        extra = np.setdiff1d(gk, wk); missing = np.setdiff1d(wk, gk)
        print("extra", len(extra), "missing", len(missing), "dups in gpu", len(gk) - len(np.unique(gk)))
        both, ia, ib = np.intersect1d(gk, wk, return_indices=True)
        bad = gc[ia] != wc[ib]
        print("count mismatches among common", int(bad.sum()), list(zip(gc[ia][bad][:8], wc[ib][bad][:8])))


if __name__ == "__main__":
    main()
