#!/usr/bin/env python3
"""Count stage under low-complexity skew: every `--every`-th read of the synthetic set is replaced by
poly-A (all-zero packed words).  Prints per-kernel milliseconds with and without the skew."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbp", type=float, default=5.0)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--cover", type=int, default=30)
    a = ap.parse_args()
    import torch
    import reflexiv_amd
    rfx = reflexiv_amd.Reflexiv(0)
    L, k, wpr, G = 150, 31, 5, 4_640_000
    n_reads = int(a.gbp * 1e9 / L) // 2 * 2
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(1, G, dg.data_ptr())
    rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr())
    rfx.sync()
    N = rfx.kmers_per_read(L, k) * n_reads
    cap = max(1 << 20, N // 8)
    dk = torch.empty(cap, dtype=torch.int64, device="cuda")
    dc = torch.empty(cap, dtype=torch.int32, device="cuda")
    for label in ("uniform", f"poly-A every {a.every}th read"):
        if label != "uniform":
            dw.view(n_reads, wpr)[::a.every] = 0
        torch.cuda.synchronize()
        for _ in range(2):
            m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, a.cover)
        t = {n_: round(ms, 2) for n_, (ms, _) in rfx.count_timing().items()}
        print(label, "kept/distinct", (m, nd), t, "total", round(sum(t.values()), 2), flush=True)


if __name__ == "__main__":
    main()
