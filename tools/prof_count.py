#!/usr/bin/env python3
"""Profiling driver: one count stage (and optionally assembly) on synthetic reads, sized by
--gbp, for `rocprofv3 --kernel-trace --stats` / `--pmc` runs (profiles/ holds the summaries)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbp", type=float, default=1.0)
    ap.add_argument("--genome", type=int, default=4_640_000)
    ap.add_argument("--cover", type=int, default=30)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--assemble", action="store_true")
    ap.add_argument("--assemble-runs", type=int, default=1)
    a = ap.parse_args()
    import torch
    import reflexiv_amd
    rfx = reflexiv_amd.Reflexiv(0)
    L, k = 150, 31
    wpr = 5
    n_reads = int(a.gbp * 1e9 / L) // 2 * 2
    dg = torch.empty((a.genome + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(1, a.genome, dg.data_ptr())
    rfx.synth_reads_dev(1, dg.data_ptr(), a.genome, 0, n_reads, L, wpr, dw.data_ptr())
    rfx.sync()
    N = rfx.kmers_per_read(L, k) * n_reads
    cap = max(1 << 20, N // 8)
    dk = torch.empty(cap, dtype=torch.int64, device="cuda")
    dc = torch.empty(cap, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(a.steps):
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, a.cover)
    print("instances", inst, "distinct", nd, "kept", m, rfx.count_timing())
    if a.assemble:
        prm = reflexiv_amd.default_params(min_cov=a.cover, partitions=8)
        for _ in range(a.assemble_runs):      # (the first run of a process loads the extend kernels: RFX_TRACE of the second is the warm one)
            text, nc, trace = rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), m, prm)
        print("contigs", nc, "passes", len(trace))


if __name__ == "__main__":
    main()
