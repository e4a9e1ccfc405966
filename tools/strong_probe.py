"""dev helper: the 50 Gbp read set (or --gbp) on one GPU through rfx_dev_sharded_count in G generations: per-kernel times and the
leaf statistics (leaves, table passes, overflows) -- what a rank of 8 meets per site at 10780x."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbp", type=float, default=50.0)
    ap.add_argument("--gens", default="8")
    ap.add_argument("--genome", type=int, default=4_640_000)
    ap.add_argument("--k", type=int, default=31)
    a = ap.parse_args()
    import torch
    import reflexiv_amd
    rfx = reflexiv_amd.Reflexiv(0)
    L, k = 150, a.k
    wpr = 5
    n_reads = int(round(a.gbp * 1e9 / L)) & ~1
    cover = max(2, int(round(30 * a.gbp / 5.0 * 4_640_000 / a.genome)))
    dg = torch.empty((a.genome + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    cap = max(1 << 24, 2 * a.genome)
    wide = k > 32
    dk = torch.empty(cap * (2 if wide else 1), dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64 if wide else torch.int32, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(1, a.genome, dg.data_ptr())
    rfx.synth_reads_dev(1, dg.data_ptr(), a.genome, 0, n_reads, L, wpr, dw.data_ptr())
    rfx.sync()
    rfx.comm_init(reflexiv_amd.Reflexiv.comm_unique_id(), 0, 1)
    for G in [int(x) for x in a.gens.split(",")]:
        for rep in range(2):
            t0 = time.perf_counter()
            m, tot = rfx.sharded_count_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, cover, generations=G)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        tm = {n: round(ms, 2) for n, (ms, ln) in rfx.count_timing().items()}
        print(f"G={G}: {dt * 1e3:.1f} ms, kept {m}, totals {tot}, {tm}", flush=True)
    rfx.close()


if __name__ == "__main__":
    main()
