#!/usr/bin/env python3
"""Interleaved A/B of count-stage variants in ONE process on ONE device (env overrides are read
per call by the library): prints the median per-kernel milliseconds of each variant.

    python tools/ab_count.py --gbp 5 --rounds 3 "RFX_LEVEL_BITS=9,10" "RFX_LEVEL_BITS=9,10 RFX_TPB=8"
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--gbp", type=float, default=5.0)
    ap.add_argument("--genome", type=int, default=4_640_000)
    ap.add_argument("--cover", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch
    import reflexiv_amd
    rfx = reflexiv_amd.Reflexiv(0)
    L, k, wpr = 150, 31, 5
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
    keys = set()
    res = {v: [] for v in a.variants}
    check = {}
    managed = set()
    for v in a.variants:
        for kv in v.split():
            managed.add(kv.split("=")[0])
    for r in range(a.rounds + 1):
        for v in a.variants:
            for name in managed:
                os.environ.pop(name, None)
            for kv in v.split():
                if "=" in kv:
                    n_, val = kv.split("=", 1)
                    os.environ[n_] = val
            m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, a.cover)
            check.setdefault((m, nd), []).append(v)
            if r > 0:                      # round 0 = warm-up
                t = {n_: ms for n_, (ms, _) in rfx.count_timing().items()}
                t["total"] = sum(t.values())
                res[v].append(t)
                keys.update(t)
    if len(check) != 1:
        print("NOTE: variants disagree (expected for ablation flags):", {k_: sorted(set(v_)) for k_, v_ in check.items()})
    order = [k_ for k_ in ("hist1", "part1", "hist2", "part2", "hist3", "part3", "leaf", "sort", "total") if k_ in keys]
    print("instances", N, "kept/distinct", list(check)[0])
    print("variant".ljust(44), " ".join(k_.rjust(7) for k_ in order))
    for v in a.variants:
        print((v or "(default)").ljust(44), " ".join(f"{statistics.median(t.get(k_, 0.0) for t in res[v]):7.2f}" for k_ in order))


if __name__ == "__main__":
    main()
