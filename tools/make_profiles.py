#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/refresh_profiles.sh on the GPU box) into the
tracked summaries under profiles/: <tag>_bench.json, <tag>_bench_kernel_stats.csv,
<tag>_hbm_pmc.csv, <tag>_hbm_traffic.json.  FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled
for gfx950 as MI355X_MICROARCH.md prescribes."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (level 1 is one sweep by default: k_sk_sample_hist + k_plan_regions, then k_sk_onesweep + k_fix_holes; the two-pass
# kernels k_sk_hist / k_sk_scatter appear when it falls back)
FAMILY = [("hist1", ("k_sk_sample_hist", "k_plan_regions", "k_sk_hist")), ("part1", ("k_sk_onesweep", "k_fix_holes", "k_sk_scatter")),
          ("hist2", ("k_l2_sample", "k_l2_caps", "k_rec_hist")), ("part2", ("k_rec_l2sweep", "k_l2_check", "k_rec_scatter")),
          ("leaf", ("k_leaf_count",))]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    # kernel stats of the bench command
    # (gpurun merges new files into gpurun_out/ without deleting older runs' files: take the newest)
    stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    assert stats, "no kernel_stats.csv"
    rows = list(csv.DictReader(open(stats[-1])))
    with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    # PMC passes
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    pmc_rows = []
    for which in ("fetch", "write"):
        for fn in sorted(glob.glob(os.path.join(src, which, "**", "*counter_collection.csv"), recursive=True),
                         key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(fn)):
                name = r["Kernel_Name"]
                for fam, key in FAMILY:
                    if any(x in name for x in key):
                        per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                        launches[fam][r["Counter_Name"]] += 1
                        pmc_rows.append({"family": fam, "kernel": name[:80], "counter": r["Counter_Name"],
                                         "value_KiB": r["Counter_Value"]})
    with open(os.path.join(dst, tag + "_hbm_pmc.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["family", "kernel", "counter", "value_KiB"])
        w.writeheader()
        w.writerows(pmc_rows)
    n_inst = bench["config"]["kmer_instances_per_gpu"]
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 "
                   "--warmup 0 --no-cpu-baseline --no-contigs` on one MI355X (tools/refresh_profiles.sh); counters are KiB; "
                   "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                   "WRITE_SIZE as read; bytes of the family over the one step the pass runs (`hbm_bytes_per_launch`: a step launches every family once, the leaf kernel a second time when there are heavy leaves)",
           "kmer_instances": n_inst, "kernels": {}}
    for fam, key in FAMILY:
        if fam not in per:
            continue
        # the passes run ONE step: the family's bytes per step (a step may launch the family more than once --
        # heavy leaves go through a second launch of the leaf kernel), as bench.py prices its time per step
        nl = max(1, launches[fam].get("FETCH_SIZE", 1))
        fetch = per[fam].get("FETCH_SIZE", 0.0) * 1024 * 2
        write = per[fam].get("WRITE_SIZE", 0.0) * 1024
        out["kernels"][fam] = {"kernel": key, "launches_per_step": nl, "fetch_bytes_corrected": fetch, "write_bytes": write,
                               "hbm_bytes_per_launch": fetch + write}
    json.dump(out, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
    # the bench line, with roofline.traffic taken from the PMC passes
    rf = bench.get("roofline") or {}
    fam = rf.get("kernel")
    if fam in out["kernels"]:
        rf["traffic"] = out["kernels"][fam]["hbm_bytes_per_launch"]
    json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in out["kernels"].items()}))
    print("bench:", bench["value"], bench["ms_per_step"], rf)


if __name__ == "__main__":
    main()
