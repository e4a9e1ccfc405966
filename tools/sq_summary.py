#!/usr/bin/env python3
"""gpurun_out/<tag>_SQ_*/**/counter_collection.csv (written by tools/pmc_leaf.sh <tag>) -> profiles/<round>_sq_counters.txt:
the SQ counters of the leaf kernel summed over its launches, per k-mer instance.
    python tools/sq_summary.py r02pmc r02 [instances]"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
inst = int(sys.argv[3]) if len(sys.argv) > 3 else 4000000080
tot = collections.defaultdict(float)
kname = None
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_SQ_*"))):
    if not os.path.isdir(d):
        continue
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])):
        if "k_leaf_count" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            kname = re.search(r"k_leaf_count<[^>]*>", r["Kernel_Name"]).group(0)
lines = [f"SQ counters of {kname} (rocprofv3 --pmc, three passes of `python3 tools/prof_count.py --gbp 5 --steps 1`; "
         f"tools/pmc_leaf.sh {tag}; tools/sq_summary.py), round {rnd[1:].lstrip('0')}", f"k-mer instances: {inst}", ""]
for k in sorted(tot):
    lines.append(f"{k:<28}{int(tot[k]):>18}   per k-mer instance {tot[k] / inst:8.3f}")
lines.append("")
if tot.get("SQ_LDS_IDX_ACTIVE"):
    lines.append(f"LDS bank-conflict cycles / LDS active cycles = {tot['SQ_LDS_BANK_CONFLICT'] / tot['SQ_LDS_IDX_ACTIVE']:.3f}")
lines.append(f"wave-level VALU instructions per k-mer slot (x64 lanes / instances) = {tot['SQ_INSTS_VALU'] * 64 / inst:.1f}")
lines.append(f"wave-level SALU per k-mer slot = {tot['SQ_INSTS_SALU'] * 64 / inst:.1f}; LDS = {tot['SQ_INSTS_LDS'] * 64 / inst:.1f}")
open(os.path.join(ROOT, "profiles", rnd + "_sq_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
