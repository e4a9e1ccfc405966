#!/usr/bin/env python3
"""gpurun_out/<tag>_SQ_*/**/counter_collection.csv (tools/pmc_leaf.sh <tag>) -> per-kernel SQ counter table on stdout
(every kernel of the count stage, not only the leaf): python tools/sq_all.py <tag> [instances]"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
inst = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000080
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_SQ_*"))):
    if not os.path.isdir(d):
        continue
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])):
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        kn = re.sub(r"\(.*", "", kn)
        tot[kn][r["Counter_Name"]] += float(r["Counter_Value"])
for kn, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if t.get("SQ_INSTS_VALU", 0) * 64 / inst < 0.5:
        continue
    print(f"== {kn}")
    per = lambda c: t.get(c, 0) * 64 / inst
    print(f"   per k-mer slot: VALU {per('SQ_INSTS_VALU'):.1f}  SALU {per('SQ_INSTS_SALU'):.1f}  LDS {per('SQ_INSTS_LDS'):.2f}  VMEM_RD {per('SQ_INSTS_VMEM_RD'):.3f}")
    if t.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS bank-conflict / active = {t['SQ_LDS_BANK_CONFLICT'] / t['SQ_LDS_IDX_ACTIVE']:.3f}; LDS active cycles / busy cycles = {t['SQ_LDS_IDX_ACTIVE'] / max(1, t.get('SQ_BUSY_CYCLES', 0)):.3f}")
    if t.get("SQ_WAVE_CYCLES"):
        print(f"   WAIT_ANY / WAVE_CYCLES = {t.get('SQ_WAIT_ANY', 0) / t['SQ_WAVE_CYCLES']:.3f}; WAIT_INST_LDS / WAVE_CYCLES = {t.get('SQ_WAIT_INST_LDS', 0) / t['SQ_WAVE_CYCLES']:.3f}; "
              f"ACTIVE_INST_VALU / BUSY = {t.get('SQ_ACTIVE_INST_VALU', 0) / max(1, t.get('SQ_BUSY_CYCLES', 0)):.3f}; waves {int(t.get('SQ_WAVES', 0))}; busy cycles {int(t.get('SQ_BUSY_CYCLES', 0))}")
