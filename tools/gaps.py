#!/usr/bin/env python3
"""Timeline of the LAST count step in a rocprofv3 kernel trace (tools/prof_cnt.sh): every kernel's start, duration and
the idle gap before it.    python tools/gaps.py [gpurun_out/prof_cnt]"""
import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_cnt"
start_at = sys.argv[2].split(",") if len(sys.argv) > 2 else ["k_sk_sample_hist", "k_sk_hist"]      # e.g. k_rc_expand: the extend stage
thr_gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 20000
f = max(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = [i for i, n in enumerate(names) if any(x in n for x in start_at)]
i0 = first[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0; gap_sum = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev); gap_sum += gap
    nm = r["Kernel_Name"]
    nm = nm[nm.find("k_"):].split("(")[0][:36] if "k_" in nm else nm[:36]
    if gap > thr_gap or e - s > 100000:
        print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f}  gap {gap / 1e3:7.1f} us  {nm}")
    prev = max(prev, e)
print(f"kernels busy {(prev - t0 - gap_sum) / 1e6:.3f} ms, gaps {gap_sum / 1e6:.3f} ms, span {(prev - t0) / 1e6:.3f} ms")
