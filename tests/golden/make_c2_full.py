#!/usr/bin/env python3
"""Full-size parity pin for BASELINE config 2 (and the same reads at k = 63): runs the CPU ORACLE on the exact
workload bench.py times -- seed 1, 33,333,334 PE150 reads from the 4.64 Mbp genome, 0.5 % substitutions,
-cover 30, 8 logical partitions -- and writes tests/golden/c2_full.json: instance / distinct / kept counts, sha256
of the survivor list (keys and counts), the extend trace, sha256 and summary of the contig text.
tests/test_gpu_full_size.py runs the HIP path at the same size and compares.

The 4.0e9 (2.9e9 two-word) instances do not fit in this container's memory at once, so the k-mer space is counted in
passes over range buckets (oracle.count_reads_omp(buckets=...)); the passes' outputs concatenate to the full ascending
list.  Run here (8 cores, ~64 GB): about 20 minutes.   python tests/golden/make_c2_full.py [--reads N] [--out FILE]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O      # noqa: E402


def count_in_passes(bases, off, k, cover, n_pass, cap=1 << 24, keep=True):
    ks, cs, nd, ni, nkept = [], [], 0, 0, 0
    hk, hc = hashlib.sha256(), hashlib.sha256()
    cuts = [4096 * i // n_pass for i in range(n_pass + 1)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        t = time.time()
        a, b, d, i = O.count_reads_omp(bases, off, k, cover, buckets=(lo, hi), cap=cap)
        hk.update(a.tobytes()); hc.update(b.tobytes())
        nkept += len(b)
        if keep:
            ks.append(a); cs.append(b)
        nd += d; ni += i
        print(f"  k={k} buckets [{lo},{hi}): {i} instances, {d} distinct, {len(b)} kept, {time.time() - t:.0f} s", flush=True)
    if not keep:                                          # (--count-only: the survivors are hashed pass by pass and dropped)
        return None, nkept, nd, ni, hk.hexdigest(), hc.hexdigest()
    return np.concatenate(ks), np.concatenate(cs), nd, ni, hk.hexdigest(), hc.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=33_333_334)
    ap.add_argument("--genome", type=int, default=4_640_000)
    ap.add_argument("--cover", type=int, default=30)
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--ks", default="31,63")
    ap.add_argument("--out", default=os.path.join(HERE, "c2_full.json"))
    ap.add_argument("--cap", type=int, default=1 << 24, help="survivors per pass")
    ap.add_argument("--count-only", action="store_true",
                    help="pin the count stage only (config 5's per-GPU share: 4.5e8 survivors at -cover 2 are 9e8 records, beyond "
                         "what the oracle's extend stage holds in this container's memory)")
    args = ap.parse_args()
    seed, L = 1, 150
    O.set_threads(O.host_cores())
    t0 = time.time()
    g = O.synth_genome(seed, args.genome)
    bases, off = O.synth_reads(seed, g, args.genome, 0, args.reads, L)
    print(f"reads: {args.reads} x {L} in {time.time() - t0:.0f} s", flush=True)
    out = {"workload": {"seed": seed, "genome": args.genome, "reads": args.reads, "read_len": L, "cover": args.cover,
                        "partitions": args.partitions, "err_per_2_32": 21474836},
           "made_by": "tests/golden/make_c2_full.py (oracle/reflexiv_oracle.c, threaded form)"}
    if os.path.exists(args.out):                         # --ks 63 alone refreshes that record and keeps the others
        old = json.load(open(args.out))
        if old.get("workload") == out["workload"]:
            out.update({k: v for k, v in old.items() if k.startswith("k")})
    for k in [int(x) for x in args.ks.split(",")]:
        keys, counts, nd, ni, hk, hc = count_in_passes(bases, off, k, args.cover, args.passes, args.cap, not args.count_only)
        rec = {"n_instances": ni, "n_distinct": nd, "n_kept": int(counts if args.count_only else len(counts)), "sha256_keys": hk, "sha256_counts": hc}
        if args.count_only:
            out[f"k{k}"] = rec
            json.dump(out, open(args.out, "w"), indent=1)
            continue
        t = time.time()
        if k <= 31:
            prm = O.default_params(k=k, min_cov=args.cover, partitions=args.partitions)
            text, nc, trace, _ = O.assemble_from_counts(keys, counts, prm)
        else:
            prm = O.default_params(k=k, min_cov=args.cover, partitions=args.partitions)
            text, nc, trace, _ = O.assemble_from_counts(O.counter_to_asm_w(keys, k), counts.astype(np.int32), prm)
        lens = sorted((int(h.split("-")[1]) for h in text.split("\n") if h.startswith(">")), reverse=True)
        rec.update({"extras": int(prm.extras) if k > 31 else None, "trace": trace, "n_contigs": nc, "sha256_contig_text": hashlib.sha256(text.encode()).hexdigest(),
                    "contig_lengths": lens, "contig_text_bytes": len(text)})
        print(f"  k={k}: assembled in {time.time() - t:.0f} s: {nc} contigs {lens[:4]}, {len(trace)} passes", flush=True)
        out[f"k{k}"] = rec
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out, f"({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
