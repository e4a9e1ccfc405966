"""dev helper / bench block: the two "next" rows of SURVEY.md 8(f) at scale.
f-4 contig RC de-duplication on N random contigs + their reverse complements, shuffled (rfx_dedup_contigs);
f-2 the dynamic-k passes on the (k-1)-mer rows of a random genome's k-mers, both strands (rfx_dyn_run: random reflection,
four FirstFour passes, Iteration passes).  Prints one JSON object: times and the bytes each stage's records hold."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


LUT = np.frombuffer(b"ACGT", np.uint8)


def dedup_block(rfx, n_pairs, seed=5, lo=600, hi=3000):
    """N random contigs + their reverse complements, shuffled -> rfx_dedup_contigs.  Algorithmic bytes: every base read once and
    every surviving base written once (one byte per base in HBM)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi, n_pairs)
    off = np.zeros(n_pairs + 1, np.int64)
    off[1:] = np.cumsum(lens)
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    contigs = []
    for i in range(n_pairs):
        c = codes[off[i]:off[i + 1]]
        contigs.append(LUT[c].tobytes().decode())
        contigs.append(LUT[3 - c[::-1]].tobytes().decode())
    order = rng.permutation(len(contigs))
    contigs = [contigs[i] for i in order]
    total = sum(map(len, contigs))
    rfx.dedup_contigs(contigs[:64], 500)                      # warm-up
    t0 = time.perf_counter()
    surv, text, rounds = rfx.dedup_contigs(contigs, 500)
    dt = time.perf_counter() - t0
    lib_ms = getattr(rfx, "last_call_ms", None)
    out_b = sum(map(len, surv))
    return {"what": "contig RC de-duplication (P/ReflexivDSDynamicKmerDedup.java) of random contigs + their reverse complements, shuffled; "
                    "host strings in, host strings out", "contigs_in": len(contigs), "bases_in": total,
            "contigs_after_each_round": rounds, "contigs_out": len(surv), "bases_out": out_b, "wall_ms": dt * 1e3,
            "inside_the_c_abi_ms": lib_ms,      # the rest of wall_ms is this harness turning Python strings into arrays and back
            "algorithmic_bytes": total + out_b, "achieved_GBps": (total + out_b) / dt / 1e9, "hbm_frac": (total + out_b) / dt / 1e9 / 8000.0}


def dyn_block(rfx, genome_len, k=31, seed=7, P=8, iterations=(5, 14)):
    from reflexiv_amd.api import DynRecords
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, genome_len).astype(np.uint8)
    # every k-mer of both strands once (a repeat-free random genome at this k): key = its first k - 1 bases, ext = the last
    n1 = genome_len - k + 1
    idx = np.arange(n1)[:, None] + np.arange(k)[None, :]
    fw = g[idx]                                               # [n1, k]
    rv = (3 - fw[:, ::-1])
    km = np.concatenate([fw, rv])
    n = len(km)
    o = rng.permutation(n)
    km = km[o]
    key = np.ascontiguousarray(km[:, :k - 1]).reshape(-1)
    ext = np.ascontiguousarray(km[:, k - 1])
    r = DynRecords(key, np.arange(n + 1, dtype=np.int64) * (k - 1), ext, np.arange(n + 1, dtype=np.int64), np.ones(n, np.int32),
                   np.full(n, -1, np.int32), np.full(n, -1, np.int32))
    small = DynRecords(key[:(k - 1) * 1000].copy(), r.key_off[:1001].copy(), ext[:1000].copy(), r.ext_off[:1001].copy(), r.marker[:1000].copy(),
                       r.left[:1000].copy(), r.right[:1000].copy())
    rfx.dyn_run(small, P, True, 4, iterations[0], iterations[1])   # warm-up
    t0 = time.perf_counter()
    out, trace = rfx.dyn_run(r, P, True, 4, iterations[0], iterations[1])
    dt = time.perf_counter() - t0
    lib_ms = getattr(rfx, "last_call_ms", None)
    lens = np.sort((np.diff(out.key_off) + np.diff(out.ext_off)))[::-1]
    # algorithmic bytes: a pass reads and writes every row once -- one byte per base + 12 bytes of marker / left / right + two
    # 8-byte offsets; the bases of the set are conserved (2 n (k - 1) + n), the rows shrink as the trace says
    rows_seen = n + sum(trace[:-1])
    bases = n * k
    algo = 2 * (len(trace) * bases + 28 * rows_seen)
    return {"what": "dynamic-k passes (P/ReflexivDSDynamicKmerFirstFour.java / ...Iteration.java) on the (k-1)-mer rows of a random genome's "
                    "k-mers, both strands: random reflection, four FirstFour passes, Iteration passes; host rows in, host rows out",
            "rows_in": n, "k": k, "P": P, "passes": len(trace),
            "rows_after_each_pass": trace[:6] + (["..."] if len(trace) > 6 else []) + trace[-2:],
            "rows_out": out.n, "longest": [int(x) for x in lens[:3]], "wall_ms": dt * 1e3, "inside_the_c_abi_ms": lib_ms, "rows_per_s": rows_seen / dt,
            "algorithmic_bytes": algo, "achieved_GBps": algo / dt / 1e9, "hbm_frac": algo / dt / 1e9 / 8000.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=50_000)
    ap.add_argument("--genome", type=int, default=5_000_000)
    a = ap.parse_args()
    import reflexiv_amd
    rfx = reflexiv_amd.Reflexiv(0)
    out = {}
    if a.pairs:
        out["dedup"] = dedup_block(rfx, a.pairs)
        print(json.dumps(out["dedup"]), flush=True)
    if a.genome:
        out["dyn"] = dyn_block(rfx, a.genome)
        print(json.dumps(out["dyn"]), flush=True)
    rfx.close()


if __name__ == "__main__":
    main()
