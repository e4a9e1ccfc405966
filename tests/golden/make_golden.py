"""Regenerates tests/golden/*.npz.  Run in the build container (needs /root/reference
for the example reads; everything else comes from the oracle):

    python tests/golden/make_golden.py

Fixtures are data only: the example reads (ASCII bases + offsets), the documented
known answer (docs/example.html:331-343 -- header and first 1200 bases), and
oracle outputs per operator.
"""
import hashlib
import html
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"


def rec_dict(prefix, r):
    return {f"{prefix}_key": r.key, f"{prefix}_marker": r.marker, f"{prefix}_ext_off": r.ext_off,
            f"{prefix}_ext": r.ext, f"{prefix}_left": r.left, f"{prefix}_right": r.right}


def stages(keys, counts, k, P, min_err, twin):
    """Every operator's output under the order contract, as a dict of arrays."""
    d = {}
    r = O.rc_expand_subkmer(keys, counts, k)
    d.update(rec_dict("k5", r))
    r = O.sort_records(r)
    r, ps = O.fork_filter_forward(r, O.partition_starts(r.key, P), k, min_err, twin)
    d.update(rec_dict("k6", r))
    r = O.reflect_from_forward(r, k)
    d.update(rec_dict("k7", r))
    r = O.sort_records(r)
    r, ps = O.fork_filter_reflected(r, O.partition_starts(r.key, P), k, min_err, twin)
    d.update(rec_dict("k8", r)); d["k8_part_start"] = ps
    r = O.random_reflection(r, ps, k)
    d.update(rec_dict("k9", r))
    for i in range(6):                       # first six extend passes, record for record
        r = O.sort_records(r)
        r, ps = O.extend_pass(r, O.partition_starts(r.key, P), k, twin)
        d.update(rec_dict(f"pass{i}", r))
    return d


def example():
    bases, off = O.load_fastq([f"{REF}/example/paired_dat1.fq.gz", f"{REF}/example/paired_dat2.fq.gz"])
    doc = open(f"{REF}/docs/example.html").read()
    m = re.search(r"<pre>\s*((?:&gt;|>)Contig-4558-0)\n((?:[ACGT]{100}\n?)+)</pre>", doc)
    header = html.unescape(m.group(1))
    prefix = m.group(2).replace("\n", "")
    assert len(prefix) == 1200
    km = O.extract_canon(bases, off, 31)
    keys, counts, nd = O.count_filter(km, 3, 10_000_000)
    out = {"bases": bases, "read_off": off, "doc_header": np.array(header),
           "doc_prefix1200": np.array(prefix), "doc_part_bytes": np.array(4619),
           "k1_first4": O.extract_canon(bases[:off[4]], off[:5], 31),
           "n_instances": np.array(len(km)), "n_distinct": np.array(nd),
           "keys_cov3": keys, "counts_cov3": counts}
    out.update(stages(keys, counts, 31, 4, 8, O.TWIN_DS))
    for P in (1, 2, 4, 8):
        for twin, tn in ((O.TWIN_DS, "ds"), (O.TWIN_RDD, "rdd")):
            prm = O.default_params(min_cov=3, partitions=P, twin=twin)
            text, nc, trace, rec = O.assemble_from_counts(keys, counts, prm)
            out[f"contigs_{tn}_P{P}"] = np.array(text)
            out[f"trace_{tn}_P{P}"] = np.array(trace, np.int64)
    np.savez_compressed(os.path.join(HERE, "example.npz"), **out)
    print("example.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:12]})


def planted():
    """10 kbp genome, two haplotypes (one SNP) + one 300-bp repeat, 30x PE100 reads, 0.5 % errors."""
    seed, G, L = 20251003, 10_016, 100
    g = O.synth_genome(seed, G)
    # unpack, plant a repeat and a SNP, repack
    b = np.zeros(G, np.uint8)
    for i in range(G):
        b[i] = (int(g[i >> 5]) >> (62 - 2 * (i & 31))) & 3
    b[7000:7300] = b[2000:2300]                      # repeat
    hap_b = b.copy(); hap_b[5000] = (hap_b[5000] + 2) & 3   # SNP bubble

    def pack(x):
        w = np.zeros((G + 31) // 32, np.uint64)
        for i in range(G):
            w[i >> 5] |= np.uint64(int(x[i]) << (62 - 2 * (i & 31)))
        return w
    ga, gb = pack(b), pack(hap_b)
    n_reads = 2 * (30 * G // (2 * L))
    ra, off = O.synth_reads(seed, ga, G, 0, n_reads, L)
    rb, _ = O.synth_reads(seed, gb, G, 0, n_reads, L)
    ra = ra.reshape(n_reads, L).copy(); rb = rb.reshape(n_reads, L)
    odd_pairs = ((np.arange(n_reads) >> 1) & 1) == 1
    ra[odd_pairs] = rb[odd_pairs]
    bases = ra.reshape(-1)
    out = {"bases": bases, "read_off": off}
    for k in (31, 21):
        km = O.extract_canon(bases, off, k)
        keys, counts, nd = O.count_filter(km, 2, 10_000_000)
        out[f"k{k}_keys"] = keys; out[f"k{k}_counts"] = counts
        for twin, tn in ((O.TWIN_DS, "ds"), (O.TWIN_RDD, "rdd")):
            st = stages(keys, counts, k, 4, 8, twin)
            nb = int(((st["k8_left"] >= 0) | (st["k8_right"] >= 0)).sum())
            for name in ("k6", "k8", "k9", "pass0", "pass3", "pass5"):
                for f in ("key", "marker", "ext_off", "ext", "left", "right"):
                    out[f"k{k}_{tn}_{name}_{f}"] = st[f"{name}_{f}"]
            prm = O.default_params(k=k, min_cov=2, partitions=4, twin=twin, min_contig=100)
            text, nc, trace, rec = O.assemble_from_counts(keys, counts, prm)
            out[f"k{k}_{tn}_contigs"] = np.array(text)
            out[f"k{k}_{tn}_trace"] = np.array(trace, np.int64)
            print(f"planted k={k} {tn}: kmers {len(keys)} fork-marked {nb} contigs {nc} passes {len(trace)}")
    np.savez_compressed(os.path.join(HERE, "planted.npz"), **out)


def wide():
    """k > 31 (the counter's multi-word k-mers) on the example reads."""
    ex = np.load(os.path.join(HERE, "example.npz"))
    bases, off = ex["bases"], ex["read_off"]
    out = {}
    for k in (63, 47):
        km = O.extract_canon_w(bases, off, k)
        keys, counts, nd = O.count_filter_w(km, k, 3, 10_000_000)
        out[f"k{k}_first4"] = O.extract_canon_w(bases[:off[4]], off[:5], k)
        out[f"k{k}_n_instances"] = np.array(len(km)); out[f"k{k}_n_distinct"] = np.array(nd)
        out[f"k{k}_keys_cov3"] = keys; out[f"k{k}_counts_cov3"] = counts
        out[f"k{k}_text_first"] = np.array(O.kmer_text_w(keys[0], k))
        print(f"wide k={k}: instances {len(km)} distinct {nd} kept {len(keys)}")
    np.savez_compressed(os.path.join(HERE, "wide.npz"), **out)


if __name__ == "__main__":
    if "--wide-only" in sys.argv:
        wide()
        sys.exit(0)
    example()
    planted()
    wide()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            p = os.path.join(HERE, f)
            print(f, os.path.getsize(p), hashlib.sha256(open(p, "rb").read()).hexdigest()[:16])
