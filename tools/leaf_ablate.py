"""Leaf-kernel ablations in ONE process (the RFX_* knobs are read per call): k = 31 and k = 63 on the config-2 reads.
RFX_LEAF_DBG bits: 1 stream (+ record table) only, 2 expand but no k-mer table, 64 no record table, 128 record-table statistics."""
import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
n_reads, L, G = 33333334, 150, 4640000
wpr = 5
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1, G, dg.data_ptr()); rfx.synth_reads_dev(1, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
cap = 1 << 24
dk = torch.empty(cap * 2, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()

def run(k, env, reps=3):
    old = {a: os.environ.get(a) for a in env}
    os.environ.update(env)
    try:
        best = None
        for _ in range(reps):
            try:
                if k > 32:
                    m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 30)
                else:
                    m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, 30)
            except Exception as e:
                m, nd = -1, str(e)[:60]
            st = rfx.count_timing()
            leaf = st.get("leaf", (0, 0))[0]
            best = leaf if best is None else min(best, leaf)
        print(f"k={k} {env}: leaf {best:.2f} ms, kept {m}, distinct {nd}, passes {st.get('stat_passes', (0, 0))[1]}, "
              f"overflows {st.get('stat_overflows', (0, 0))[1]}, all {({a: round(b[0], 2) for a, b in st.items() if not a.startswith('stat_')})}", flush=True)
    finally:
        for a, v in old.items():
            if v is None: os.environ.pop(a, None)
            else: os.environ[a] = v

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "31"):
    run(31, {})
    run(31, {"RFX_LEAF_DBG": "128"}, 1)
    run(31, {"RFX_LEAF_DBG": "1"})
    run(31, {"RFX_LEAF_DBG": "2"})
    run(31, {"RFX_LEAF_DBG": "64"})
    run(31, {"RFX_LEAF_DBG": "66"})
    run(31, {"RFX_LEAF_DBG": "65"})
if which in ("all", "63"):
    run(63, {})
    run(63, {"RFX_WIDE_STATS": "1", "RFX_TRACE": "1"}, 1)
    run(63, {"RFX_WIDE_DBG": "1"})
    run(63, {"RFX_WIDE_NOAGG": "1"})
    run(63, {"RFX_WIDE_NOAGG": "1", "RFX_WIDE_DBG": "1"})
    for ps in (1600, 2000, 3200, 4000):
        run(63, {"RFX_WIDE_PRESPLIT": str(ps)})
if which == "seg":
    for _ in range(2):
        run(31, {"RFX_TRACE": "1"}, 2)
        run(31, {"RFX_SK_SEG": "16", "RFX_TRACE": "1"}, 2)
    run(25, {})
    run(25, {"RFX_SK_SEG": "16"})
    run(21, {})
    run(21, {"RFX_SK_SEG": "16"})
if which == "63one":
    run(63, {}, 1)
if which == "bits":
    for env in ({}, {"RFX_LEVEL_BITS": "8,10"}, {"RFX_LEVEL_BITS": "9,10"}, {"RFX_LEVEL_BITS": "8,9"}, {"RFX_LEAF_TARGET": "24576"}, {"RFX_PRESPLIT": "12000"}, {}):
        run(31, env, 2)
