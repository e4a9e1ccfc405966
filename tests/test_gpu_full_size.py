"""BASELINE config 2 at FULL size (33,333,334 PE150 reads, 4.0e9 k-mer instances; and the same reads at k = 63)
on the GPU against the pin the CPU oracle produced on the same workload (tests/golden/c2_full.json, made by
tests/golden/make_c2_full.py): instance / distinct / kept counts, sha256 of the survivor list, the extend trace and
sha256 of the contig text -- the exact workload bench.py times, checked against something other than itself."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_full.json")


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


@pytest.fixture(scope="module")
def reads_dev(rfx):
    import torch
    w = json.load(open(PIN))["workload"]
    n_reads, L, G = w["reads"], w["read_len"], w["genome"]
    wpr = (L + 31) // 32
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(w["seed"], G, dg.data_ptr())
    rfx.synth_reads_dev(w["seed"], dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr(), w["err_per_2_32"])
    rfx.sync()
    return w, dw, wpr


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/c2_full.json not generated")
@pytest.mark.parametrize("k", [31, 63])
def test_full_size_count_and_contigs_match_the_oracle_pin(rfx, reads_dev, k):
    import torch
    import reflexiv_amd
    pin = json.load(open(PIN))
    if f"k{k}" not in pin:
        pytest.skip(f"no k={k} record in the pin")
    rec = pin[f"k{k}"]
    w, dw, wpr = reads_dev
    n_reads, L, cover, P = w["reads"], w["read_len"], w["cover"], w["partitions"]
    cap = 1 << 24
    prm = reflexiv_amd.default_params(k=k, min_cov=cover, partitions=P)
    if k <= 31:
        dk = torch.empty(cap, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, cover)
    else:
        W = k // 32 + 1
        dk = torch.empty(cap * W, dtype=torch.int64, device="cuda"); dc = torch.empty(cap, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m, nd, inst = rfx.count_reads_w_dev(dw.data_ptr(), n_reads, wpr, L, k, dk.data_ptr(), dc.data_ptr(), cap, cover)
        dk = dk[:m * W]
    assert (inst, nd, m) == (rec["n_instances"], rec["n_distinct"], rec["n_kept"])
    keys = dk[:m * (1 if k <= 31 else k // 32 + 1)].cpu().numpy().view(np.uint64)
    counts = dc[:m].cpu().numpy()
    assert hashlib.sha256(keys.tobytes()).hexdigest() == rec["sha256_keys"]
    assert hashlib.sha256(counts.tobytes()).hexdigest() == rec["sha256_counts"]
    if k <= 31:
        text, nc, trace = rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), m, prm)
    else:
        aw = (k - 1) // 31 + 1
        ak = torch.empty(m * aw, dtype=torch.int64, device="cuda"); ac = torch.empty(m, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m2 = rfx.counter_to_asm_dev(dk.data_ptr(), dc.data_ptr(), m, k, ak.data_ptr(), ac.data_ptr(), cover)
        assert m2 == m
        text, nc, trace = rfx.assemble_w_dev(ak.data_ptr(), ac.data_ptr(), m2, prm)
    assert trace == rec["trace"]
    assert nc == rec["n_contigs"] and len(text) == rec["contig_text_bytes"]
    assert hashlib.sha256(text.encode()).hexdigest() == rec["sha256_contig_text"]
