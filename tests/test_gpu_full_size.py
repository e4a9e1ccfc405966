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

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIN = os.path.join(GOLD, "c2_full.json")
# config-5-SHAPED pin: shallow coverage (48x of a 40 Mbp genome, 1.9 Gbp), the error cutoff on
# (-cover 3), distinct / instances = 0.17 -- the regime where the leaves' tables fill, overflow and split; made by
# `make_c2_full.py --reads 12666668 --genome 40000000 --cover 3 --ks 31 --out tests/golden/c5_shape.json`
PIN_C5 = os.path.join(GOLD, "c5_shape.json")


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


def make_reads(rfx, pin_path):
    import torch
    w = json.load(open(pin_path))["workload"]
    n_reads, L, G = w["reads"], w["read_len"], w["genome"]
    wpr = (L + 31) // 32
    dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda")
    dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rfx.synth_genome_dev(w["seed"], G, dg.data_ptr())
    rfx.synth_reads_dev(w["seed"], dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr(), w["err_per_2_32"])
    rfx.sync()
    return w, dw, wpr


@pytest.fixture(scope="module")
def reads_dev(rfx):
    return make_reads(rfx, PIN)


@pytest.mark.skipif(not os.path.exists(PIN_C5), reason="tests/golden/c5_shape.json not generated")
def test_config5_shaped_workload_matches_the_oracle_pin(rfx):
    check_against_pin(rfx, make_reads(rfx, PIN_C5), 31, PIN_C5)
    stats = rfx.count_timing()
    # (the assembly ran after the count: the statistics are the count's only if the driver did not clear them)


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/c2_full.json not generated")
@pytest.mark.parametrize("k", [31, 63])
def test_full_size_count_and_contigs_match_the_oracle_pin(rfx, reads_dev, k):
    check_against_pin(rfx, reads_dev, k, PIN)


def check_against_pin(rfx, reads_dev, k, pin_path):
    import torch
    import reflexiv_amd
    pin = json.load(open(pin_path))
    if f"k{k}" not in pin:
        pytest.skip(f"no k={k} record in the pin")
    rec = pin[f"k{k}"]
    w, dw, wpr = reads_dev
    n_reads, L, cover, P = w["reads"], w["read_len"], w["cover"], w["partitions"]
    cap = max(1 << 24, int(rec["n_kept"] * 1.05) + 4096)
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
    if k <= 31 and rec["n_contigs"] <= 64:
        # f-4 at full size: the de-duplicated contigs (P/ReflexivDSDynamicKmerDedup.java) against the oracle on the same text
        from oracle import oracle as O
        contigs = ["".join(part.split("\n")[1:]) for part in text.split(">")[1:]]
        want = O.dedup_contigs(contigs)
        dtext, dnc, drn = rfx.dedup_contig_text(text)
        assert drn == [len(r) for r in want["rounds"]] and dtext == want["text"]
        assert dnc < rec["n_contigs"] and sum(len(c) for c in want["rounds"][2]) < 0.6 * sum(len(c) for c in contigs)


def test_full_size_pins_hold_with_every_allocation_poisoned():
    """RFX_POISON=7 (rfx_internal.h): every workspace slot, scratch allocation and reused record slot is filled with 0xA5
    before the library uses it -- the config-2 pins (k = 31 and 63, count and contigs) and the config-5-shaped one must not
    notice.  A child process: the mask is read once per process."""
    import subprocess
    import sys
    env = dict(os.environ, RFX_POISON="7")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", here, "-k", "oracle_pin"],
                       env=env, cwd=os.path.dirname(os.path.dirname(here)), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_extend_stage_without_the_mailbox():
    """RFX_MAILBOX=0: the extend stage's few-byte readbacks as queued copies + stream waits again (what rounds 1-3 had, and the
    fallback of mailbox_wait) -- the reference-made vectors and the k > 31 driver tests must hold on that path too.  A child
    process: the switch is read once per process."""
    import subprocess
    import sys
    env = dict(os.environ, RFX_MAILBOX="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_reference_vectors.py"), os.path.join(root, "tests", "test_gpu_asm_w.py")],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
