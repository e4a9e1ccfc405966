"""The HIP operators (through the C ABI) against vectors made by the REFERENCE'S OWN operator classes
(tests/golden/reference_vectors.npz; see tests/golden/make_reference_vectors.py and tests/test_reference_vectors.py,
whose helpers and documented deviations are shared here).  Every operator of every chain, every fuzz set, and the
device-resident drivers end to end (reads -> contig text) for k = 25, 31 (both twins) and k = 47, 63, 95."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import test_reference_vectors as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rfx():
    import reflexiv_amd
    r = reflexiv_amd.Reflexiv()
    yield r
    r.close()


@pytest.fixture(scope="module")
def vec():
    z = np.load(T.VEC)
    return {k: z[k] for k in z.files}


X_OPS = {"x_double": 0, "x_extendable_pairs": 1, "x_unextendable": 2, "x_first_of_key": 3, "x_longer_of_key": 4,
         "x_left_ends": 5, "x_right_ends": 6}


def run_stage_gpu(rfx, label, r, starts, k, mec, twin):
    if label == "fork_forward":
        return rfx.FilterForkSubKmer(r, starts, k, mec, twin)
    if label == "fork_reflected":
        return rfx.FilterForkReflectedSubKmer(r, starts, k, mec, twin)
    if label == "reflect":
        return rfx.ReflectedSubKmerExtractionFromForward(r, k), starts
    if label == "random_reflection":
        return rfx.kmerRandomReflection(r, starts, k), starts
    if label.startswith("extend_"):
        stage = 0 if "single" in label else 1 if "first" in label else 2
        return rfx.ExtendReflexivKmer(r, starts, k, twin, stage, 3 if label.endswith("scr3") else 2)
    if label in X_OPS:
        g, gst = rfx.extras_operator(X_OPS[label], r, starts, k)
        return g, (None if label in ("x_double", "x_left_ends", "x_right_ends") else gst)
    raise KeyError(label)


@pytest.mark.parametrize("name", T.names("chain_"))
def test_gpu_chain_every_operator_equals_the_reference_classes(rfx, vec, name):
    fam = T.fam_of(name)
    meta = [int(x) for x in vec[name + "/meta"]]
    k, P, min_cov, mec = meta[:4]
    twin = T.twin_of(fam)
    stages = sorted({key.split("/")[1] for key in vec if key.startswith(name + "/s") and key.split("/")[1][1:3].isdigit()})
    stats = {}
    for s in stages:
        label = s[4:]
        r, st = T.load_records(vec, f"{name}/{s}/in", k)
        want, wst = T.load_records(vec, f"{name}/{s}/out", k)
        got, gst = run_stage_gpu(rfx, label, r, st, k, mec, twin)
        T.compare(T.rec_tuples(got), T.rec_tuples(want), fam, label, stats)
        if gst is not None and wst is not None:
            assert np.array_equal(np.asarray(gst, np.int64), wst), (s, gst, wst)
    if fam == "ds64":
        fwd = rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(vec[name + "/asm_keys"], vec[name + "/asm_counts"], k)
    else:
        fwd = rfx.KmerReverseComplement_and_ForwardSubKmerExtraction(vec[name + "/kept_keys"], vec[name + "/kept_counts"], k)
    want, _ = T.load_records(vec, name + "/forward", k)
    T.compare(T.rec_tuples(fwd), T.rec_tuples(want), fam, "forward", stats)


@pytest.mark.parametrize("name", T.names("chain_") + T.names("example_"))
def test_gpu_end_to_end_equals_the_reference(rfx, vec, name):
    """reads -> extraction -> count / filter -> the device-resident driver -> contig text, against the text and the
    per-pass record counts the reference's classes produced"""
    import torch
    import reflexiv_amd
    fam = T.fam_of(name)
    meta = [int(x) for x in vec[name + "/meta"]]
    k, P, min_cov, mec, max_iter, min_iter, min_contig = meta
    if fam == "ds64":
        reads = bytes(vec[name + "/reads"]).decode().split("\n")[:-1]
        bases = np.frombuffer("".join(reads).encode(), np.uint8)
        off = np.zeros(len(reads) + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in reads])
        inst = rfx.ReverseComplementKmerBinaryExtractionFromDataset64(bases, off, k)
        assert np.array_equal(np.asarray(inst).reshape(-1), vec[name + "/instances"].reshape(-1))
        keys, counts, _ = rfx.groupBy_count_filter_w(inst, k, min_cov)
        W = k // 32 + 1
        aw = (k - 1) // 31 + 1
        m = len(counts)
        dk = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64).reshape(-1)).cuda()
        dc = torch.from_numpy(np.asarray(counts, np.int64)).cuda()
        a_k = torch.empty(max(1, m) * aw, dtype=torch.int64, device="cuda")
        a_c = torch.empty(max(1, m), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m2 = rfx.counter_to_asm_dev(dk.data_ptr(), dc.data_ptr(), m, k, a_k.data_ptr(), a_c.data_ptr(), min_cov)
        rfx.sync()
        assert np.array_equal(a_k[:m2 * aw].cpu().numpy().view(np.uint64).reshape(m2, aw), vec[name + "/asm_keys"].reshape(m2, aw))
        assert np.array_equal(a_c[:m2].cpu().numpy(), vec[name + "/asm_counts"])
        prm = reflexiv_amd.default_params(k=k, min_cov=min_cov, min_error_cov=mec, partitions=P, max_iter=max_iter,
                                          min_iter=min_iter, min_contig=min_contig)
        text, nc, trace = rfx.assemble_w_dev(a_k.data_ptr(), a_c.data_ptr(), m2, prm)
        assert W >= 2
    else:
        twin = T.twin_of(fam)
        if name.startswith("example_"):
            import os
            ex = np.load(os.path.join(os.path.dirname(T.VEC), "example.npz"))
            bases, off = ex["bases"], ex["read_off"]
        else:
            fq = bytes(vec[name + "/fastq"])
            seq_off, seq_len = O.fastq_group(fq)                 # (host-side line grouping: the oracle's, checked on the CPU)
            buf = np.frombuffer(fq, np.uint8)
            bases = np.concatenate([buf[o:o + n] for o, n in zip(seq_off, seq_len)])
            off = np.zeros(len(seq_off) + 1, np.int64)
            off[1:] = np.cumsum(seq_len)
        inst = rfx.ReverseComplementKmerBinaryExtraction(bases, off, k)
        if not name.startswith("example_"):
            assert np.array_equal(inst, vec[name + "/instances"])
        keys, counts, _ = rfx.KmerCounting_and_CoverageFilter(inst, min_cov, 10_000_000, twin)
        assert np.array_equal(keys, vec[name + "/kept_keys"]) and np.array_equal(counts, vec[name + "/kept_counts"])
        dk = torch.from_numpy(keys.view(np.int64)).cuda()
        dc = torch.from_numpy(counts).cuda()
        torch.cuda.synchronize()
        prm = reflexiv_amd.default_params(k=k, min_cov=min_cov, min_error_cov=mec, partitions=P, max_iter=max_iter,
                                          min_iter=min_iter, min_contig=min_contig, twin=twin)
        text, nc, trace = rfx.assemble_dev(dk.data_ptr(), dc.data_ptr(), len(keys), prm)
    assert list(trace) == [int(x) for x in vec[name + "/trace"]]
    assert text == bytes(vec[name + "/contigs"]).decode()


@pytest.mark.parametrize("name", T.names("fuzz_"))
def test_gpu_fuzz_operator_equals_the_reference_class(rfx, vec, name):
    fam = T.fam_of(name)
    cls = name.split("_", 3)[3]
    k = int(vec[name + "/meta"][0])
    twin = T.twin_of(fam)
    r, st = T.load_records(vec, name + "/in", k)
    want, wst = T.load_records(vec, name + "/out", k)
    base = cls.replace("_scr3", "")
    base = base[2:] if base.startswith("DS") else base
    mec = 0
    if "FilterFork" in base:
        mec = 8 if "ErrorCorrection" in base else 0
        label = "fork_reflected" if "Reflected" in base else "fork_forward"
    elif base.startswith("ExtendReflexivKmer"):
        label = {"ExtendReflexivKmer": "extend_single", "ExtendReflexivKmerToArrayFirstTime": "extend_first_array",
                 "ExtendReflexivKmerToArrayLoop": "extend_array"}[base] + ("_scr3" if cls.endswith("_scr3") else "")
    else:
        label = {"ReflexivAndForwardKmer": "x_double", "FilterExtendableKmerPairs": "x_extendable_pairs",
                 "FilterUnExtendableKmer": "x_unextendable", "FilterStillExtendableKmerFromPairs": "x_first_of_key",
                 "FilterStillExtendableKmerEnds": "x_longer_of_key", "FilterUnExtendableKmerLeftEnds": "x_left_ends",
                 "FilterUnExtendableKmerRightEnds": "x_right_ends"}[base]
    got, gst = run_stage_gpu(rfx, label, r, st, k, mec, twin)
    stats = {}
    T.compare(T.rec_tuples(got), T.rec_tuples(want), fam, label, stats)
    if gst is not None:
        assert np.array_equal(np.asarray(gst, np.int64), wst)
