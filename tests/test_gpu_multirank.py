"""The multi-GPU branch of the C ABI with MORE THAN ONE RANK, on the one GPU of the test box: every rank is a process of its
own (context + communicator on cuda:0), and the ten RCCL entry points libreflexiv_hip.so binds are served by a stand-in
(tests/fake_rccl: messages through /dev/shm) because RCCL refuses two ranks on one device.  What this covers that the
one-rank tests cannot: the count matrix every rank derives its send AND receive layout from, receive offsets by source,
several peers per group, the rounds under a small per-peer cap, gather to root from real peers, rfx_sharded_assemble_reads
with the reads dealt round the ranks -- against the fused one-GPU count of all the reads and the oracle
(tests/multirank_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = os.path.join(HERE, "fake_rccl")
SHIM = os.path.join(SHIM_DIR, "libfake_rccl.so")


def build_shim():
    subprocess.run(["make", "-s", "-C", SHIM_DIR], check=True, capture_output=True)
    return SHIM


@pytest.mark.parametrize("world,limit,sweep", [(2, None, False), (3, 65536, False), (4, None, False), (2, None, True), (3, 32768, True)])
def test_sharded_count_and_assemble_on_several_ranks(tmp_path, world, limit, sweep):
    env = dict(os.environ, RFX_RCCL_LIB=build_shim(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if limit:
        env["RFX_COMM_LIMIT_BYTES"] = str(limit)                # many rounds per exchange
    if sweep:
        env["RFX_SK_ONESWEEP"] = "2"                            # the sender's bucketing by level 1's one sweep: owner buckets in pieces
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:                                         # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and os.path.exists(tmp_path / f"ok{r}"), f"rank {r} of {world}:\n{o[-3000:]}"
    leftovers = [f for f in os.listdir("/dev/shm") if f.startswith("frccl-")]
    assert not leftovers, leftovers


def test_world_of_eight_as_threads_of_one_process(tmp_path):
    """The node size of BASELINE configs 3-5: 8 ranks (threads of ONE process, each with its context + communicator on
    cuda:0 -- the box's process guard allows six GPU processes).  Exercises the 8-row count matrix with real peers, 8 x 8
    receive offsets, and the G * world <= 64 cap (generations = 16 asked, 8 taken)."""
    env = dict(os.environ, RFX_RCCL_LIB=build_shim(), HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_TIMEOUT_S="120", RFX_BACKTRACE="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "multirank_worker.py"), "threads", "8", str(tmp_path)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and all(os.path.exists(tmp_path / f"ok{r}") for r in range(8)), p.stdout[-4000:]
    leftovers = [f for f in os.listdir("/dev/shm") if f.startswith("frccl-")]
    assert not leftovers, leftovers


def bench_line(extra, env, nproc):
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py")] + extra
    if nproc > 1:
        port = str(29600 + os.getpid() % 300)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(root, "bench.py")] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("k", [31, 63])
def test_bench_multi_gpu_branch_gives_the_one_gpu_answer(k):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed with both ranks
    on this box's one GPU: the same reads as a one-GPU run of twice the per-GPU size, so distinct / kept k-mers and the
    contigs must be the same.  (Timing fields of the rehearsal mean nothing and the line says so.)"""
    common = ["--genome", "200000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-k63", "--no-ingest", "--k", str(k)]
    env1 = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = bench_line(["--gpus", "1", "--gbp", "0.04", "--cover", "4"] + common, env1, 1)
    env2 = dict(env1, RFX_RCCL_LIB=build_shim(), RFX_BENCH_SHARED_GPU="1")
    two = bench_line(["--gpus", "2", "--gbp", "0.02", "--cover", "4"] + common, env2, 2)
    assert two["n_gpus"] == 2 and "rehearsal" in two and "rehearsal" not in one
    assert two["exchange"]["impl"].startswith("rfx_dev_sharded_count") and not two["exchange"]["exchange_free"]
    for f in ("distinct_kmers", "kmers_kept"):
        assert two["config"][f] == one["config"][f], f
    assert two["config"]["kmer_instances_per_gpu"] * 2 == one["config"]["kmer_instances_per_gpu"]
    for f in ("n_contigs", "longest", "total_bases", "extend_passes"):
        assert two["contigs"][f] == one["contigs"][f], f
    assert two["contigs"]["n_contigs"] > 0
