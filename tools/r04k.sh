cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_asm_w.py tests/test_gpu_reference_vectors.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r4k_tests.log 2>&1; tail -1 gpurun_out/r4k_tests.log
for m in 0 1; do
RFX_MAILBOX=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-strong > gpurun_out/r4k.json 2> gpurun_out/r4k.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r4k.json").read().strip().splitlines()[-1])
c = d["contigs"]; k = d["k63"]
print("RFX_MAILBOX=$m", round(d["ms_per_step"], 2), "contigs from counts", round(c["wall_ms_from_counts"], 2), "reads->contigs", round(c["wall_ms_reads_to_contigs"], 2), c["sha256_text"][:12], "dedup", round(c["dedup"]["wall_ms"], 2), "| k63", round(k["ms_per_step"], 2), {kk: (round(v, 2) if isinstance(v, float) else v) for kk, v in k.get("contigs", {}).items() if "wall" in kk})
PY
done
