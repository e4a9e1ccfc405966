cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_r04
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_level_of_received or owner_buckets or sharded" > gpurun_out/r4d_tests.log 2>&1; tail -4 gpurun_out/r4d_tests.log
RFX_TRACE=1 python bench.py --force-dist --gbp 6.25 --steps 1 --warmup 0 --no-cpu-baseline --no-contigs > gpurun_out/fd31_trace.json 2> gpurun_out/fd31_trace.err
grep -v "^\[W\|amdgpu" gpurun_out/fd31_trace.err | head -6
python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd31.json 2> gpurun_out/prof_r04/fd31.err &&
python bench.py --force-dist --gbp 6.25 --k 63 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd63.json 2> gpurun_out/prof_r04/fd63.err
python - <<PY
import json
for f in ("gpurun_out/prof_r04/fd31.json", "gpurun_out/prof_r04/fd63.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 2), d["roofline"].get("per_kernel_ms_per_step"), d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
