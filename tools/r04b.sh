# the multi-GPU rehearsal and the default line after a change of the count stage (GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r04
RFX_TRACE=1 python bench.py --force-dist --gbp 6.25 --steps 1 --warmup 0 --no-cpu-baseline --no-contigs > gpurun_out/fd31_trace.json 2> gpurun_out/fd31_trace.err
grep -v "^\[W\|amdgpu" gpurun_out/fd31_trace.err | head -8
python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd31.json 2> gpurun_out/prof_r04/fd31.err &&
python bench.py --force-dist --gbp 6.25 --k 63 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd63.json 2> gpurun_out/prof_r04/fd63.err &&
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r4_l2s2.json 2> gpurun_out/r4_l2s2.err
python - <<PY
import json
for f in ("gpurun_out/prof_r04/fd31.json", "gpurun_out/prof_r04/fd63.json", "gpurun_out/r4_l2s2.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 2), d["roofline"].get("per_kernel_ms_per_step"), d.get("k63", {}).get("ms_per_step"), d.get("strong_50gbp", {}).get("ms_per_step"))
PY
