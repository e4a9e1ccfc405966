# parity of the count stage + the default line (GPU box) after a leaf change
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_asm_w.py -x -q -m gpu -k "wide or _w" > gpurun_out/r4c_tests.log 2>&1; tail -2 gpurun_out/r4c_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-strong > gpurun_out/r4c.json 2> gpurun_out/r4c.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r4c.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), d["roofline"].get("per_kernel_ms_per_step"))
k = d.get("k63", {})
print("k63", k.get("ms_per_step"), k.get("per_kernel_ms_per_step") or k.get("roofline", {}).get("per_kernel_ms_per_step"))
PY
