cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err &&
python bench.py --force-dist --gbp 6.25 --cover 38 --steps 5 --warmup 2 > gpurun_out/r03b_fd31.json 2> gpurun_out/r03b_fd31.err &&
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03b_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["per_kernel_ms_per_step"], d["roofline"]["frac"], d["contigs"], d.get("k63",{}).get("ms_per_step"))
d=json.loads(open("gpurun_out/r03b_fd31.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["per_kernel_ms_per_step"])
PY
