cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_r04
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r4f.json 2> gpurun_out/r4f.err
python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd31.json 2> gpurun_out/prof_r04/fd31.err
python bench.py --force-dist --gbp 6.25 --k 63 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd63.json 2> gpurun_out/prof_r04/fd63.err
python - <<PY
import json
for f in ("gpurun_out/r4f.json", "gpurun_out/prof_r04/fd31.json", "gpurun_out/prof_r04/fd63.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 2), d["roofline"].get("per_kernel_ms_per_step"), d.get("k63", {}).get("ms_per_step"), d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
    s = d.get("strong_50gbp")
    if s: print("  strong", s["ms_per_step"], s["per_kernel_ms_last_step"], s["distinct_kmers"], s["kmers_kept"])
PY
