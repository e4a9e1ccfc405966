cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_r04
RFX_TRACE=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "outgrow" -s > gpurun_out/r4e_tests.log 2>&1; grep "spilled\|sweep\|passed\|failed\|Error\|assert" gpurun_out/r4e_tests.log | sort | uniq -c | sort -rn | head -12
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_asm_w.py -x -q -m gpu > gpurun_out/r4e_tests2.log 2>&1; tail -2 gpurun_out/r4e_tests2.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-strong > gpurun_out/r4e.json 2> gpurun_out/r4e.err
python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd31.json 2> gpurun_out/prof_r04/fd31.err
RFX_REC_ONESWEEP=1 python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/fd31_claim.json 2> gpurun_out/fd31_claim.err
python - <<PY
import json
for f in ("gpurun_out/r4e.json", "gpurun_out/prof_r04/fd31.json", "gpurun_out/fd31_claim.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 2), d["roofline"].get("per_kernel_ms_per_step"), d.get("k63", {}).get("ms_per_step"), d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
