cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "outgrow or first_level_of_received" > gpurun_out/r4i_tests.log 2>&1; tail -1 gpurun_out/r4i_tests.log
for m in 0 1; do
RFX_REC_ONESWEEP=$m python bench.py --force-dist --gbp 6.25 --k 63 --steps 4 --warmup 2 --no-cpu-baseline --no-contigs > gpurun_out/fd63_m.json 2> gpurun_out/fd63_m.err
python - <<PY
import json
d = json.loads(open("gpurun_out/fd63_m.json").read().strip().splitlines()[-1])
print("k63 RFX_REC_ONESWEEP=$m", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["per_kernel_ms_per_step"].items()}, d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
done
python bench.py --force-dist --gbp 6.25 --steps 4 --warmup 2 --no-cpu-baseline --no-contigs > gpurun_out/fd31_m.json 2> gpurun_out/fd31_m.err
python - <<PY
import json
d = json.loads(open("gpurun_out/fd31_m.json").read().strip().splitlines()[-1])
print("k31", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["per_kernel_ms_per_step"].items()}, d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
