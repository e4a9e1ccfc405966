cd $GRAFT_REPO_ROOT
for b in "8,9" "8,8" "7,9" "9,9"; do
RFX_LEVEL_BITS=$b python bench.py --force-dist --gbp 6.25 --steps 4 --warmup 2 --no-cpu-baseline --no-contigs > gpurun_out/fd_bits.json 2> gpurun_out/fd_bits.err
python - <<PY
import json
d = json.loads(open("gpurun_out/fd_bits.json").read().strip().splitlines()[-1])
print("$b", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["per_kernel_ms_per_step"].items()}, d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
done
