cd $GRAFT_REPO_ROOT
for b in "" "9,9" "8,10"; do
RFX_LEVEL_BITS=$b python bench.py --k 63 --steps 4 --warmup 2 --no-cpu-baseline --no-contigs --no-strong --no-next-rows --no-ingest > gpurun_out/k63_bits.json 2> gpurun_out/k63_bits.err
python - <<PY
import json
d = json.loads(open("gpurun_out/k63_bits.json").read().strip().splitlines()[-1])
print("k63 bits '$b'", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["per_kernel_ms_per_step"].items()}, d["config"].get("distinct_kmers"), d["config"].get("kmers_kept"))
PY
done
