# k = 31 fused count: level-bit plans (one box, one run: only these numbers compare)
for v in "RFX_X=1" "RFX_TPB=8" "RFX_TPB=16" "RFX_TPB=64" "RFX_X=1"; do
  echo "== $v"
  env $v RFX_TRACE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-contigs --no-cpu-baseline > gpurun_out/bs.log 2>&1
  grep "^leaves" gpurun_out/bs.log | tail -1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bs.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print(round(j["ms_per_step"],1), j["config"]["distinct_kmers"], {k:round(v,1) for k,v in j["roofline"]["per_kernel_ms_per_step"].items()})
else: print(open("gpurun_out/bs.log").read()[-800:])
PY
done
