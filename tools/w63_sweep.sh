# k = 63 record path: knob sweep (one box, one run: only these numbers compare)
for v in "RFX_X=1" "RFX_WLEAF_PER_CU=4" "RFX_WLEAF_PER_CU=16" "RFX_WLEAF_PER_CU=64" "RFX_WLEAF_PER_CU=16 RFX_WIDE_RECORDS=0"; do
  echo "== $v"
  env RFX_WIDE_RECORDS=1 $v RFX_TRACE=1 timeout -k 10 90 python bench.py --k 63 --steps 2 --warmup 1 > gpurun_out/w63.log 2>&1
  grep "wide leaves" gpurun_out/w63.log | tail -1
  python - <<PY
import json
l=[x for x in open("gpurun_out/w63.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print(round(j["ms_per_step"],1), j["config"]["distinct_kmers"], {k:round(v,1) for k,v in j["roofline"]["per_kernel_ms_per_step"].items()})
else: print(open("gpurun_out/w63.log").read()[-800:])
PY
done
