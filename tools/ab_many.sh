# A/B of several BUILDS in one gpurun call: every reflexiv_amd/lib_*.so.bak is copied over the library in turn and
# the count-stage bench line is run (two rounds, interleaved)
for r in 1 2; do
for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  echo "== $f"; timeout -k 10 100 python bench.py --steps 4 --warmup 1 --no-contigs --no-cpu-baseline --no-k63 --no-ingest 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['roofline']['per_kernel_ms_per_step'].items()})"
done; done
