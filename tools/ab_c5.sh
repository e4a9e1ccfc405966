for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  echo "== $f"; timeout -k 10 300 python bench.py --gbp 18.75 --genome 400000000 --steps 2 --warmup 1 --no-contigs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['roofline']['per_kernel_ms_per_step'].items()})"
done
