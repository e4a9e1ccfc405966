# as ab_many.sh, on the k = 63 count (RFX_WIDE_PRESPLIT from $PS, default the library's)
for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  for ps in ${PS:-2600}; do
  echo "== $f presplit $ps"; RFX_WLEAF_PER_CU=${WPC:-1} RFX_WIDE_PRESPLIT=$ps timeout -k 10 100 python bench.py --k 63 --steps 2 --warmup 1 --no-contigs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['roofline']['per_kernel_ms_per_step'].items()})"
  done
done
