for r in 1 2; do
for v in B A; do
  if [ $v = A ]; then cp reflexiv_amd/lib_A.so.bak /tmp/cur.so; else cp reflexiv_amd/lib_B.so.bak /tmp/cur.so; fi
  cp /tmp/cur.so reflexiv_amd/libreflexiv_hip.so
  echo "== $v"; timeout -k 10 100 python bench.py --steps 4 --warmup 1 --no-contigs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['roofline']['per_kernel_ms_per_step'].items()})"
done; done
