# A/B of builds on the extend stage: every reflexiv_amd/lib_*.so.bak in turn, counts -> contigs wall time of the bench (two rounds)
for r in 1 2; do
for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  echo "== $f"; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest ${K63:---no-k63} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['contigs']['wall_ms_from_counts'],2), round(j.get('k63',{}).get('contigs',{}).get('wall_ms_from_counts',0),2))"
done; done
