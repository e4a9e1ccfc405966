# every reflexiv_amd/lib_*.so.bak in turn through tools/ab_k63.py
for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  echo "== $f"; timeout -k 10 200 python tools/ab_k63.py 2>&1 | grep -v "^level\|^leaves\|one-sweep\|amdgpu.ids"
done
