# A/B of several BUILDS (reflexiv_amd/lib_*.so.bak) on the human-scale share (18.75 Gbp of a 400 Mbp genome, -cover 2)
cd $GRAFT_REPO_ROOT
for f in reflexiv_amd/lib_*.so.bak; do
  cp $f reflexiv_amd/libreflexiv_hip.so
  echo "== $f"; timeout -k 10 400 python tools/prof_count.py --gbp 18.75 --genome 400000000 --cover 2 --steps 2 2>&1 | grep -o "'leaf': ([0-9.]*\|kept [0-9]*"
done
