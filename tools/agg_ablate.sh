cd $GRAFT_REPO_ROOT
for d in 0 1 2 64 65 66 32; do
  echo "RFX_LEAF_DBG=$d"; RFX_LEAF_DBG=$d python tools/prof_count.py --gbp 5 --steps 2 2>&1 | grep -o "leaf waves.*\|'leaf': ([0-9.]*" 
done
