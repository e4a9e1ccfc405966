cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/agg_parity.log 2>&1 || { tail -20 gpurun_out/agg_parity.log; exit 1; }
tail -2 gpurun_out/agg_parity.log
for d in 128 384 0 256 0 256; do
echo "RFX_LEAF_DBG=$d"; RFX_LEAF_DBG=$d python tools/prof_count.py --gbp 5 --steps 2 2>&1 | grep -o "record table.*\|'leaf': ([0-9.]*\|kept [0-9]*" 
done
