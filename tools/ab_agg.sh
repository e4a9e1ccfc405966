cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_vectors.py -x -q > gpurun_out/agg_parity.log 2>&1 || { tail -20 gpurun_out/agg_parity.log; exit 1; }
tail -2 gpurun_out/agg_parity.log
for d in 0 64; do
echo "config 2, RFX_LEAF_DBG=$d"; RFX_LEAF_DBG=$d python tools/prof_count.py --gbp 5 --steps 2 2>&1 | grep -o "'leaf': ([0-9.]*\|kept [0-9]*" 
done
for d in 0 64 512; do
echo "human-scale share, RFX_LEAF_DBG=$d"; RFX_LEAF_DBG=$d timeout -k 10 400 python tools/prof_count.py --gbp 18.75 --genome 400000000 --cover 2 --steps 2 2>&1 | grep -o "'leaf': ([0-9.]*\|kept [0-9]*" 
done
