cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_vectors.py -x -q > gpurun_out/wagg_parity.log 2>&1 || { tail -20 gpurun_out/wagg_parity.log; exit 1; }
tail -2 gpurun_out/wagg_parity.log
echo "agg on";  timeout -k 10 300 python tools/k63_stats.py 2>&1 | grep -v amdgpu
echo "agg off"; RFX_WIDE_NOAGG=1 timeout -k 10 300 python tools/k63_stats.py 2>&1 | grep -v amdgpu
echo "stats"; RFX_WIDE_STATS=1 timeout -k 10 300 python tools/k63_stats.py 2>&1 | grep "wide rec" | sort | uniq -c | sort -rn | head -4
