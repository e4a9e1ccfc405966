cd $GRAFT_REPO_ROOT
RFX_LEAF_DBG=128 python tools/prof_count.py --gbp 5 --steps 1 > gpurun_out/agg_stat.log 2>&1; tail -5 gpurun_out/agg_stat.log
