cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cnt -- python3 $R/tools/prof_count.py --gbp 5 --steps ${STEPS:-1} > $R/gpurun_out/prof_cnt.log 2>&1
