# per-pass wall times of the extend stage at config-2 size (RFX_TRACE), then the kernel table of the same run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RFX_TRACE=1 timeout -k 10 300 python3 $R/tools/prof_count.py --gbp 5 --steps 1 --assemble > $R/gpurun_out/trace_asm.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_asm -- python3 $R/tools/prof_count.py --gbp 5 --steps 1 --assemble > $R/gpurun_out/prof_asm.log 2>&1
