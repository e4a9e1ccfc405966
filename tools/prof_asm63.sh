# kernel trace of the whole k = 63 bench command (count + counter_to_asm + assemble_w), one step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_asm63 -- python3 $R/bench.py --k 63 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_asm63.log 2>&1
