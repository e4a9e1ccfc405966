# Round 4's profile material in one call on the GPU box:  gpurun --timeout 1200 -- 'bash tools/r04.sh'
# then here:  python tools/make_profiles.py r04 ; python tools/sq_summary.py r04pmc r04 ; cp the force-dist lines
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1 || { tail -5 gpurun_out/r04_refresh.log; exit 1; }
cd $GRAFT_REPO_ROOT
python bench.py --force-dist --gbp 6.25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd31.json 2> gpurun_out/prof_r04/fd31.err &&
python bench.py --force-dist --gbp 6.25 --k 63 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r04/fd63.json 2> gpurun_out/prof_r04/fd63.err &&
bash tools/pmc_leaf.sh r04pmc > gpurun_out/r04_pmc.log 2>&1
tail -2 gpurun_out/prof_r04/bench.json | cut -c1-600
cd $GRAFT_REPO_ROOT && bash tools/pmc_wide.sh r04w > gpurun_out/r04_pmcw.log 2>&1
