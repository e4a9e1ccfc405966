# Regenerates the raw material of profiles/ on the GPU box (one MI355X):
#   gpurun -- 'bash tools/refresh_profiles.sh r01'
# then, back in the repo:  python tools/make_profiles.py r01
# rocprofv3 is given the program itself after `--`; the PMC passes are separate runs.
T=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k63 --no-ingest --no-strong --no-next-rows > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contigs --no-k63 --no-strong --no-next-rows > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contigs --no-k63 --no-strong --no-next-rows > $O/write.log 2>&1 || exit 1
ls -R $O | head -40
