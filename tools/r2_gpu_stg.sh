set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 400 python bench.py --config c5 --steps 2 --warmup 1 > gpurun_out/r2/bench_c5.json 2> gpurun_out/r2/bench_c5.err; echo "bench c5 rc=$?"; cut -c1-500 gpurun_out/r2/bench_c5.json; tail -4 gpurun_out/r2/bench_c5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_c5 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2/bench_c5_prof.log 2>&1; echo "prof rc=$?"
head -16 $GRAFT_REPO_ROOT/gpurun_out/r2/prof_c5/*kernel_stats.csv
