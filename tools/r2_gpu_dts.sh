# round 2: DiffusionTS parity tests + the c3 bench, then its per-kernel rocprof stats
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_dts_gpu.py -m gpu -q -s > gpurun_out/r2/dts_tests.log 2>&1; rc=$?; echo "dts tests rc=$rc"
grep -n "DiffusionTS\|passed\|failed" gpurun_out/r2/dts_tests.log | head -40
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err; echo "bench c3 rc=$?"; cut -c1-600 gpurun_out/r2/bench_c3.json; tail -3 gpurun_out/r2/bench_c3.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_c3 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2/bench_c3_prof.log 2>&1; echo "prof rc=$?"
head -25 $GRAFT_REPO_ROOT/gpurun_out/r2/prof_c3/*kernel_stats.csv
