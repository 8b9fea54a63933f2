set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/bench_f16.json
NF=2 timeout -k 10 120 python tools/chain_microbench.py 8 2 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f16 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?"
ls $GRAFT_REPO_ROOT/gpurun_out/prof_f16
