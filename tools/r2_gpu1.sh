# round 2, GPU call 1: parity tests, smoke, a short bench (N=1), the N=2 self-launch rehearsed over gloo on the one GPU
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r2/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --cpu-windows 1 > gpurun_out/r2/bench1.json 2> gpurun_out/r2/bench1.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r2/bench1.json; tail -5 gpurun_out/r2/bench1.err
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --max-windows 16 > gpurun_out/r2/bench_n2_gloo.json 2> gpurun_out/r2/bench_n2_gloo.err; echo "bench n2 rc=$?"; cut -c1-400 gpurun_out/r2/bench_n2_gloo.json; tail -3 gpurun_out/r2/bench_n2_gloo.err
