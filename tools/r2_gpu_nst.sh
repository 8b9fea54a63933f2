# round 2: f(x) backbone parity tests + work-queue claim A/B of the headline bench, one GPU call
set -o pipefail
mkdir -p gpurun_out/r2n
timeout -k 10 600 python -m pytest tests/test_nst_gpu.py tests/test_tmdm.py -m gpu -x -q > gpurun_out/r2n/tests.log 2>&1; rc=$?; tail -25 gpurun_out/r2n/tests.log; [ $rc = 0 ] || exit $rc
for c in 1 8 32; do
  MPV_CHAIN_CLAIM=$c timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes > gpurun_out/r2n/bench_claim$c.json 2> gpurun_out/r2n/bench_claim$c.err || exit 1
  python - <<PY
import json; r = json.load(open("gpurun_out/r2n/bench_claim$c.json")); print("claim $c", r["value"], r["roofline"]["kernel_avg_ms"])
PY
done
