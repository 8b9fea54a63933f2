# round 4: GPU suite + the driver's two commands (smoke, default bench); outputs under gpurun_out/r4check
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4check}
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
if [ -z "$NO_BENCH" ]; then
timeout -k 10 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc $? line bytes $(wc -c < $OUT/bench_n1.json)"
cp $GRAFT_REPO_ROOT/gpurun_out/bench_records.jsonl $OUT/bench_records.jsonl 2>/dev/null
cat $OUT/bench_n1.json
fi
