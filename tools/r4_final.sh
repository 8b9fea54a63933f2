# round 4 final measurement call 1: full GPU suite, smoke, the default bench (the driver's command), the headline step under rocprofv3 --kernel-trace --stats
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -4 $OUT/smoke.log
MPV_BENCH_RECORDS=$OUT/bench_n1_final_records.jsonl python bench.py > $OUT/bench_n1_final.json 2> $OUT/bench_n1_final.err || { tail -5 $OUT/bench_n1_final.err; exit 1; }
echo "line bytes $(wc -c < $OUT/bench_n1_final.json)"; cat $OUT/bench_n1_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
head -5 $OUT/stats/*kernel_stats.csv | cut -c1-160
rm -f $OUT/stats/*kernel_trace.csv
