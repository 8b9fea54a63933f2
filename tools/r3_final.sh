# round 3 final measurement call: full GPU suite, smoke, default bench (headline + other_modes + other_configs + CPU baselines), and the headline step
# under rocprofv3 --kernel-trace --stats (-> profiles/r03/)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -4 $OUT/smoke.log
python bench.py > $OUT/bench_n1_final.json 2> $OUT/bench_n1_final.err || { tail -5 $OUT/bench_n1_final.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/bench_n1_final.json')); print('headline', d['value'], d['roofline']['frac'], d['roofline']['kernel_avg_ms']);
[print(k, v.get('value'), (v.get('roofline') or {}).get('bound'), (v.get('roofline') or {}).get('frac'), v.get('error')) for k, v in d['other_configs'].items()]"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
head -5 $OUT/stats/*kernel_stats.csv | cut -c1-160
rm -f $OUT/stats/*kernel_trace.csv
