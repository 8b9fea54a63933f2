# round 3: DiffSTG parity tests + c5 bench + kernel stats (one GPU call).  usage: bash tools/r3_c5.sh <tag>
set -o pipefail
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c5_$TAG
mkdir -p $OUT
python -m pytest tests/test_stg_gpu.py tests/test_nss_gpu.py tests/test_full_size_gpu.py -x -q -s -k "stg or nss or c5" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log
grep -v amdgpu.ids $OUT/tests.log | grep -i "passed\|failed\|pytest rc\|error" | tail -8
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err || { tail -5 $OUT/c5.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/c5.json')); print('c5 windows/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
head -14 $OUT/stats/*kernel_stats.csv | cut -c1-140
rm -f $OUT/stats/*kernel_trace.csv
