# round 3: DiffusionTS parity tests + c3 bench (one window) + kernel stats.  usage: bash tools/r3_c3.sh <tag> [env for the bench, e.g. MPV_DTS_GEMM=f32]
set -o pipefail
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c3_$TAG
mkdir -p $OUT
python -m pytest tests/test_dts_gpu.py tests/test_full_size_gpu.py tests/test_harness_families_gpu.py -x -q -s -k "dts or c3 or diffusionts" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log
grep -v amdgpu.ids $OUT/tests.log | grep -i "DiffusionTS\|C3\|passed\|failed\|pytest rc\|error" | tail -40
env "$@" python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err || { tail -5 $OUT/c3.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/c3.json')); print('c3 s/window', d['ms_per_step']/1e3, 'frac', d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
head -16 $OUT/stats/*kernel_stats.csv | cut -c1-150
rm -f $OUT/stats/*kernel_trace.csv
