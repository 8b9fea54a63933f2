# round 4: DiffSTG / NsDiff_spatial parity tests, then the c5 / nss bench records for the env settings given (one per argument, "-" = none)
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c5_$TAG
mkdir -p $OUT
if [ -z "$NO_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_stg_gpu.py tests/test_nss_gpu.py tests/test_nst_spatial_gpu.py tests/test_full_size_gpu.py tests/test_harness_families_gpu.py -x -q -k "not c3 and not diffusionts" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log
tail -3 $OUT/tests.log
fi
for spec in "$@"; do
  envs=""; [ "$spec" != "-" ] && envs=$(echo $spec | tr ',' ' ')
  env $envs python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c5_$spec.json 2> $OUT/c5_$spec.err; python -c "
import json; d=json.load(open('$OUT/c5_$spec.json')); print('c5 [$spec]', round(d['value'],3), 'windows/s')"
  env $envs python bench.py --config nss --no-cpu-baseline > $OUT/nss_$spec.json 2> $OUT/nss_$spec.err; python -c "
import json; d=json.load(open('$OUT/nss_$spec.json')); print('nss [$spec]', round(d['value'],3), 'windows/s')"
done
