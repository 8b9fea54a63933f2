# round 4: DiffusionTS parity tests, then per-kernel times of the c3 bench on a 10-step plan for the libraries / env settings given
# usage: bash tools/r4_c3.sh <tag> <kernel pattern> spec...   (spec = default | <variant>[:ENV=V,...])
set -o pipefail
TAG=$1; PAT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c3_$TAG
mkdir -p $OUT
if [ -z "$NO_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_dts_gpu.py tests/test_full_size_gpu.py tests/test_harness_families_gpu.py -x -q -s -k "dts or c3 or diffusionts" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log
grep -v amdgpu.ids $OUT/tests.log | grep -i "DiffusionTS\|C3\|passed\|failed\|pytest rc\|error" | tail -40
fi
bash tools/debug/r3_c3_variants.sh "$PAT" "$@" 2>&1 | tee $OUT/variants.txt
