# round 4: f(x) / DiffSTG / NsDiff_spatial parity tests that go through the split-f16 GEMM, then f(x) microbench + c5 / nss bench records
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4fx_${1:-a}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_nst_gpu.py tests/test_nst_spatial_gpu.py tests/test_stg_gpu.py tests/test_nss_gpu.py tests/test_full_size_gpu.py -x -q -k "not c3" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log
tail -3 $OUT/tests.log
python tools/nst_microbench.py 181 3 f16x3 2>/dev/null | tail -1 | tee $OUT/fx.txt
MPV_GEMM_BM=128 python tools/nst_microbench.py 181 3 f16x3 2>/dev/null | tail -1 | tee -a $OUT/fx.txt
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err; python -c "
import json; d=json.load(open('$OUT/c5.json')); print('c5', d['value'], 'windows/s')"
MPV_GEMM_BM=128 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c5_bm128.json 2> $OUT/c5_bm128.err; python -c "
import json; d=json.load(open('$OUT/c5_bm128.json')); print('c5 (128-row tiles everywhere)', d['value'], 'windows/s')"
python bench.py --config nss --no-cpu-baseline > $OUT/nss.json 2> $OUT/nss.err; python -c "
import json; d=json.load(open('$OUT/nss.json')); print('nss', d['value'], 'windows/s')"
