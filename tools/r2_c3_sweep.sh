# c3 (DiffusionTS) with the batch run in row blocks of different sizes: cache residency of the per-layer activations
set -o pipefail
mkdir -p gpurun_out/r2
for rows in 3200 1600 800 400 200; do
  MPV_DTS_ROWS=$rows timeout -k 10 300 python bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2/c3_rows_$rows.json 2> gpurun_out/r2/c3_rows_$rows.err || exit 1
  python3 -c "import json,sys; d=json.load(open('gpurun_out/r2/c3_rows_$rows.json')); print('rows', $rows, 'ms/window', round(d['ms_per_step']), 'TFLOP/s', round(d['roofline']['achieved'],1))"
done
