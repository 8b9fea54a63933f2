# round 2, final measurements from the shipped binary: full GPU suite, smoke, the three benches, 2-rank rehearsal, profiles
set -o pipefail
OUT=gpurun_out/r2final
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log; [ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err || exit 1
timeout -k 10 500 python bench.py --config c3 --steps 1 --warmup 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
timeout -k 10 400 python bench.py --config c5 --steps 2 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err || exit 1
python tools/nst_microbench.py 181 3 f16x3 > $OUT/nst_microbench.log && python tools/nst_microbench.py 181 2 f32 >> $OUT/nst_microbench.log
bash tools/r2_profiles.sh > $OUT/profiles.log 2>&1 || { tail -5 $OUT/profiles.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -o bench -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/prof_c3.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c5 -o bench -- python3 $R/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/prof_c5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_nst -o nst -- python3 $R/tools/nst_microbench.py 181 1 > $R/$OUT/prof_nst.log 2>&1 || exit 1
cd $R
python - <<'PY'
import json
for n in ("bench_n1", "bench_n2_gloo", "bench_c3", "bench_c5"):
    r = json.load(open("gpurun_out/r2final/%s.json" % n))
    print(n, round(r["value"], 4), r["unit"], "frac", round(r["roofline"]["frac"], 3), "kernel ms", r["roofline"].get("kernel_avg_ms"))
r = json.load(open("gpurun_out/r2final/bench_n1.json"))
print("full_nsdiff", r["full_nsdiff"]["value"], "fx ms", r["full_nsdiff"]["fx_ms_per_step"])
print(open("gpurun_out/r2final/nst_microbench.log").read())
PY
