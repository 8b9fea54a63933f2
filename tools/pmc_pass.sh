# per-kernel HBM traffic of one bench step of c2 (headline) / c5 / c3 / c4 / nss: three separate rocprofv3 runs each (stats, FETCH_SIZE, WRITE_SIZE);
# the JSON it writes carries the sha256 of the library measured (tools/pmc_traffic.py) -- copy it to profiles/<round>/ for THAT binary only
# usage (on the GPU box): bash tools/pmc_pass.sh c2|c5|c3|c4|nss
set -o pipefail
CFG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $CFG in
  c2) ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes --no-other-configs"; WIN=181 ;;
  c5) ARGS="--config c5 --steps 1 --warmup 0 --stg-windows 4 --no-cpu-baseline"; WIN=4 ;;
  c3) ARGS="--config c3 --steps 1 --warmup 0 --dts-windows 1 --no-cpu-baseline"; WIN=1 ;;
  c4) ARGS="--config c4 --steps 1 --warmup 0 --no-cpu-baseline"; WIN=181 ;;
  nss) ARGS="--config nss --no-cpu-baseline"; WIN=12 ;;      # NsDiff_spatial family record: 1 warm-up + 2 timed steps of 4 windows
esac
PMCENV=""
[ $CFG = nss ] && export MPV_BENCH_NSS_NO_FX=1      # the native-f(x) leg of the record would double the launches counted against its 12 windows
[ $CFG = c3 ] && PMCENV=10      # c3: the counter passes run a 10-step plan (same launches, 10x fewer: a counter pass of the 36 k launches of the full plan takes > 7 min)
if [ -z "$SKIP_STATS" ]; then
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err || exit 1
fi
echo "stats pass done"
[ -n "$PMCENV" ] && export MPV_BENCH_DTS_STEPS=$PMCENV
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || exit 1
  echo "$c pass done"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $CFG $WIN $OUT $OUT/pmc_traffic_$CFG.json "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE, three separate runs of: python3 bench.py $ARGS (tools/pmc_pass.sh)" > $OUT/summary.txt || exit 1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv
# raw traces are large: keep only the aggregates
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cat $OUT/summary.txt
