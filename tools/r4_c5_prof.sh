# round 4: c5 bench value + per-kernel stats for the in-tree library and build_variants/<name>.so given as arguments ("default" = in-tree)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c5prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],3), 'windows/s')"
  rm -rf $OUT/$lib; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python3 - $OUT/$lib <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])))
for r in rows[:12]:
    print("    %-70s calls %5s avg %8.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -f $OUT/$lib/*kernel_trace.csv
done
