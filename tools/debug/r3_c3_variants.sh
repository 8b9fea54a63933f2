#!/bin/bash
# DiffusionTS c3, 10-step plan under rocprofv3 --stats for the default library and variants: per-kernel average durations.
#   usage: tools/debug/r3_c3_variants.sh <pattern of kernels to print> lib1[:ENV=V,...] lib2 ...   (lib "default" = the in-tree library)
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/c3v_$(echo $spec | tr ':=,' '___')
  rm -rf $OUT; mkdir -p $OUT
  env MPV_BENCH_DTS_STEPS=10 $envs rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log || { echo "$spec FAILED"; tail -3 $OUT/err.log; continue; }
  python3 - "$spec" $OUT "$PAT" <<'PY'
import csv, glob, sys, re
spec, out, pat = sys.argv[1:4]
rows = list(csv.DictReader(open(glob.glob(out + "/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "dts" in r["Name"])
print("%-40s dts kernels total %.1f ms" % (spec, tot / 1e6))
for r in rows:
    if re.search(pat, r["Name"]):
        print("    %-60s calls %5s avg %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  rm -f $OUT/*kernel_trace.csv
done
