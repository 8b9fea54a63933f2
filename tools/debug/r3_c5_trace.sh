#!/bin/bash
# per-dispatch durations of the split-f16 GEMM inside one c5 step, grouped by grid size (= shape)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_h3" in r["Kernel_Name"] or "stg_gemm" in r["Kernel_Name"]:
        g[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s grid %8s  n %5d  avg %8.1f us  total %8.1f ms  %5.1f %%" % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot))
PY
rm -f $OUT/*kernel_trace.csv
