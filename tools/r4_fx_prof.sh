set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/fx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/nst_microbench.py 181 3 f16x3 > $OUT/mb.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fx/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%-70s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"].replace("void (anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $OUT/prof
