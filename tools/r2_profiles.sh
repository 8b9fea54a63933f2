# round 2 profiles of the headline bench (c2) from the shipped binary: kernel stats + HBM traffic (separate --pmc passes, as the
# microarchitecture guide prescribes), for the fused-summary step and for the sample-writing step
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
for mode in fused store; do
  extra=""; [ $mode = store ] && extra="--store-samples"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${mode}_$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes $extra > $OUT/pmc_${mode}_$c.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import csv, glob, json, os
out_dir = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2prof"
res = {}
for mode in ("fused", "store"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(out_dir + "/pmc_%s_%s/*counter_collection.csv" % (mode, c))[0]
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1]
                per.setdefault(k, []).append(float(r["Counter_Value"]))
        res.setdefault(mode, {})[c] = {k: {"launches": len(v), "sum_KB": sum(v)} for k, v in per.items()}
json.dump(res, open(out_dir + "/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
head -8 $OUT/stats/*kernel_stats.csv
