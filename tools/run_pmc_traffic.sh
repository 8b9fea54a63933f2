# two separate --pmc passes over one bench launch each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(root + "/gpurun_out/pmc_%s/*counter_collection.csv" % c)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0][-40:]
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    out[c] = {k: {"launches": len(v), "sum_KB": sum(v)} for k, v in per.items()}
print(json.dumps(out, indent=1))
json.dump(out, open(root + "/gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
PY
