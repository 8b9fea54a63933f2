# round 4: SQ / LDS counters of one c5 step (4 windows), summed per kernel name: are the TCN kernels' LDS reads conflict-free, how busy are the pipes?
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c5sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 0 --stg-windows 4 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o run -- $CMD > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES --output-format csv -d $OUT/p2 -o run -- $CMD > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4c5sq"
for p in ("p1", "p2"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % p, recursive=True)[0]
    per = collections.OrderedDict()
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
        per.setdefault(k, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1
    for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_LDS_IDX_ACTIVE", 0)))[:12]:
        print(p, "%-40s n=%5d " % (k, n[k]) + " ".join("%s=%.4g" % (a.replace("SQ_", ""), v) for a, v in c.items()))
PY
rm -rf $OUT/p1 $OUT/p2
