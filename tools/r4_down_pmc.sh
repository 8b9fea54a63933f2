# round 4: LDS bank conflicts of the downsampling kernel's phases (tools/ubench/gemm_h3 GEMM_DOWN_PROBE dispatches: PROBE 14 = operand reads only,
# 13 = activation split + stores only, 11 = weight stores only)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4down_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GEMM_DOWN=1 GEMM_DOWN_PROBE=${1:-5}
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p -o run -- $GRAFT_REPO_ROOT/build_variants/gemm_h3 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4down_pmc"
f = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per.setdefault((r["Dispatch_Id"], k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = collections.Counter()
for (d, k), c in per.items():
    seen[k] += 1
    if seen[k] != 2: continue
    print("%-34s " % k + " ".join("%s=%.4g" % (a.replace("SQ_", ""), v) for a, v in c.items()))
PY
rm -rf $OUT/p
