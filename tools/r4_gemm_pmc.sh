# round 4: where does the split-f16 GEMM's k-loop lose its time?  SQ counters of the ubench dispatches (tools/ubench/gemm_h3), two passes.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4gemm_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -o run -- $GRAFT_REPO_ROOT/tools/ubench/gemm_h3 > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -o run -- $GRAFT_REPO_ROOT/tools/ubench/gemm_h3 > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4gemm_pmc"
for p in ("p1", "p2"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % p, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    per = collections.OrderedDict()
    for r in rows:
        key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
        per.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = collections.Counter()
    for (d, k, g), c in per.items():
        seen[(k, g)] += 1
        if seen[(k, g)] != 2: continue            # second dispatch of each (kernel, grid): warm
        print(p, k, "grid", g, " ".join("%s=%.4g" % (n, v) for n, v in c.items()))
PY
