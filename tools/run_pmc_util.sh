# matrix-pipe / vector-issue utilisation of the chain kernel: separate --pmc passes over one bench launch each
set -o pipefail
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_util$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes > $GRAFT_REPO_ROOT/gpurun_out/pmc_util$i.log 2>&1 || { echo "pass $i failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_util$i.log; }
done
python3 - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]
out = {}
for f in sorted(glob.glob(root + "/gpurun_out/pmc_util*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "nsdiff_chain_f16" in r["Kernel_Name"]:
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(json.dumps(out, indent=1))
json.dump(out, open(root + "/gpurun_out/pmc_util.json", "w"), indent=1)
PY
