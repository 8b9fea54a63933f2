# round 4: what bounds the TCN kernels?  per-kernel averages of one c5 step with the stores / the loads / both removed (results are wrong: timing only)
cd /tmp && export TMPDIR=/tmp
for lib in default tcnp1 tcnp2 tcnp3; do
  if [ $lib = default ]; then unset MPV_LIB; else export MPV_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/tcnprobe_$lib; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>$OUT/err.log
  python3 - $lib $OUT <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[2] + "/*kernel_stats.csv")[0])))
print(sys.argv[1], " ".join("%s=%.0f" % (r["Name"].split("stg_tcn_mfma_kernel")[1].split("(")[0], float(r["AverageNs"]) / 1e3) for r in rows if "stg_tcn_mfma" in r["Name"]))
PY
  rm -f $OUT/*kernel_trace.csv
done
