root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_bwd_ops.py tests/test_hip_train.py -q -m gpu -x -k "attention or train" 2>&1 | tail -3 > gpurun_out/r2_t32_tests.log
cd /tmp && export TMPDIR=/tmp MVIT_NO_SIDE_STREAM=1
for shp in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 6272" "8 8 1568 1568"; do
for mode in rows flat; do
  if [ $mode = rows ]; then export MVIT_ATT_DELTA_ROWS=1; else unset MVIT_ATT_DELTA_ROWS; fi
  rm -rf $root/gpurun_out/ab_d
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ab_d -- python3 $root/tools/opbench.py attnbwd $shp 20 > /dev/null 2>&1
  echo "== $mode $shp"
  python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/ab_d/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "delta" in r["Name"]: print("   %-50s calls %4d avg %9.1f us" % (r["Name"][:50], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done; done > $root/gpurun_out/r2_t32_delta.txt 2>&1
rm -rf $root/gpurun_out/ab_d
