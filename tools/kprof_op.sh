# tools/kprof_op.sh <opbench args...>: rocprofv3 kernel durations of one opbench run (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kop
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kop -- python3 $GRAFT_REPO_ROOT/tools/opbench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kop/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kop
