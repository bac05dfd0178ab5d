cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sb -- python3 $GRAFT_REPO_ROOT/tools/opbench.py stembwd 8 20 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/sb/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("stem", "slab", "reduce_partials")):
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/sb
