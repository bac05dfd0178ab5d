# tools/kseq.sh: the launch sequence of one eval forward around every __amd_rocclr_copyBuffer (what issues the small copies?)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kseq
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kseq -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kseq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:50] for r in rows]
# the last 400 launches = the end of the last forward
tail = rows[-420:]
for r in tail[:160]:
    print(r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kseq
