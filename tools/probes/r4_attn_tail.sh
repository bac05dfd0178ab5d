#!/bin/bash
# stage-3 attention forward (B=8, 4 heads, Lq 6272, Lk 1568) with / without the key-split ragged tile: kernel-level times (rocprofv3) + in-model forward
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sp in 1 0; do
  export MVIT_ATT_TAIL_SPLIT=$sp
  rm -rf $root/gpurun_out/att_tail_$sp
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/att_tail_$sp -- python3 $root/tools/attn_tail_bench.py > $root/gpurun_out/att_tail_$sp.log 2>&1
  echo "== MVIT_ATT_TAIL_SPLIT=$sp"; grep -v amdgpu.ids $root/gpurun_out/att_tail_$sp.log
  python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/att_tail_$sp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn_fwd" in r["Name"]:
        print("   %-70s calls %4d avg %9.1f us" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
