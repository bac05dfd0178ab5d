cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
python3 $root/tools/blaslt_ref.py > $root/gpurun_out/r2_blaslt_ref.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r2_blaslt_trace -- python3 $root/tools/blaslt_ref.py > /dev/null 2>&1
python3 - <<'PY' > $root/gpurun_out/r2_blaslt_kernels.txt 2>&1
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(root + "/gpurun_out/r2_blaslt_trace/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:200], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    e = acc.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += d
for k, (n, t) in acc.items():
    print("%6d calls avg %8.1f us grid %s wg %s lds %s vgpr %s  %s" % (n, t / n / 1e3, k[1], k[2], k[3], k[4], k[0]))
PY
rm -rf $root/gpurun_out/r2_blaslt_trace
