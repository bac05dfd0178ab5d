#!/bin/bash
# round 5 (GPU box, repo root): where do the dK/dV pass's query-tile streams come from?  HBM fetch / L2 hit-miss counters per launch (separate PMC passes)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for shape in "8 4 6272 6272" "8 4 6272 1568"; do
  rm -rf $root/gpurun_out/dkvl2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/dkvl2/f -- python3 $root/tools/opbench.py attnbwd $shape 5 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $root/gpurun_out/dkvl2/h -- python3 $root/tools/opbench.py attnbwd $shape 5 > /dev/null 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $root/gpurun_out/dkvl2/t -- python3 $root/tools/opbench.py attnbwd $shape 5 > /dev/null 2>&1
  echo "== attnbwd $shape"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/dkvl2/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("attn_bwd_dkv_kernel","attn_bwd_dq_kernel"):
            if k in r["Kernel_Name"]: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c,v in sorted(acc[k].items()): print("%-22s %-30s %16.0f  (n=%d)" % (k, c, sum(v)/len(v), len(v)))
PY
done
rm -rf $root/gpurun_out/dkvl2
