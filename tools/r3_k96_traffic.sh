#!/bin/bash
# GPU box: HBM bytes per launch of linear_k96_kernel (PMC, separate passes; FETCH_SIZE KiB x2 on gfx950, WRITE_SIZE KiB x1) against the algorithmic bytes
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for shape in "802816 384 96 b" "802816 288 96 b" "802816 576 96 b"; do
  rm -rf $root/gpurun_out/k96t
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/k96t/f -- python3 $root/tools/opbench.py gemm $shape 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/k96t/w -- python3 $root/tools/opbench.py gemm $shape 5 > /dev/null 2>&1
  python3 - "$shape" <<PY
import csv, glob, sys, collections
M, N, K, _ = sys.argv[1].split()
M, N, K = int(M), int(N), int(K)
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/k96t/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_k96" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch = sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"])) * 2048.0
write = sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"])) * 1024.0
alg_r, alg_w = M * K * 2 + N * K * 2 + N * 4, M * N * 2
print("linear_k96 M=%d N=%d K=%d: fetch %.1f MB (algorithmic %.1f), write %.1f MB (algorithmic %.1f), launches sampled %d" % (M, N, K, fetch / 1e6, alg_r / 1e6, write / 1e6, alg_w / 1e6, len(acc["FETCH_SIZE"])))
PY
done
rm -rf $root/gpurun_out/k96t
