#!/bin/bash
# round 5 (GPU box, repo root): PMC evidence for the pooling-conv kernels at the stage-3 shape -> gpurun_out/r5_pmc_pool_*.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
box=$(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 | awk '{print $NF}')
hdr() { echo "# $1"; echo "# box: ${box:-unknown} ($(hostname)); tree: $(cat $root/.r5_commit 2>/dev/null); counters averaged per launch (rocprofv3 --pmc, two SQ passes + FETCH_SIZE / WRITE_SIZE passes)"; }
traffic() {   # kernel-substring, opbench args...
  k=$1; shift
  cd /tmp && export TMPDIR=/tmp
  rm -rf $root/gpurun_out/tt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/tt/f -- python3 $root/tools/opbench.py "$@" > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/tt/w -- python3 $root/tools/opbench.py "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/tt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$k" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc["FETCH_SIZE"] and acc["WRITE_SIZE"]:
    f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 2048.0
    w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024.0
    print("HBM-side bytes per launch: fetch %.1f MB (FETCH_SIZE KiB x 2: gfx950 tallies 128-B requests at 64 B), write %.1f MB" % (f / 1e6, w / 1e6))
PY
  rm -rf $root/gpurun_out/tt
  cd $root
}
{ hdr "pool_march_kernel<bf16, 1, 0>: q pooling conv + LayerNorm, stride 1, B=8 heads=4 T=8 28x28 (stage 3); algorithmic bytes 38.5 MB in + 38.5 MB out"
  python3 tools/opbench.py pool 8 4 8 28 28 1 50 | tail -1
  tools/pmc.sh r5pmc_a pool_march -- pool 8 4 8 28 28 1 20
  traffic pool_march pool 8 4 8 28 28 1 20; } > gpurun_out/r5_pmc_pool_march_q_28.txt 2>&1
{ hdr "pool_march_kernel<bf16, 2, 0> pair launch: k and v pooling conv + LayerNorm, stride 2, B=8 heads=4 T=8 28x28 -> 14x14 (stage 3); algorithmic bytes 77 MB in + 19.3 MB out"
  python3 tools/opbench.py poolkv 8 4 8 28 28 50 | head -1
  tools/pmc.sh r5pmc_b pool_march -- poolkv 8 4 8 28 28 20
  traffic pool_march poolkv 8 4 8 28 28 20; } > gpurun_out/r5_pmc_pool_march_kv_28.txt 2>&1
{ hdr "pool_wgrad_march_kernel<1>: weight gradient of the q pooling conv, stride 1, B=8 heads=4 T=8 28x28 (inside mvit_pool_conv_ln_bwd_saved)"
  python3 tools/opbench.py poolbwd 8 4 8 28 28 1 30 | tail -1
  tools/pmc.sh r5pmc_c pool_wgrad_march -- poolbwd 8 4 8 28 28 1 20
  traffic pool_wgrad_march poolbwd 8 4 8 28 28 1 20; } > gpurun_out/r5_pmc_pool_wgrad_march_28.txt 2>&1
rm -rf gpurun_out/r5pmc_a gpurun_out/r5pmc_b gpurun_out/r5pmc_c
