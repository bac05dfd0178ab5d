#!/bin/bash
# round 5: pooling-conv op bench, baseline library (variants/libmvit_hip_r5base.so = HEAD before the LDS-read fix) vs the tree's, same box,
# interleaved.  usage (GPU box, repo root): tools/r5_pool_ab.sh > gpurun_out/r5_pool_ab.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
base=$root/aicity_action_amd/lib/variants/libmvit_hip_r5base.so
for rep in 1 2; do
  for shape in "8 1 8 112 112 1" "8 2 8 56 56 1" "8 4 8 28 28 1" "8 8 8 14 14 1" "8 4 8 28 28 2" "8 2 8 56 56 2"; do
    echo "base: $(MVIT_HIP_LIB=$base python3 $root/tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
    echo "new : $(python3 $root/tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
  done
  echo "base: $(MVIT_HIP_LIB=$base python3 $root/tools/opbench.py poolkv 8 4 8 28 28 50 2>/dev/null | head -1)"
  echo "new : $(python3 $root/tools/opbench.py poolkv 8 4 8 28 28 50 2>/dev/null | head -1)"
done
