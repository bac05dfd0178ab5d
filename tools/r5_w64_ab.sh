#!/bin/bash
# round 5 (GPU box, repo root): attn_fwd_w64_kernel alone, tree vs the variant library of the previous kernel (interleaved), then stamps
root=${GRAFT_REPO_ROOT:-$(pwd)}
V=$root/aicity_action_amd/lib/variants
for rep in 1 2 3; do
for n in new old; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ $n = old ] && lib=$V/libmvit_hip_w64old.so
  for shape in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 1568" "8 8 1568 1568" "8 4 6272 6272"; do
    echo "$n: $(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py attn $shape 30 2>/dev/null | tail -1)"
  done
done
done
lib=$V/libmvit_hip_w64stamp.so
[ -f $lib ] && for shape in "8 4 6272 1568" "8 1 100352 1568"; do W_STAMP=1 MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py attn $shape 10 2>/dev/null | tail -3; done
