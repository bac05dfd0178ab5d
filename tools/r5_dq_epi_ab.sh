#!/bin/bash
# round 5 (GPU box, repo root): dQ pass with the lane-swap epilogue (tree) vs the previous one (variant bwdold): per-kernel durations, then the train step
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for n in tree bwdold; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ $n != tree ] && lib=$root/aicity_action_amd/lib/variants/libmvit_hip_$n.so
  for shape in "8 4 6272 1568" "8 1 100352 1568"; do
    echo "== $n attnbwd $shape"
    MVIT_HIP_LIB=$lib bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep "attn_bwd_dq"
  done
done
done
