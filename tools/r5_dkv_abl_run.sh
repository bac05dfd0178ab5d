#!/bin/bash
# round 5 (GPU box, repo root): timing ablations of the dK/dV pass (BWD_ABL bits 8 / 16 / 32 / 64), per-kernel durations (rocprofv3)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for n in "" $ABLS; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ -n "$n" ] && lib=$root/aicity_action_amd/lib/variants/libmvit_hip_bwdabl$n.so
  for shape in "8 4 6272 1568" "8 4 6272 6272"; do
    echo "== BWD_ABL=${n:-0} attnbwd $shape: $(MVIT_HIP_LIB=$lib bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep 'attn_bwd_dkv_kernel' | cut -c60-150)"
  done
done
done
