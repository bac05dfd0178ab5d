#!/bin/bash
# round 5 (GPU box, repo root): ablation timings of attn_fwd_w64_kernel alone (bf16 build; variants from tools/r5_w64_abl.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
V=$root/aicity_action_amd/lib/variants
for rep in 1 2; do
for n in "" $ABLS; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ -n "$n" ] && lib=$V/libmvit_hip_w64abl$n.so
  [ -f $lib ] || continue
  for shape in "8 4 6272 1568" "8 1 100352 1568"; do
    echo "W_ABL=${n:-0}: $(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py attn $shape 30 2>/dev/null | tail -1)"
  done
done
done
lib=$V/libmvit_hip_w64stamp.so
[ -f $lib ] && for shape in "8 4 6272 1568" "8 1 100352 1568"; do W_STAMP=1 MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py attn $shape 10 2>/dev/null | tail -3; done
