#!/bin/bash
# round 5 (GPU box, repo root): fp16 build of the pooling march with explicit conversions (tree) vs v_fma_mix_f32 taps (variant f16mix), op level
# (the fp16 library is loaded in the bf16 slot of opbench: same kernels, operand bit patterns read as fp16), then the fp16 forward
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for n in tree mix; do
  lib=$root/aicity_action_amd/lib/libmvit_hip_f16.so; [ $n = mix ] && lib=$root/aicity_action_amd/lib/variants/libmvit_hip_f16mix.so
  for shape in "8 4 8 28 28 1" "8 1 8 112 112 1" "8 2 8 56 56 1"; do
    echo "$n: $(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
  done
  echo "$n: $(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py poolkv 8 4 8 28 28 50 2>/dev/null | tail -1)"
done
done
