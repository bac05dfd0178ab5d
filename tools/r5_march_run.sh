#!/bin/bash
# round 5 (GPU box, repo root): ablation timings of pool_march at the stage-3 q-pool shape, then PMC of the product kernels
root=${GRAFT_REPO_ROOT:-$(pwd)}
V=$root/aicity_action_amd/lib/variants
echo "== ablations, pool B=8 h=4 THW=8x28x28 s=1 (us) =="
for n in "" 1 2 4 8 16 3 $EXTRA_VARIANTS; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ -n "$n" ] && lib=$V/libmvit_hip_march$n.so
  [ -f $lib ] || continue
  for shape in "8 4 8 28 28 1" "8 1 8 112 112 1"; do
    echo "MARCH_ABL=${n:-0}: $(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py pool $shape 50 2>/dev/null | tail -1)"
  done
done
if [ -z "$NO_PMC" ]; then
echo "== PMC pool_march_kernel<bf16,1,0> stage-3 q-pool =="
$root/tools/pmc.sh r5pmc_march pool_march -- pool 8 4 8 28 28 1 20
echo "== PMC pool_tiled_kernel<bf16,2> stage-3 k/v pair =="
$root/tools/pmc.sh r5pmc_tiled pool_tiled -- poolkv 8 4 8 28 28 20
fi
