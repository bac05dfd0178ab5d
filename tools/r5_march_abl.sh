#!/bin/bash
# builds variant libraries of the tree with pool_march.hip compiled under -DMARCH_ABL=<n> (timing ablations; see pool_march.hip)
#   tools/r5_march_abl.sh 1 2 4 8 16 ...   ->  aicity_action_amd/lib/variants/libmvit_hip_march<n>.so   (extra flags: MARCH_EXTRA="-D...")
cd "$(dirname "$0")/../aicity_action_amd/csrc" || exit 1
mkdir -p ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -fno-slp-vectorize"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DMARCH_ABL=$n $MARCH_EXTRA -c pool_march.hip -o ../lib/variants/pool_march_$n.o || exit 1
  objs=$(ls ../lib/obj/*.o | grep -v pool_march.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libmvit_hip_march$n.so $objs ../lib/variants/pool_march_$n.o || exit 1
  rm -f ../lib/variants/pool_march_$n.o
  echo built libmvit_hip_march$n.so
done
