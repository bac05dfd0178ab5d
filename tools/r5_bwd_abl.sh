#!/bin/bash
# builds variant libraries with attention_bwd.hip compiled under -DBWD_ABL=<n> (timing ablations of the dQ pass; see attention_bwd.hip)
cd "$(dirname "$0")/../aicity_action_amd/csrc" || exit 1
mkdir -p ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -fno-slp-vectorize"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DBWD_ABL=$n -c attention_bwd.hip -o ../lib/variants/attention_bwd_$n.o 2>/dev/null || exit 1
  objs=$(ls ../lib/obj/*.o | grep -v attention_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libmvit_hip_bwdabl$n.so $objs ../lib/variants/attention_bwd_$n.o || exit 1
  rm -f ../lib/variants/attention_bwd_$n.o
  echo built libmvit_hip_bwdabl$n.so ) &
done
wait
