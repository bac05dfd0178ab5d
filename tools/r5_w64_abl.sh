#!/bin/bash
# builds variant libraries of the tree with attention_w64.hip compiled under -DW_ABL=<n> (timing ablations; see attention_w64.hip)
#   tools/r5_w64_abl.sh 1 2 4 8 ...   ->  aicity_action_amd/lib/variants/libmvit_hip_w64abl<n>.so   (extra flags: W64_EXTRA="-D...")
cd "$(dirname "$0")/../aicity_action_amd/csrc" || exit 1
mkdir -p ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -Wno-inline-asm -fno-slp-vectorize -Wno-uninitialized"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DW_ABL=$n $W64_EXTRA -c attention_w64.hip -o ../lib/variants/attention_w64_$n.o 2>/dev/null || exit 1
  objs=$(ls ../lib/obj/*.o | grep -v attention_w64.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libmvit_hip_w64abl$n.so $objs ../lib/variants/attention_w64_$n.o || exit 1
  rm -f ../lib/variants/attention_w64_$n.o
  echo built libmvit_hip_w64abl$n.so ) &
done
wait
