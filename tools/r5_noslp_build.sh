#!/bin/bash
# builds variant libraries of the whole tree with -fno-slp-vectorize on EVERY file (bf16 and fp16 builds):
#   aicity_action_amd/lib/variants/libmvit_hip_noslp.so, libmvit_hip_f16_noslp.so
cd "$(dirname "$0")/../aicity_action_amd/csrc" || exit 1
mkdir -p ../lib/variants /tmp/noslp /tmp/noslp16
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -Wno-inline-asm -fno-slp-vectorize"
for f in *.hip; do
  b=${f%.hip}
  ( /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/noslp/$b.o 2>/dev/null || echo "FAILED $f" ) &
  ( /opt/rocm/bin/hipcc $FLAGS -DMVIT_HALF_IS_FP16 -c $f -o /tmp/noslp16/$b.o 2>/dev/null || echo "FAILED16 $f" ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libmvit_hip_noslp.so /tmp/noslp/*.o && echo built noslp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libmvit_hip_f16_noslp.so /tmp/noslp16/*.o && echo built f16 noslp
