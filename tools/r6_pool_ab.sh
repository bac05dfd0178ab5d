#!/bin/bash
# GPU box: A/B of the full-lane march kernel against the 48-lane one (MVIT_POOL_MARCH2=0), alone, interleaved; "occ" = the variant
# library whose full-lane kernel is capped at 128 registers (two workgroups per CU)
out=${1:-gpurun_out/r6_pool_march2_ab.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for shp in "8 4 8 28 28 1" "8 1 8 112 112 1" "8 2 8 56 56 1" "8 8 8 14 14 1" "3 4 8 28 28 1" "3 1 8 112 112 1"; do
  echo "old48   bf16 $(MVIT_POOL_MARCH2=0 python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "full    bf16 $(python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "fullocc bf16 $(MVIT_HIP_LIB=$V/libmvit_hip_march2occ.so python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "old48   fp16 $(MVIT_POOL_MARCH2=0 MVIT_HIP_LIB=aicity_action_amd/lib/libmvit_hip_f16.so python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "full    fp16 $(MVIT_HIP_LIB=aicity_action_amd/lib/libmvit_hip_f16.so python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "fullocc fp16 $(MVIT_HIP_LIB=$V/libmvit_hip_f16_march2occ.so python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
done
done
cat $out
