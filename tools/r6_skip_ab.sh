#!/bin/bash
# GPU box: fused widening skip path, forward kernel with two weight slabs of lead (product) vs one (round-5 form, variant library); alone + in the model; interleaved
out=${1:-gpurun_out/r6_skip_slab_lead_ab.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for shp in "8 8 28 28 384 768" "8 8 56 56 192 384" "8 8 112 112 96 192" "3 8 28 28 384 768"; do
  echo "lead 2  $(python tools/opbench.py projpool $shp 30 2>/dev/null | grep "fwd fused" | tr '\n' ' ')" >> $out
  echo "lead 1  $(MVIT_HIP_LIB=$V/libmvit_hip_spold.so python tools/opbench.py projpool $shp 30 2>/dev/null | grep "fwd fused" | tr '\n' ' ')" >> $out
done
done
for v in new old new old new old; do
  l16=aicity_action_amd/lib/libmvit_hip_f16.so; [ $v = old ] && l16=$V/libmvit_hip_f16_spold.so
  echo "$v fwd fp16: $(MVIT_HIP_LIB_F16=$l16 python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
for v in new old new old; do
  lib=aicity_action_amd/lib/libmvit_hip.so; [ $v = old ] && lib=$V/libmvit_hip_spold.so
  echo "$v train bf16: $(MVIT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
cat $out
