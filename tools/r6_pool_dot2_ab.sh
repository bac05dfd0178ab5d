#!/bin/bash
# GPU box: fp16 build, inference forward pooling convs on v_dot2_f32_f16 (product) vs the FMA + conversion form (-DMARCH_NO_DOT2 variant), alone and in the forward; interleaved
out=${1:-gpurun_out/r6_pool_dot2_ab.txt}
: > $out
V=aicity_action_amd/lib/variants
F16=aicity_action_amd/lib/libmvit_hip_f16.so
for rep in 1 2; do
for shp in "8 4 8 28 28 1" "3 4 8 28 28 1" "8 1 8 112 112 1" "8 2 8 56 56 1" "8 8 8 14 14 1" "8 4 8 28 28 2" "3 4 8 28 28 2" "8 2 8 56 56 2" "8 1 8 112 112 2"; do
  echo "fp16 dot2    $(MVIT_HIP_LIB=$F16 python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  echo "fp16 fma+cvt $(MVIT_HIP_LIB=$V/libmvit_hip_f16_nodot2.so python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
  [ $rep = 1 ] && echo "bf16 fma     $(python tools/opbench.py pool $shp 50 2>/dev/null | tail -1)" >> $out
done
done
for v in dot2 nodot2 dot2 nodot2 dot2 nodot2; do
  lib=$F16; [ $v = nodot2 ] && lib=$V/libmvit_hip_f16_nodot2.so
  echo "$v fwd fp16: $(MVIT_HIP_LIB_F16=$lib python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
cat $out
