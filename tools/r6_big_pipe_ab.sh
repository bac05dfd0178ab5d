#!/bin/bash
# GPU box: linear_big_kernel with two slabs in flight (product, BIG_PIPE=2) vs one (round-5 loop, -DBIG_PIPE=1 variant); alone, stamps, in the model; interleaved
out=${1:-gpurun_out/r6_gemm_big_pipe_ab.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemmdual 200704 768 192 dgder 30" "gemmdual 12544 3072 768 dgder 30" "gemm 50176 384 1536 r 30" "gemm 200704 192 768 r 30" "gemm 12544 768 3072 r 30" "gemm 802816 96 384 r 20"; do
  echo "pipe 2  $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  echo "pipe 1  $(MVIT_HIP_LIB=$V/libmvit_hip_pipe1.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
done
done
echo "--- stamps, two slabs in flight" >> $out
MVIT_HIP_LIB=$V/libmvit_hip_bigstamp.so python tools/r6_big_stamps.py dgder 2>&1 | grep -v amdgpu.ids >> $out
MVIT_HIP_LIB=$V/libmvit_hip_bigstamp.so python tools/r6_big_stamps.py r 2>&1 | grep -v amdgpu.ids >> $out
for v in new old new old new old; do
  lib=aicity_action_amd/lib/libmvit_hip.so; [ $v = old ] && lib=$V/libmvit_hip_pipe1.so
  echo "$v train bf16: $(MVIT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
for v in new old new old; do
  l16=aicity_action_amd/lib/libmvit_hip_f16.so; [ $v = old ] && l16=$V/libmvit_hip_f16_pipe1.so
  echo "$v fwd fp16: $(MVIT_HIP_LIB_F16=$l16 python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
cat $out
