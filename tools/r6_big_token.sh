#!/bin/bash
# GPU box: per-CU token around the main loop of linear_big_kernel (BIG_ABL=32, results valid) against the product, alone and in the train step
out=${1:-gpurun_out/r6_gemm_big_token.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemmdual 200704 768 192 dgder 30" "gemm 50176 384 1536 r 30" "gemm 50176 1152 384 b 30"; do
  echo "product  $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  echo "token    $(timeout 120 env MVIT_HIP_LIB=$V/libmvit_hip_big32.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
done
done
for v in prod tok prod tok; do
  lib=aicity_action_amd/lib/libmvit_hip.so; [ $v = tok ] && lib=$V/libmvit_hip_big32.so
  echo "$v train bf16: $(timeout 300 env MVIT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
cat $out
