#!/bin/bash
# GPU box: linear_big_kernel with the epilogue operands requested under the main loop (product) vs. after it (BIG_ABL=4 = round-5 form),
# + ablations (1 = no stores, 2 = no operand loads, 5 = neither), alone and in the train step; interleaved
out=${1:-gpurun_out/r6_gemm_big_prefetch_ab.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemmdual 200704 768 192 dgder 30" "gemmdual 12544 3072 768 dgder 30" "gemm 50176 384 1536 r 30" "gemm 200704 192 768 r 30" "gemm 12544 768 3072 r 30"; do
  echo "prefetch   $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  echo "r5 form    $(MVIT_HIP_LIB=$V/libmvit_hip_big4.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  if [ $rep = 1 ]; then
  echo "  no stores (r5 form + abl 1) $(MVIT_HIP_LIB=$V/libmvit_hip_big5.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  echo "  no operand loads (abl 2)    $(MVIT_HIP_LIB=$V/libmvit_hip_big2.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  echo "  prefetch, no stores (abl 1) $(MVIT_HIP_LIB=$V/libmvit_hip_big1.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  fi
done
done
for v in prod big4 prod big4 prod big4; do
  lib=aicity_action_amd/lib/libmvit_hip.so; [ $v = big4 ] && lib=$V/libmvit_hip_big4.so
  echo "$v train bf16: $(MVIT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out
done
cat $out
