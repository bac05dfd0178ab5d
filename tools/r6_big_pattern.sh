#!/bin/bash
# GPU box: is the epilogue of linear_big_kernel bound by the ACCESS PATTERN of its stores / operand loads (32 rows x 32 B per instruction)?
# ablation builds that move the same bytes as contiguous 1-KiB runs (results invalid): 8 = stores, 16 = operand loads, 24 = both; 3 = neither stores nor loads
out=${1:-gpurun_out/r6_gemm_big_pattern.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemmdual 200704 768 192 dgder 30"; do
  echo "product (prefetch)            $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  for n in 8 16 24 1 2 3; do
    echo "  abl $n $(MVIT_HIP_LIB=$V/libmvit_hip_big$n.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  done
done
done
cat $out
