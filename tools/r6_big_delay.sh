#!/bin/bash
# GPU box: one-time start delay for the second resident workgroup of every CU in linear_big_kernel (BIG_ABL=64, BIG_DELAY x 8128 cycles) vs the product; interleaved
out=${1:-gpurun_out/r6_gemm_big_delay.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemmdual 200704 768 192 dgder 30" "gemm 50176 384 1536 r 30"; do
  echo "product  $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  for d in 1 2 4; do echo "delay $d  $(MVIT_HIP_LIB=$V/libmvit_hip_bigd$d.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out; done
done
done
cat $out
