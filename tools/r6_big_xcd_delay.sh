#!/bin/bash
# GPU box: XCD-staggered start of linear_big_kernel's first round (BIG_ABL=128) vs the product; interleaved
out=${1:-gpurun_out/r6_gemm_big_xcd_delay.txt}
: > $out
V=aicity_action_amd/lib/variants
for rep in 1 2; do
for args in "gemmdual 50176 1536 384 dgder 30" "gemm 50176 384 1536 r 30"; do
  echo "product    $(python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out
  for d in 1 2 4; do echo "xcd delay $d $(MVIT_HIP_LIB=$V/libmvit_hip_bigx$d.so python tools/opbench.py $args 2>/dev/null | tail -1)" >> $out; done
done
done
cat $out
