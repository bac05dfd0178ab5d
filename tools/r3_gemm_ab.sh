#!/bin/bash
# A/B of the ping-pong GEMM (MVIT_GEMM_PP=1) against the 128x192 kernels (=0) on the model's shapes, one process per line.
out=${1:-gpurun_out/r3_gemm_ab.txt}
: > $out
for shp in "50176 1152 384 b" "50176 384 384 br" "50176 1536 384 bg" "50176 1536 384 b" "50176 384 1536 br" "50176 384 1536 b" \
           "200704 576 192 b" "200704 768 192 bg" "200704 192 768 br" \
           "12544 2304 768 b" "12544 3072 768 bg" "12544 768 3072 br" "12544 768 3072 b"; do
  for e in 0 1; do
    echo "pp=$e $(MVIT_GEMM_PP=$e python3 tools/opbench.py gemm $shp 30 2>&1 | tail -1)" >> $out
  done
done
for shp in "50176 1536 384 pre" "50176 1536 384 der" "50176 1536 384 dgpre" "50176 1536 384 dgder" "12544 3072 768 der" "12544 3072 768 dgder" "200704 768 192 der"; do
  for e in 0 1; do
    echo "pp=$e $(MVIT_GEMM_PP=$e python3 tools/opbench.py gemmdual $shp 30 2>&1 | tail -1)" >> $out
  done
done
cat $out
