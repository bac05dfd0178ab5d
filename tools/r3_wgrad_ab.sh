#!/bin/bash
# weight-gradient shapes of the model: ping-pong kernel (MVIT_WGRAD_PP default) vs the 128 x 192 kernel (=0), kernel-level (opbench)
out=${1:-gpurun_out/r3_wgrad_ab.txt}
: > $out
for shp in "50176 1152 384" "50176 384 384" "50176 1536 384" "50176 384 1536" "12544 2304 768" "12544 768 768" "12544 3072 768" "12544 768 3072" "200704 768 192" "200704 192 768"; do
  for e in 1 0; do
    echo "pp=$e $(MVIT_WGRAD_PP=$e python3 tools/opbench.py wgrad $shp 30 2>&1 | tail -1)" >> $out
  done
done
cat $out
