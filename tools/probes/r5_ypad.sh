#!/bin/bash
# does the base-address offset between the two 16-bit outputs of the fc1 GEMM (GELU | GELU') / the aux input of the fc2 data gradient matter?
for rep in 1 2; do for pad in 0 256 1024 4096 65536 1048576; do
  echo "pad $pad: $(OPB_Y_PAD=$pad python3 tools/opbench.py gemmdual 50176 1536 384 der 30 2>/dev/null | tail -1) | $(OPB_Y_PAD=$pad python3 tools/opbench.py gemmdual 50176 1536 384 dgder 30 2>/dev/null | tail -1)"
done; done
