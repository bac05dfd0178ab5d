#!/bin/bash
# round 5 (GPU box, repo root): s_memtime stamps of attn_fwd_w64_kernel (variant library built with W64_EXTRA=-DW_STAMP tools/r5_w64_abl.sh 0),
# persistent grid (default) and one workgroup per item (MVIT_ATT_W64_WGS=0)
lib=aicity_action_amd/lib/variants/libmvit_hip_w64stamp.so
for shape in "8 4 6272 1568" "8 1 100352 1568"; do
  echo "persistent grid"; W_STAMP=1 MVIT_HIP_LIB=$lib python3 tools/opbench.py attn $shape 10 2>/dev/null | tail -3
  echo "one workgroup per item"; MVIT_ATT_W64_WGS=0 W_STAMP=1 MVIT_HIP_LIB=$lib python3 tools/opbench.py attn $shape 10 2>/dev/null | tail -3
done
