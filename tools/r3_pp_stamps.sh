#!/bin/bash
# cycle stamps of the ping-pong GEMM (lib from tools/build_pp_abl.sh 8)
export MVIT_GEMM_PP=1 PP_STAMPS=1
export MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/pp_abl_8.so
for shp in "50176 384 1536 b" "50176 1152 384 b" "50176 384 1536 br" "50176 1536 384 bg"; do
  python3 tools/opbench.py gemm $shp 20 2>&1 | grep -v amdgpu.ids
done
