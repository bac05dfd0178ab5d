#!/bin/bash
# one GPU call: correctness of the ping-pong GEMM (forced on for every shape it takes, then the default routing), A/B on the model's shapes, stamps
mkdir -p gpurun_out
(MVIT_GEMM_PP=1 timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear or gelu" 2>&1 | tail -5) > gpurun_out/r3_t4.txt
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -x -q -m gpu 2>&1 | tail -5) >> gpurun_out/r3_t4.txt
cat gpurun_out/r3_t4.txt
timeout 900 bash tools/r3_gemm_ab.sh gpurun_out/r3_gemm_ab4.txt
export MVIT_GEMM_PP=1 PP_STAMPS=1 MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/pp_abl_8.so
for shp in "50176 384 1536 b" "50176 1152 384 b" "50176 384 1536 br" "50176 1536 384 bg"; do
  python3 tools/opbench.py gemm $shp 20 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_pp_stamps5.txt 2>&1
cat gpurun_out/r3_pp_stamps5.txt
