#!/bin/bash
# one GPU call: correctness of the ping-pong GEMM, A/B on the model's shapes, ablations + stamps
mkdir -p gpurun_out
(MVIT_GEMM_PP=1 timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear" 2>&1 | tail -5) > gpurun_out/r3_t2.txt
cat gpurun_out/r3_t2.txt
timeout 900 bash tools/r3_gemm_ab.sh gpurun_out/r3_gemm_ab2.txt
bash -c 'SHAPES=("50176 384 1536 b" "50176 1152 384 b"); source tools/ab_pp_abl.sh' > gpurun_out/r3_pp_abl2.txt 2>&1
for shp in "50176 384 1536 b" "50176 1152 384 b"; do
MVIT_GEMM_PP=1 PP_STAMPS=1 MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/pp_abl_8.so python3 tools/opbench.py gemm $shp 20 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_pp_abl2.txt
done
cat gpurun_out/r3_pp_abl2.txt
