bash tools/pmc.sh r2_pmc_g1 linear -- gemm 50176 384 1536 br > gpurun_out/r2_pmc_gemm_fc2.txt 2>&1
bash tools/pmc.sh r2_pmc_g2 linear -- gemm 50176 1536 384 bg > gpurun_out/r2_pmc_gemm_fc1.txt 2>&1
bash tools/pmc.sh r2_pmc_g3 linear -- gemm 50176 1152 384 b > gpurun_out/r2_pmc_gemm_qkv.txt 2>&1
rm -rf gpurun_out/r2_pmc_g1 gpurun_out/r2_pmc_g2 gpurun_out/r2_pmc_g3
for a in "50176 384 1536 br" "50176 1536 384 bg" "50176 1152 384 b" "50176 384 384 br" "200704 576 192 b" "200704 768 192 bg" "200704 192 768 br" "802816 288 96 b" "802816 384 96 bg" "802816 96 384 br" "12544 2304 768 b" "12544 3072 768 bg" "12544 768 3072 br"; do
python tools/opbench.py gemm $a 2>/dev/null >> gpurun_out/r2_gemm_shapes.txt
done
