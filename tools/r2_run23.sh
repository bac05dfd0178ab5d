root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for shp in "50176 384 1536 b" "32768 384 1536 b" "65536 384 1536 b" "50176 384 384 b" "50176 1152 384 b" "50176 1536 384 bg"; do
  echo "pers:"; python tools/opbench.py gemm $shp 30
  echo "big :"; MVIT_GEMM_NO_PERS=1 python tools/opbench.py gemm $shp 30
done
done > gpurun_out/r2_t23_pers_vs_big.txt 2>&1
