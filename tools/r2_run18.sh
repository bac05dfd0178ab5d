root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for shp in "802816 576 96 b" "802816 384 96 b" "802816 384 96 bg" "802816 192 96 b"; do
  echo "-- big/pers path"; python tools/opbench.py gemm $shp 20
  echo "-- dma path"; MVIT_GEMM_NO_BIG=1 python tools/opbench.py gemm $shp 20
done
done > gpurun_out/r2_t18_gemm_k96.txt 2>&1
