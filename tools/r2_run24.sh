root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
MVIT_GEMM_BM256=1 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "linear" 2>&1 | tail -3 > gpurun_out/r2_t24_tests.log
for i in 1 2; do
for shp in "50176 1152 384 b" "50176 1536 384 bg" "50176 384 384 b" "50176 384 1536 b" "200704 576 192 b" "200704 768 192 bg" "12544 2304 768 b" "12544 3072 768 bg" "12544 768 3072 b" "802816 384 96 bg"; do
  echo "128:"; python tools/opbench.py gemm $shp 30
  echo "256:"; MVIT_GEMM_BM256=1 python tools/opbench.py gemm $shp 30
done
done > gpurun_out/r2_t24_bm256.txt 2>&1
