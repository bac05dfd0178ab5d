root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_bwd_ops.py tests/test_hip_train.py -q -m gpu -x -k "wgrad or train" 2>&1 | tail -3 > gpurun_out/r2_t33_tests.log
for i in 1 2; do
for shp in "802816 384 96" "802816 288 96" "802816 576 96" "802816 96 96" "200704 192 96"; do
  echo "khalf:"; python tools/opbench.py wgrad $shp 20
  echo "full :"; MVIT_WGRAD_NO_KHALF=1 python tools/opbench.py wgrad $shp 20
done; done > gpurun_out/r2_t33_khalf.txt 2>&1
