root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_bwd_ops.py -q -m gpu -x -k "attention" 2>&1 | tail -3 > gpurun_out/r2_t14_tests.log
(tools/ab_lib.sh a attnbwd 8 4 6273 1569 30; tools/ab_lib.sh b attnbwd 8 1 100353 1569 10; tools/ab_lib.sh c attnbwd 8 2 25089 6273 10) > gpurun_out/r2_t14_ab.txt 2>&1
