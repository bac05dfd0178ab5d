root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
T="tests/test_hip_train.py tests/test_hip_bwd_ops.py tests/test_hip_engine.py tests/test_hip_ddp.py"
(echo "== MVIT_NO_SIDE_STREAM=1"; MVIT_NO_SIDE_STREAM=1 MVIT_WGRAD_STREAM=0 python -m pytest $T -q -m gpu 2>&1 | tail -3
 echo "== MVIT_REDUCE_QUEUE=0 MVIT_LN_EMIT16=0"; MVIT_REDUCE_QUEUE=0 MVIT_LN_EMIT16=0 python -m pytest $T -q -m gpu 2>&1 | tail -3
 echo "== MVIT_GEMM_BM256=1"; MVIT_GEMM_BM256=1 python -m pytest $T tests/test_hip_model.py tests/test_hip_ops.py -q -m gpu 2>&1 | tail -3
 echo "== MVIT_POOL_MARCH=0"; MVIT_POOL_MARCH=0 python -m pytest $T tests/test_hip_model.py -q -m gpu 2>&1 | tail -3
 echo "== MVIT_GELU_DSAVE=0 MVIT_GEMM_NO_PERS=1"; MVIT_GELU_DSAVE=0 MVIT_GEMM_NO_PERS=1 python -m pytest $T tests/test_hip_model.py -q -m gpu 2>&1 | tail -3
) > gpurun_out/r2_t28_toggles.log 2>&1
