root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_bwd_ops.py -q -m gpu -x -k "kv_pair" 2>&1 | tail -5 > gpurun_out/r2_t55_tests.log
python -m pytest tests/test_hip_train.py tests/test_hip_engine.py tests/test_hip_ddp.py tests/test_hip_model.py -q -m gpu 2>&1 | tail -5 >> gpurun_out/r2_t55_tests.log
for i in 1 2 3; do
for e in "MVIT_POOL_KV_BATCH=0" "MVIT_POOL_KV_BATCH=1"; do
  echo "[train | $e]"; env $e python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done > gpurun_out/r2_t55_kvbatch.txt 2>&1
