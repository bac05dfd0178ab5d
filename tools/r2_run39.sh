root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_train.py tests/test_hip_bwd_ops.py tests/test_hip_engine.py tests/test_hip_ddp.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r2_t39_tests.log
for i in 1 2; do
for cfg in "--mode train" "--mode fwd" "--mode fwd --streams 1"; do
for e in "" "MVIT_NO_SIDE_STREAM=1"; do
  echo "[$cfg | $e]"; env $e python bench.py $cfg --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done; done > gpurun_out/r2_t39_streams.txt 2>&1
