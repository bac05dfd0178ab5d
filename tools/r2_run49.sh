root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2 3; do
for e in "" "MVIT_WGRAD_JOIN=end"; do
  echo "[train | $e]"; env $e python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done > gpurun_out/r2_t49_join.txt 2>&1
MVIT_WGRAD_JOIN=end python -m pytest tests/test_hip_train.py -q -m gpu -x 2>&1 | tail -2 >> gpurun_out/r2_t49_join.txt
