root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_hip_bwd_ops.py tests/test_hip_train.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r2_t26_tests.log
for i in 1 2 3; do
for wgs in 0 384 448; do
  echo "WGS=$wgs"; MVIT_WGRAD_WGS=$wgs python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done
done > gpurun_out/r2_t26_wgs_model.txt 2>&1
