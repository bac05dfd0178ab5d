root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for cfg in "" "MVIT_NO_SIDE_STREAM=1" "MVIT_WGRAD_STREAM=0" "MVIT_NO_SIDE_STREAM=1 MVIT_WGRAD_STREAM=0"; do
  echo "[$cfg]"; env $cfg python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done > gpurun_out/r2_t37_streams.txt 2>&1
