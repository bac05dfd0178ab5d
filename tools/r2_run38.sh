root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for cfg in "" "MVIT_POOL_FWD_SIDE=0" "MVIT_POOL_BWD_SIDE=0" "MVIT_ATT_BWD_SIDE=0" "MVIT_POOL_FWD_SIDE=0 MVIT_POOL_BWD_SIDE=0" "MVIT_NO_SIDE_STREAM=1"; do
  echo "[$cfg]"; env $cfg python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done > gpurun_out/r2_t38_streams.txt 2>&1
