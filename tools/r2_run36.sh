root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2 3; do
for m in 1 0 auto; do
  if [ $m = auto ]; then unset MVIT_ATT_BWD_SIDE; else export MVIT_ATT_BWD_SIDE=$m; fi
  echo "side=$m"; python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done; done > gpurun_out/r2_t36_attn_side_model.txt 2>&1
unset MVIT_ATT_BWD_SIDE
for shp in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 6272" "8 4 6272 6272" "8 8 1568 6272" "8 8 1568 1568" "8 2 25088 1568"; do
for m in 1 0; do echo "side=$m"; MVIT_ATT_BWD_SIDE=$m python tools/opbench.py attnbwd $shp 20; done; done > gpurun_out/r2_t36_attn_side_op.txt 2>&1
