root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for i in 1 2; do
for e in "" "MVIT_POOL_MARCH=0" "MVIT_ATT_PIPE=0" "MVIT_GELU_DSAVE=0" "MVIT_GEMM_BM256=0" "MVIT_GEMM_NO_PERS=1" "MVIT_ATT_SLOT=1"; do
  echo "[train | $e]"; env $e python bench.py --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done
for e in "" "MVIT_POOL_MARCH=0" "MVIT_ATT_PIPE=0" "MVIT_GEMM_NO_PERS=1" "MVIT_ATT_SLOT=1"; do
  echo "[fwd | $e]"; env $e python bench.py --mode fwd --no-cpu-baseline --no-forward-record --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_event_median'])"
done
done > gpurun_out/r2_t40_toggles.txt 2>&1
