set -x
python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -q -x -k pool 2>&1 | tail -8 > gpurun_out/r2_t4_pool_tests.log
for m in 0 1; do
for cfg in "8 1 8 112 112 1" "8 2 8 112 112 2" "8 2 8 56 56 1" "8 4 8 56 56 2" "8 4 8 28 28 1" "8 4 8 28 28 2" "8 8 8 28 28 2" "8 8 8 14 14 1"; do
  set -- $cfg
  MVIT_POOL_MARCH=$m python tools/opbench.py pool $1 $2 $3 $4 $5 $6 2>/dev/null >> gpurun_out/r2_t4_pool_m$m.txt
done
done
