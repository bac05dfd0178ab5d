set -x
python -m pytest tests/test_hip_bwd_ops.py -q -x -k attention 2>&1 | tail -5 > gpurun_out/r2_t3_attn_tests.log
for cfg in "8 1 100352 1568" "16 1 25088 6272" "16 1 25088 1568" "32 1 6272 6272" "32 1 6272 1568" "64 1 1568 6272" "64 1 1568 1568"; do
  set -- $cfg
  MVIT_NO_SIDE_STREAM=1 python tools/opbench.py attnbwd $1 $2 $3 $4 >> gpurun_out/r2_t3_attnbwd_noside.txt 2>&1
  python tools/opbench.py attnbwd $1 $2 $3 $4 >> gpurun_out/r2_t3_attnbwd_side.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp MVIT_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_t3_prof -- python3 $GRAFT_REPO_ROOT/tools/opbench.py attnbwd 32 1 6272 1568 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/kstats.py gpurun_out/r2_t3_prof 11 > gpurun_out/r2_t3_prof.txt; rm -rf gpurun_out/r2_t3_prof
