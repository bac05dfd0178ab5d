cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  MVIT_ATT_DQ_W64=$w rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dqprof$w -- python3 $GRAFT_REPO_ROOT/tools/opbench.py attnbwd 8 1 100352 1568 10 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/dqprof$w 11 8 | grep -i "attn_bwd"
  MVIT_ATT_DQ_W64=$w rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dqprofb$w -- python3 $GRAFT_REPO_ROOT/tools/opbench.py attnbwd 8 4 6272 1568 10 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/dqprofb$w 11 8 | grep -i "attn_bwd"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/dqprof$w $GRAFT_REPO_ROOT/gpurun_out/dqprofb$w
done
