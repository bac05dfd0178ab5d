#!/bin/bash
# GPU box: MFMA-shape clock probe + PMC issue counters of the three attention kernels at the stage-3 shape (B=8, 4 heads, Lq 6273, Lk 1569)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 tools/probes/mfma_shape.py > gpurun_out/r3_mfma_shape_probe.txt 2>&1
python3 tools/opbench.py attn 8 4 6273 1569 50 > gpurun_out/r3_attn_issue.txt 2>&1
python3 tools/opbench.py attnbwd 8 4 6273 1569 30 >> gpurun_out/r3_attn_issue.txt 2>&1
for k in attn_fwd_pipe attn_bwd_dq attn_bwd_dkv; do
  op=attnbwd; [ $k = attn_fwd_pipe ] && op=attn
  echo "== $k ($op 8 4 6273 1569), counters per launch" >> gpurun_out/r3_attn_issue.txt
  bash tools/pmc.sh r3_pmc_$k $k -- $op 8 4 6273 1569 10 >> $root/gpurun_out/r3_attn_issue.txt 2>&1
done
