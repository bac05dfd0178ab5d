#!/bin/bash
# GPU box: the 64-query dQ pass (attention_bwd_w64.hip; MVIT_ATT_DQ_W64=0 selects the 32-query kernel) -- parity, kernel-alone timing
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 600 python -m pytest tests/test_hip_bwd_ops.py -q -k "attention" -x 2>&1 | tail -4
for w in 0 1; do
  echo "MVIT_ATT_DQ_W64=$w: $(MVIT_ATT_DQ_W64=$w timeout 120 python3 tools/opbench.py attnbwd 8 4 6272 1568 30 2>&1 | tail -1)"
  echo "MVIT_ATT_DQ_W64=$w: $(MVIT_ATT_DQ_W64=$w timeout 120 python3 tools/opbench.py attnbwd 8 1 100352 1568 10 2>&1 | tail -1)"
  echo "MVIT_ATT_DQ_W64=$w: $(MVIT_ATT_DQ_W64=$w timeout 120 python3 tools/opbench.py attnbwd 8 8 1568 1568 30 2>&1 | tail -1)"
done
