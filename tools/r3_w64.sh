#!/bin/bash
# GPU box: the 64-query-per-wave attention forward (MVIT_ATT_W64=1) -- parity tests, then kernel-alone timing against the default
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
MVIT_ATT_W64=1 timeout 600 python -m pytest tests/test_hip_ops.py -q -k "attention" -x 2>&1 | tail -3
for w in 0 1; do
  echo "MVIT_ATT_W64=$w: $(MVIT_ATT_W64=$w timeout 120 python3 tools/opbench.py attn 8 4 6272 1568 50 2>&1 | tail -1)"
  echo "MVIT_ATT_W64=$w: $(MVIT_ATT_W64=$w timeout 120 python3 tools/opbench.py attn 8 1 100352 1568 10 2>&1 | tail -1)"
  echo "MVIT_ATT_W64=$w: $(MVIT_ATT_W64=$w timeout 120 python3 tools/opbench.py attn 8 8 1568 1568 50 2>&1 | tail -1)"
done
if [ -f aicity_action_amd/lib/w64_stamp.so ]; then
  MVIT_HIP_LIB=$PWD/aicity_action_amd/lib/w64_stamp.so MVIT_ATT_W64=1 W_STAMP=1 python3 tools/opbench.py attn 8 1 100352 1568 5 2>&1 | tail -1
fi
