#!/bin/bash
# round 5 (GPU box, repo root): start offsets for the first round of dQ workgroups (MVIT_ATT_DQ_STAGGER units of 1024 cycles per step, 16 steps)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for u in 0 1 2 4; do
  for shape in "8 1 100352 1568" "8 2 25088 1568" "8 4 6272 1568"; do
    echo "== stagger $u attnbwd $shape: $(MVIT_ATT_DQ_STAGGER=$u MVIT_ATT_DQ_STAGGER_ROUNDS=2 bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep 'attn_bwd_dq' | cut -c60-140)"
  done
done
done
