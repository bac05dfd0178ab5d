#!/bin/bash
# round 5 probe (GPU box, repo root): s_setprio 1 for the workgroups whose linear id has bit MVIT_ATT_DKV_PRIO_BIT set (dK/dV pass)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for b in -1 0 1 8 9; do
  for shape in "8 4 6272 1568" "8 4 6272 6272"; do
    echo "== prio bit $b attnbwd $shape: $(MVIT_ATT_DKV_PRIO_BIT=$b bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep 'attn_bwd_dkv_kernel' | cut -c60-150)"
  done
done
done
