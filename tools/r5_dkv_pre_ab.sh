#!/bin/bash
# round 5 (GPU box, repo root): dK/dV pass, accumulators started at + lse / + delta on negated K / V fragments (MVIT_ATT_DKV_PRE=1, default) vs
# the multiply-add / subtract form (0): per-kernel durations (rocprofv3), then the train step, interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in 1 0; do
  for shape in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 1568" "8 4 6272 6272"; do
    echo "== MVIT_ATT_DKV_PRE=$v attnbwd $shape"
    MVIT_ATT_DKV_PRE=$v bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep "attn_bwd_dkv_kernel"
  done
done
done
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  for v in 1 0; do
    echo "train step, MVIT_ATT_DKV_PRE=$v: $(MVIT_ATT_DKV_PRE=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  done
done
