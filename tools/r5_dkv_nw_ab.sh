#!/bin/bash
# round 5 (GPU box, repo root): dK/dV pass with eight waves (256 keys) per workgroup (default for Lk > 128) vs four (MVIT_ATT_DKV_WAVES=4):
# per-kernel durations (rocprofv3), then the train step, interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for w in 8 4; do
  for shape in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 1568" "8 4 6272 6272" "8 2 25088 6272" "8 8 1568 6272" "8 8 1568 1568"; do
    echo "== waves $w attnbwd $shape: $(MVIT_ATT_DKV_WAVES=$w bash $root/tools/kprof_op.sh attnbwd $shape 12 | grep 'attn_bwd_dkv_kernel' | cut -c60-150)"
  done
done
done
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  for w in 8 4; do
    echo "train step, $w waves: $(MVIT_ATT_DKV_WAVES=$w python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  done
done
