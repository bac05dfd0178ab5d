#!/bin/bash
# round 5 (GPU box, repo root): dK/dV pass with hand-placed LDS reads three fragments ahead (tree) vs the compiler's schedule (variant bwdold):
# per-kernel durations (rocprofv3), then the train step, interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}
V=$root/aicity_action_amd/lib/variants
for rep in 1 2; do
for n in tree bwdold; do
  lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ $n != tree ] && lib=$V/libmvit_hip_$n.so
  for shape in "8 4 6272 1568" "8 1 100352 1568" "8 2 25088 1568" "8 4 6272 6272" "8 8 1568 1568"; do
    echo "== $n attnbwd $shape: $(MVIT_HIP_LIB=$lib bash $root/tools/kprof_op.sh attnbwd $shape 20 | grep 'attn_bwd_dkv_kernel' | cut -c60-150)"
  done
done
done
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  echo "train step, tree  : $(python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "train step, bwdold: $(MVIT_HIP_LIB=$V/libmvit_hip_bwdold.so python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
done
