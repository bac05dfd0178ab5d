#!/bin/bash
# round 5 (GPU box, repo root): per-kernel durations of the eval forward in fp16 (default) and bf16, one stream, same box
root=${GRAFT_REPO_ROOT:-$(pwd)}
for p in fp16 bf16; do
  tools/prof_noside.sh r5_prec_$p --mode fwd --precision $p > /dev/null 2>&1
  python3 tools/kstats.py gpurun_out/r5_prec_$p 7 14 > gpurun_out/r5_fwd_${p}_noside_per_step.txt
  rm -rf gpurun_out/r5_prec_$p
done
