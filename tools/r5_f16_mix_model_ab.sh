#!/bin/bash
# round 5 (GPU box, repo root): fp16 forward, stride-1 pooling march with explicit conversions (tree) vs folded into v_fma_mix_f32 (variant), interleaved
V=aicity_action_amd/lib/variants
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3 4; do
  echo "fwd fp16, explicit conversions: $(python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd fp16, v_fma_mix_f32 taps  : $(MVIT_HIP_LIB_F16=$V/libmvit_hip_f16_fmamix.so python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
done
