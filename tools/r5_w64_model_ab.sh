#!/bin/bash
# round 5 (GPU box, repo root): the 64-query attention forward inside the model -- tree (persistent grid) vs MVIT_ATT_W64_WGS=0 (one workgroup
# per item) vs the previous kernel (variant library w64old; bf16 library only, so the forward is timed in bf16), interleaved
V=aicity_action_amd/lib/variants
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  echo "fwd bf16, tree        : $(python bench.py --mode fwd --precision bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd bf16, WGS=0       : $(MVIT_ATT_W64_WGS=0 python bench.py --mode fwd --precision bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd bf16, old kernel  : $(MVIT_HIP_LIB=$V/libmvit_hip_w64old.so python bench.py --mode fwd --precision bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "train, tree           : $(python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "train, WGS=0          : $(MVIT_ATT_W64_WGS=0 python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "train, old kernel     : $(MVIT_HIP_LIB=$V/libmvit_hip_w64old.so python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
done
echo "fwd fp16, tree        : $(python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
echo "fwd fp16, WGS=0       : $(MVIT_ATT_W64_WGS=0 python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
