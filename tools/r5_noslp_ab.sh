#!/bin/bash
# round 5 (GPU box, repo root): whole tree without SLP vectorisation (no v_pk_*_f32 in the epilogues / row kernels) vs the tree, interleaved:
# the epilogue-heavy training GEMMs alone, then train step and forward (bf16 and fp16)
V=aicity_action_amd/lib/variants
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2; do
for n in tree noslp; do
  lib=aicity_action_amd/lib/libmvit_hip.so; [ $n = noslp ] && lib=$V/libmvit_hip_noslp.so
  for mode in der dgder; do echo "$n: $(MVIT_HIP_LIB=$lib python3 tools/opbench.py gemmdual 50176 1536 384 $mode 30 2>/dev/null | tail -1)"; done
done
done
for rep in 1 2 3; do
  echo "train, tree : $(python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "train, noslp: $(MVIT_HIP_LIB=$V/libmvit_hip_noslp.so MVIT_HIP_LIB_F16=$V/libmvit_hip_f16_noslp.so python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "fwd fp16, tree : $(python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd fp16, noslp: $(MVIT_HIP_LIB=$V/libmvit_hip_noslp.so MVIT_HIP_LIB_F16=$V/libmvit_hip_f16_noslp.so python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd bf16, tree : $(python bench.py --mode fwd --precision bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  echo "fwd bf16, noslp: $(MVIT_HIP_LIB=$V/libmvit_hip_noslp.so MVIT_HIP_LIB_F16=$V/libmvit_hip_f16_noslp.so python bench.py --mode fwd --precision bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
done
