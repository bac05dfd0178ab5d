#!/bin/bash
# GPU box: fused skip path (csrc/skip_pool.hip) -- tests, op-level fused vs unfused, then train / forward with MVIT_SKIP_FUSE=0 / 1 in one session
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -q -k "proj_maxpool or maxpool" 2>&1 | tail -3
python3 tools/opbench.py projpool 8 8 112 112 96 192 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 56 56 192 384 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 28 28 384 768 20 2>&1 | grep projpool
for f in 0 1 0 1; do
  echo "MVIT_SKIP_FUSE=$f train: $(MVIT_SKIP_FUSE=$f python bench.py --no-cpu-baseline --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  echo "MVIT_SKIP_FUSE=$f fwd bf16: $(MVIT_SKIP_FUSE=$f python bench.py --mode fwd --precision bf16 --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
MVIT_SKIP_FUSE=1 bash tools/prof_noside.sh r3_skip1 > gpurun_out/r3_skip1.txt 2>&1
python3 tools/kstats.py gpurun_out/r3_skip1 7 80 | grep -iE "maxpool|linear_mfma|wgrad_mfma|proj_max|sum of" | cut -c1-200
