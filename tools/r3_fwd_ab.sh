#!/bin/bash
for pp in 1 0; do for s in 1 2; do
  echo "pp=$pp streams=$s: $(MVIT_GEMM_PP=$pp python bench.py --mode fwd --streams $s --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
for pp in 1 0; do
  echo "train pp=$pp: $(MVIT_GEMM_PP=$pp python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  echo "train pp=$pp no wgrad stream: $(MVIT_GEMM_PP=$pp MVIT_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
