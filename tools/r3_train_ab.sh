#!/bin/bash
# in-model A/B of an env switch over a train step: tools/r3_train_ab.sh VAR  (VAR=1 vs VAR=0), wall clock + kernel sums
v=$1
for e in 1 0 1 0; do
  echo "$v=$e: $(env $v=$e python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
for e in 1 0; do
  export $v=$e
  bash tools/prof_noside.sh r3_ab_$e 2>&1 | grep -E "sum of kernel|wgrad" | cut -c1-150
done
