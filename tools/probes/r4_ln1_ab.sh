#!/bin/bash
# next block's norm1 emitted by the fused block tail, on / off, forward clips/s in one session
for prec in fp16; do for f in 1 0 1 0; do
  echo "ln1_fuse=$f $prec streams=3: $(MVIT_LN1_FUSE=$f python bench.py --mode fwd --precision $prec --streams 3 --steps 30 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
