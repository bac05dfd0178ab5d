#!/bin/bash
# fused block tail on / off inside the model, one session: forward clips/s (fp16 default arithmetic and bf16; 3 and 1 sub-batch streams)
for prec in fp16 bf16; do for s in 3 1; do for f in 1 0 1 0; do
  echo "mlp_fuse=$f $prec streams=$s: $(MVIT_MLP_FUSE=$f python bench.py --mode fwd --precision $prec --streams $s --steps 30 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done; done
