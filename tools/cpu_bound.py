#!/usr/bin/env python3
"""Is a step limited by the host (Python + ctypes launch cost) or by the GPU?  Enqueue time vs completed time per step.

    python tools/cpu_bound.py [fwd|train] [streams]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", "bf16", "HIP.STREAMS", streams])
model = build_model(cfg)
load_synth_weights(model, 0)
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
if mode == "train":
    from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy
    model.train()
    opt = construct_optimizer(model, cfg)
    labels = torch.zeros(8, cfg.MODEL.NUM_CLASSES, device="cuda")
    labels[torch.arange(8), torch.arange(8) % cfg.MODEL.NUM_CLASSES] = 1.0

    def step():
        opt.set_lr(1e-4)
        loss = soft_target_cross_entropy(model([clip]), labels)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
else:
    model.eval()

    def step():
        with torch.no_grad():
            model([clip])
for _ in range(5):
    step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# pure host cost: ONE step enqueued into an empty queue (the back-to-back figure above includes waiting for queue slots once the
# launch queue is full, i.e. it converges to the GPU time)
single = []
for _ in range(8):
    torch.cuda.synchronize()
    a = time.perf_counter()
    step()
    single.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print("%s streams=%d: host enqueue of one step into an empty queue: median %.2f ms (min %.2f)" % (mode, streams, sorted(single)[len(single) // 2] * 1e3, min(single) * 1e3))
if mode == "train":
    print("optimizer chunk-table builds so far:", getattr(opt, "table_builds", 0))
print("%s streams=%d: host enqueue %.2f ms/step, completed %.2f ms/step (GPU-bound if enqueue << completed)" % (
    mode, streams, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
