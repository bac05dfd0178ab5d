#!/usr/bin/env python3
"""Which Python lines of a train step launch torch fill / zero kernels (each is a ~2.4 us launch on the step's one stream)?
usage (GPU box): python tools/find_fills.py"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", "bf16"])
model = build_model(cfg, gpu_id=0).train()
load_synth_weights(model, 0)
opt = construct_optimizer(model, cfg)
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
labels = torch.zeros(8, cfg.MODEL.NUM_CLASSES, device="cuda")
labels[torch.arange(8), torch.arange(8) % cfg.MODEL.NUM_CLASSES] = 1.0


def step():
    opt.set_lr(1e-4)
    loss = soft_target_cross_entropy(model([clip]), labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
hits = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("fill", "zero", "ones", "full")):
            fr = [f for f in traceback.extract_stack() if "aicity_action_amd" in f.filename or f.filename.endswith("find_fills.py")]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
            hits[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step()
torch.cuda.synchronize()
for (name, where), n in hits.most_common(40):
    print("%4d  %-28s %s" % (n, name, where))
print("total", sum(hits.values()))
