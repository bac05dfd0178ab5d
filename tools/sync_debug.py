#!/usr/bin/env python3
"""Lists the synchronizing torch calls of one training step (torch.cuda.set_sync_debug_mode("warn"))."""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402

cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", "bf16"])
model = build_model(cfg).train()
load_synth_weights(model, 0)
opt = construct_optimizer(model, cfg)
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
labels = torch.zeros(8, cfg.MODEL.NUM_CLASSES, device="cuda")
labels[torch.arange(8, device="cuda"), torch.arange(8, device="cuda") % cfg.MODEL.NUM_CLASSES] = 1.0


def step():
    opt.set_lr(1e-4)
    loss = soft_target_cross_entropy(model([clip]), labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
print("synchronizing calls in one step:", len(w))
for x in w[:20]:
    print("  ", x.filename.replace(ROOT + "/", ""), x.lineno, str(x.message)[:100])
