#!/usr/bin/env python3
"""Which torch (ATen) ops run inside one training step, and from where (torch.profiler with stacks)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=25, max_name_column_width=40, max_src_column_width=90)[:9000])
