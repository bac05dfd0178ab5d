#!/usr/bin/env python3
"""Per-block forward / backward time of one training step (HIP events around _BlockFn.forward / .backward, side streams off):
where in the network the step's 53 ms go.  usage (GPU box): MVIT_NO_SIDE_STREAM=1 MVIT_WGRAD_STREAM=0 python tools/block_times.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd import autograd as A  # noqa: E402
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402

cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", "1", "HIP.WGRAD_STREAM", "False"])
model = build_model(cfg)
load_synth_weights(model)
model.train()
opt = construct_optimizer(model, cfg)
B = 8
clip = torch.randn(B, 3, 16, 448, 448, device="cuda")
labels = torch.zeros(B, cfg.MODEL.NUM_CLASSES, device="cuda")
labels[torch.arange(B), torch.arange(B) % cfg.MODEL.NUM_CLASSES] = 1.0
ev = {"f": {}, "b": {}}
of, ob = A._BlockFn.forward, A._BlockFn.backward


def wrap(kind, fn, idx_of):
    def inner(ctx, *a):
        i = idx_of(ctx, a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(ctx, *a)
        e1.record()
        ev[kind].setdefault(i, []).append((e0, e1))
        return r
    return staticmethod(inner)


A._BlockFn.forward = wrap("f", of, lambda ctx, a: a[2].index)
A._BlockFn.backward = wrap("b", ob, lambda ctx, a: ctx.g.index)
for it in range(6):
    if it == 2:
        ev["f"].clear(); ev["b"].clear()
    loss = soft_target_cross_entropy(model([clip]), labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
tf = tb = 0.0
for i in sorted(ev["f"]):
    f = sum(a.elapsed_time(b) for a, b in ev["f"][i]) / len(ev["f"][i])
    b = sum(a.elapsed_time(b_) for a, b_ in ev["b"][i]) / len(ev["b"][i])
    g = model.geoms[i]
    print("block %2d  dim %3d->%3d heads %d  tokens %6d -> q %6d kv %5d   fwd %6.3f ms  bwd %6.3f ms" % (i, g.dim_in, g.dim_out, g.heads, g.n_in, g.lq, g.lk, f, b))
    tf += f; tb += b
print("blocks total: fwd %.2f ms  bwd %.2f ms" % (tf, tb))
