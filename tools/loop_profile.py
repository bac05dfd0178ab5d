#!/usr/bin/env python3
"""Host side of the engine's training loop (engine.train_epoch over an in-memory loader): wall time per iteration as the host sees
it, completed time, and a cProfile of the iterations -- what separates `bench.py --mode loop` from `--mode train`."""
import cProfile
import logging
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd import engine  # noqa: E402
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.solver import construct_optimizer  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402

logging.getLogger("aicity_action_amd.engine").setLevel(logging.WARNING)
cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", "bf16"])
cfg.LOG_PERIOD = 10
model = build_model(cfg).train()
load_synth_weights(model, 0)
opt = construct_optimizer(model, cfg)
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
labels = torch.arange(8, device="cuda") % cfg.MODEL.NUM_CLASSES


class _Loader(list):
    pass


def run_epoch(n, epoch):
    loader = _Loader([([clip], labels, torch.arange(8), {})] * n)
    engine.train_epoch(loader, model, opt, None, engine.TrainMeter(n, cfg), epoch, cfg)


run_epoch(5, 0)
torch.cuda.synchronize()
for n in (20, 20):
    t0 = time.perf_counter()
    run_epoch(n, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("epoch of %d iterations: host returns after %.1f ms/iter, completed %.1f ms/iter" % (n, (t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n))
pr = cProfile.Profile()
pr.enable()
run_epoch(20, 2)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
