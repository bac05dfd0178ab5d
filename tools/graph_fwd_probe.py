#!/usr/bin/env python3
"""Probe: forward (eval, B=8 @448, HIP.STREAMS 2) eager vs replayed as one hipGraph.  usage: python tools/graph_fwd_probe.py [streams]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config  # noqa: E402
from aicity_action_amd.models import build_model  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights  # noqa: E402

streams = sys.argv[1] if len(sys.argv) > 1 else "2"
cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", "1", "HIP.STREAMS", streams])
model = build_model(cfg)
load_synth_weights(model)
model.eval()
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = model([clip]).clone()
    eager = timeit(lambda: model([clip]))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model([clip])
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model([clip])
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    graphed = timeit(g.replay)
print("streams=%s eager %.3f ms (%.1f clips/s)  graph replay %.3f ms (%.1f clips/s)  identical=%s" % (streams, eager, 8e3 / eager, graphed, 8e3 / graphed, same))
