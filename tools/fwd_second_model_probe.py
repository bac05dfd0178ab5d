#!/usr/bin/env python3
"""Probe: why is the second inference model built in a process slower (17 vs 14 ms per forward)?  Prints allocator counters over
the timed window of each model."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config
from aicity_action_amd.models import build_model
from aicity_action_amd.utils.synth import load_synth_weights

clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
models = []
for i, prec in enumerate(["bf16", "bf16", "fp16"]):
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", prec, "HIP.STREAMS", 2])
    m = build_model(cfg).eval()
    load_synth_weights(m, 0)
    with torch.no_grad():
        for _ in range(5):
            m([clip])
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        for _ in range(20):
            m([clip])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        s1 = torch.cuda.memory_stats()
    print("model %d (%s) %.3f ms  device_alloc +%d device_free +%d  reserved %.1f GB  active %.1f GB  streams-known" % (
        i, prec, dt, s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
        s1["reserved_bytes.all.current"] / 1e9, s1["active_bytes.all.current"] / 1e9), flush=True)
    if mode == "keep":
        models.append(m)
    else:
        del m
