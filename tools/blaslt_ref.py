#!/usr/bin/env python3
"""What the vendor library (hipBLASLt / rocBLAS through torch.matmul) reaches on the model's GEMM shapes: a reference point for
the hand-written kernels, not part of the product path.  usage: tools/blaslt_ref.py  (prints us and TFLOP/s per shape)"""
import torch

SHAPES = [(50176, 1152, 384), (50176, 1536, 384), (50176, 384, 384), (50176, 384, 1536), (200704, 576, 192), (200704, 768, 192),
          (200704, 192, 768), (802816, 288, 96), (802816, 384, 96), (802816, 96, 384), (12544, 2304, 768), (12544, 3072, 768),
          (12544, 768, 3072), (8192, 8192, 8192)]
dev = torch.device("cuda:0")
for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for name, fn in (("nt", lambda: torch.nn.functional.linear(a, w)), ("nt+bias", lambda: torch.nn.functional.linear(a, w, bias))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("M=%d N=%d K=%d %-8s %8.1f us  %7.1f TFLOP/s" % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6), flush=True)
