#!/usr/bin/env python3
"""Vendor-library reference point for the weight-gradient GEMMs (dW[N][K] = dy[M][N]^T a[M][K], contraction over the tokens):
what torch.matmul (hipBLASLt / rocBLAS) reaches on the model's shapes.  Not part of the product path."""
import torch

SHAPES = [(50176, 1536, 384), (50176, 384, 1536), (50176, 1152, 384), (50176, 384, 384), (200704, 768, 192), (200704, 576, 192),
          (802816, 384, 96), (12544, 3072, 768), (12544, 2304, 768)]
dev = torch.device("cuda:0")
for M, N, K in SHAPES:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    fn = lambda: torch.matmul(dy.t(), a)
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
    print("wgrad M=%d N=%d K=%d  %8.1f us  %7.1f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
