"""mvit_pool_conv_ln_fwd at the model's stride-1 q-pool shapes, matrix-core form (csrc/pool_mfma.hip) against the VALU kernels (MVIT_POOL_MFMA=0 in a
second process): python tools/pool_mfma_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aicity_action_amd import _hip
half = sys.argv[1] if len(sys.argv) > 1 else "fp16"
L = _hip.lib(half)
adt = torch.float16 if half == "fp16" else torch.bfloat16
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
for B, h, T, H, W in [(8, 4, 8, 28, 28), (8, 2, 8, 56, 56), (8, 1, 8, 112, 112), (4, 4, 8, 28, 28)]:
    C = 96 * h; N = T * H * W
    qkv = torch.randn(B, N, 3 * C, device=dev).to(adt)
    w = (torch.randn(96, 1, 3, 3, 3, device=dev) * 0.3).contiguous()
    gam, bet = torch.ones(96, device=dev), torch.zeros(96, device=dev)
    q = torch.empty(B, h, N, 96, device=dev, dtype=adt)
    fn = lambda: _hip.check(L.mvit_pool_conv_ln_fwd(_hip.ptr(qkv), 3 * C, 0, _hip.ptr(w), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(q), B, h, T, H, W, 1, 1e-5, _hip.BF16, st()))
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); e1.synchronize()
    print("MVIT_POOL_MFMA=%s %s B=%d heads=%d %dx%dx%d stride 1: %.1f us" % (os.environ.get("MVIT_POOL_MFMA", "1"), half, B, h, T, H, W, e0.elapsed_time(e1) / 30 * 1e3))
