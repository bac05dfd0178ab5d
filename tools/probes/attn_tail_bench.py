"""Attention forward at the model's stage-3 / stage-4 shapes as the model calls it (_hip.attention_fwd: key-split ragged tile unless
MVIT_ATT_TAIL_SPLIT=0), HIP-event time per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aicity_action_amd import _hip
L = _hip.lib("fp16")
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
for B, h, Lq, Lk in [(8, 4, 6272, 1568), (8, 4, 6272, 6272), (8, 8, 1568, 6272), (8, 8, 1568, 1568), (3, 4, 6272, 1568)]:
    q = torch.randn(B, h, Lq, 96, device=dev).half()
    k = torch.randn(B, h, Lk, 96, device=dev).half()
    v = torch.randn(B, h, Lk, 96, device=dev).half()
    o = torch.empty(B, Lq, h * 96, device=dev, dtype=torch.float16)
    lse = torch.empty(B, h, Lq, device=dev)
    fn = lambda: _hip.check(_hip.attention_fwd(L, q, k, v, o, lse, B, h, Lq, Lk, 96 ** -0.5, 1, _hip.BF16, st()))
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("B=%d heads=%d Lq=%5d Lk=%5d  %7.1f us  %6.1f TFLOP/s" % (B, h, Lq, Lk, us, 4.0 * B * h * Lq * Lk * 96 / us / 1e6))
