"""Timing of one build of the fused block-tail kernel (MVIT_HIP_LIB selects an ablation / stamp build of tools/build_mf_abl.sh):
M = 12544 (98 tiles: one round, = the time of ONE tile) and M = 50176 (392 tiles) at C = 384; C = 192 and 96 at the model's sizes.
With a stamp build prints the per-iteration cycle split instead."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aicity_action_amd import _hip
L = _hip.lib("bf16")
dev = "cuda:0"
stamp = "stamp" in os.environ.get("MVIT_HIP_LIB", "")
st = lambda: torch.cuda.current_stream().cuda_stream
for M, C in [(12544, 384), (50176, 384), (25088, 192), (200704, 192), (65536, 96), (802816, 96)]:
    hid = 4 * C
    x = torch.randn(M, C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * 0.02, torch.zeros(hid, device=dev)
    w2, b2 = torch.randn(C, hid, device=dev) * 0.02, torch.zeros(C, device=dev)
    packed = torch.empty(L.mvit_mlp_fused_pack_bytes(C, hid), dtype=torch.uint8, device=dev)
    _hip.check(L.mvit_mlp_fused_pack(_hip.ptr(w1), _hip.ptr(b1), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(w2), _hip.ptr(packed), C, hid, st()))
    out = torch.zeros_like(x)
    fn = lambda: _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(b2), _hip.ptr(out), M, C, hid, 1e-6, _hip.BF16, st()))
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    line = "%-28s M=%6d C=%3d  %7.1f us (%6.1f TFLOP/s)" % (os.path.basename(os.environ.get("MVIT_HIP_LIB", "default")), M, C, us, 16.0 * M * C * C / us / 1e6)
    if stamp:
        tiles = (M + (128 if C == 384 else 256) - 1) // (128 if C == 384 else 256)
        t = out.flatten()[:tiles * 32].view(tiles, 4, 8).cpu()
        line += "   cycles/iteration: wait+barrier %.0f  phase1 %.0f  phase2 %.0f | prologue %.0f (min %.0f max %.0f)  loop %.0f cycles" % (
            t[:, :, 0].mean(), t[:, :, 1].mean(), t[:, :, 2].mean(), t[:, :, 3].mean(), t[:, :, 3].min(), t[:, :, 3].max(), t[:, :, 4].mean())
    print(line)
