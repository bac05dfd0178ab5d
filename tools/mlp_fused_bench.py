"""Kernel time of the fused block tail (mvit_mlp_fused_fwd) against the three launches it replaces (LayerNorm + fc1/GELU + fc2/residual)
on the model's shapes at B = 8 @448.  python tools/mlp_fused_bench.py [half]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aicity_action_amd import _hip

half = sys.argv[1] if len(sys.argv) > 1 else "fp16"
L = _hip.lib(half)
adt = torch.float16 if half == "fp16" else torch.bfloat16
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, C in [(50176, 384), (18816, 384), (12544, 384), (200704, 192), (802816, 96)]:
    hid = 4 * C
    x = torch.randn(M, C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * 0.02, torch.zeros(hid, device=dev)
    w2, b2 = torch.randn(C, hid, device=dev) * 0.02, torch.zeros(C, device=dev)
    packed = torch.empty(L.mvit_mlp_fused_pack_bytes(C, hid), dtype=torch.uint8, device=dev)
    _hip.check(L.mvit_mlp_fused_pack(_hip.ptr(w1), _hip.ptr(b1), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(w2), _hip.ptr(packed), C, hid, st()))
    out = torch.empty_like(x)
    t_f = timed(lambda: _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(b2), _hip.ptr(out), M, C, hid, 1e-6, _hip.BF16, st())))
    w1h, w2h = w1.to(adt), w2.to(adt)
    vn = torch.empty(M, C, dtype=adt, device=dev)
    hd = torch.empty(M, hid, dtype=adt, device=dev)
    o2 = torch.empty_like(x)

    def unfused():
        _hip.check(L.mvit_layernorm_fwd(_hip.ptr(x), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(vn), M, C, 1e-6, _hip.BF16, st()))
        _hip.check(L.mvit_linear_fwd(_hip.ptr(vn), _hip.BF16, C, _hip.ptr(w1h), _hip.ptr(b1), None, hid, None, 0, _hip.ptr(hd), _hip.BF16, hid, M, hid, C,
                                     _hip.EPI_BIAS | _hip.EPI_GELU, _hip.BF16, st()))
        _hip.check(L.mvit_linear_fwd(_hip.ptr(hd), _hip.BF16, hid, _hip.ptr(w2h), _hip.ptr(b2), _hip.ptr(x), C, None, 0, _hip.ptr(o2), _hip.F32, C, M, C, hid,
                                     _hip.EPI_BIAS | _hip.EPI_RESIDUAL, _hip.BF16, st()))
    t_u = timed(unfused)
    fl = 16.0 * M * C * C
    print("[%s] M=%6d C=%3d  fused %7.1f us (%6.1f TFLOP/s)   LN + fc1 + fc2 %7.1f us   max|diff| %.2e" % (
        half, M, C, t_f, fl / t_f / 1e6, t_u, (out - o2).abs().max().item()))
