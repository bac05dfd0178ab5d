"""One shape of the fused block tail (mvit_block_tail_fwd, or mvit_mlp_fused_fwd with mode=mlp) run `reps` times: the target of rocprofv3 --pmc /
--kernel-trace in tools/r4_profiles.sh.   python tools/block_tail_bench.py M C [tail|mlp] [reps] [half]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aicity_action_amd import _hip
M, C = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "tail"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
half = sys.argv[5] if len(sys.argv) > 5 else "fp16"
L = _hip.lib(half)
adt = torch.float16 if half == "fp16" else torch.bfloat16
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
hid = 4 * C
o = torch.randn(M, C, device=dev).to(adt)
res = torch.randn(M, C, device=dev)
wp, bp = torch.randn(C, C, device=dev) * 0.02, torch.zeros(C, device=dev)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w1, b1 = torch.randn(hid, C, device=dev) * 0.02, torch.zeros(hid, device=dev)
w2, b2 = torch.randn(C, hid, device=dev) * 0.02, torch.zeros(C, device=dev)
out = torch.empty_like(res)
if mode == "tail":
    pk = torch.empty(L.mvit_block_tail_pack_bytes(C, hid), dtype=torch.uint8, device=dev)
    _hip.check(L.mvit_block_tail_pack(*[_hip.ptr(t) for t in (wp, bp, w1, b1, gam, bet, w2)], _hip.ptr(pk), C, hid, st()))
    fn = lambda: _hip.check(L.mvit_block_tail_fwd(_hip.ptr(o), _hip.ptr(res), _hip.ptr(pk), _hip.ptr(b2), _hip.ptr(out), M, C, hid, 1e-6,
                                                   _hip.BF16, st()))
    flop = 16.0 * M * C * C + 2.0 * M * C * C
else:
    pk = torch.empty(L.mvit_mlp_fused_pack_bytes(C, hid), dtype=torch.uint8, device=dev)
    _hip.check(L.mvit_mlp_fused_pack(*[_hip.ptr(t) for t in (w1, b1, gam, bet, w2)], _hip.ptr(pk), C, hid, st()))
    fn = lambda: _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(res), _hip.ptr(pk), _hip.ptr(b2), _hip.ptr(out), M, C, hid, 1e-6, _hip.BF16, st()))
    flop = 16.0 * M * C * C
for _ in range(3):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print("%s %s M=%d C=%d: %.1f us  %.1f TFLOP/s (algorithmic %.2f GFLOP)" % (mode, half, M, C, us, flop / us / 1e6, flop / 1e9))
