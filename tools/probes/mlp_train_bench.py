"""(needs tools/probes/mlp_fused_train.patch applied) tools/probes/mlp_train_bench.py M C [half] [reps]: the training forward of the MLP branch, fused (mvit_mlp_fused_train_fwd, + the pack it needs every
step) against the launches it replaces (fc1 + GELU with both 16-bit outputs, fc2 + residual).  Prints us per launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aicity_action_amd import _hip

M, C = int(sys.argv[1]), int(sys.argv[2])
half = sys.argv[3] if len(sys.argv) > 3 else "bf16"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
L = _hip.lib(half)
dev = torch.device("cuda:0")
hdt = torch.bfloat16 if half == "bf16" else torch.float16
hid = 4 * C
g = torch.Generator(device="cpu").manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
y = rnd(M, C)
gam, bet = 1 + 0.1 * rnd(C), 0.1 * rnd(C)
w1, b1, w2, b2 = 0.05 * rnd(hid, C), 0.1 * rnd(hid), 0.05 * rnd(C, hid), 0.1 * rnd(C)
st = torch.cuda.current_stream().cuda_stream
packed = torch.empty(L.mvit_mlp_fused_pack_bytes(C, hid), dtype=torch.uint8, device=dev)
out = torch.empty(M, C, device=dev)
h16, d16 = torch.empty(M, hid, dtype=hdt, device=dev), torch.empty(M, hid, dtype=hdt, device=dev)
vn = torch.empty(M, C, dtype=hdt, device=dev)
w1h, w2h = w1.to(hdt), w2.to(hdt)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def pack():
    _hip.check(L.mvit_mlp_fused_pack(_hip.ptr(w1), _hip.ptr(b1), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(w2), _hip.ptr(packed), C, hid, st))


def fused():
    _hip.check(L.mvit_mlp_fused_train_fwd(_hip.ptr(y), _hip.ptr(packed), _hip.ptr(b2), None, 0, _hip.ptr(out), _hip.ptr(h16), _hip.ptr(d16), M, C, hid, 1e-6,
                                          _hip.BF16, st))


def fc1():
    _hip.check(L.mvit_linear_gelu_fwd_dsave(_hip.ptr(vn), C, _hip.ptr(w1h), _hip.ptr(b1), _hip.ptr(d16), _hip.ptr(h16), M, hid, C, _hip.BF16, st))


pack()
print("M=%d C=%d %s: pack %.1f us | fused train fwd %.1f us | fc1+GELU (two 16-bit outputs) %.1f us (+ LayerNorm ~20 us and fc2 + residual, see the model profile)"
      % (M, C, half, timeit(pack), timeit(fused), timeit(fc1)))
