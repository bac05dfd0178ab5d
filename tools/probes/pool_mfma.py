"""GPU probe of the matrix-core pooling conv (tools/probes/pool_mfma.hip) at the model's stage-3 q-pool shape (B = 8, 4 heads, 8 x 28 x 28
tokens, stride 1): checked against conv3d, timed against the production kernel (conv + LayerNorm: mvit_pool_conv_ln_fwd)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from aicity_action_amd import _hip
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "pool_mfma.so"))
lib.pool_mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
B, h, T, H, W = 8, 4, 8, 28, 28
C = 96 * h
N = T * H * W
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16)
w = torch.randn(96, 27, device=dev) * 0.3
out = torch.zeros(B * h, N, 96, device=dev, dtype=torch.bfloat16)
st = lambda: torch.cuda.current_stream().cuda_stream
rc = lib.pool_mfma_probe_launch(qkv.data_ptr(), 3 * C, 0, h, w.data_ptr(), out.data_ptr(), B * h, st()); assert rc == 0, rc
torch.cuda.synchronize()
x = qkv[:, :, :C].float().reshape(B, T, H, W, h, 96).permute(0, 4, 5, 1, 2, 3).reshape(B * h, 96, T, H, W)
ref = F.conv3d(x, w.to(torch.bfloat16).float().reshape(96, 1, 3, 3, 3), padding=1, groups=96).reshape(B * h, 96, N).transpose(1, 2)
err = (out.float() - ref).abs().max().item()
print("conv (16-bit weights, fp32 accumulate) vs conv3d: max|err| %.3e (scale %.2f)" % (err, ref.abs().max().item()))


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_probe = timed(lambda: lib.pool_mfma_probe_launch(qkv.data_ptr(), 3 * C, 0, h, w.data_ptr(), out.data_ptr(), B * h, st()))
L = _hip.lib("bf16")
gam, bet = torch.ones(96, device=dev), torch.zeros(96, device=dev)
q = torch.empty(B, h, N, 96, device=dev, dtype=torch.bfloat16)
w5 = w.reshape(96, 1, 3, 3, 3).contiguous()
t_prod = timed(lambda: _hip.check(L.mvit_pool_conv_ln_fwd(_hip.ptr(qkv), 3 * C, 0, _hip.ptr(w5), _hip.ptr(gam), _hip.ptr(bet), _hip.ptr(q), B, h, T, H, W, 1, 1e-5,
                                                           _hip.BF16, st())))
print("probe (conv only, matrix cores, simple staging) %.1f us   production pool_march (conv + LayerNorm) %.1f us" % (t_probe, t_prod))
