"""Attention forward + backward on rows with PEAKED scores (planted keys), bf16 library: errors of out / lse / dq / dk / dv against
fp32 autograd on the same 16-bit operands, relative to each tensor's max |value|.  usage: python3 tools/probes/attn_peaked.py [c ...]
MVIT_HIP_LIB=.../libmvit_hip_f16.so PEAKED_DTYPE=fp16 runs the fp16 build.  Negative c: the planted key is -|c| q (a strongly NEGATIVE score, so the
row maximum sits in the Gaussian bulk); c >= 100: (c - 100) q is planted in the FIRST key of every row's first visited (ragged) tile and in
key 0, the rest of the row is Gaussian."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aicity_action_amd import _hip
L = _hip.lib()
DEV = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
B, h, Lq, Lk = 1, 2, 512, 1568
scale = 96 ** -0.5
for c in [float(a) for a in sys.argv[1:]] or [0.0, 0.5, 1.0, 2.0, 3.0]:
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, h, n, 96, generator=g) for n in (Lq, Lk, Lk))
    do = torch.randn(B, Lq, h * 96, generator=g)
    if c >= 100:
        for i in range(0, Lq, 2):
            k[0, :, Lk - 1 - (i % 32)] = q[0, :, i] * (c - 100)
    elif c:
        for i in range(0, Lq, 2):                     # every second query row has one dominant key
            k[0, :, (37 * i) % Lk] = q[0, :, i] * c
    DT = torch.float16 if os.environ.get("PEAKED_DTYPE") == "fp16" else torch.bfloat16
    q, k, v, do = (t.to(DT) for t in (q, k, v, do))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-2, -1)) * scale
    o = (s.softmax(-1) @ vr + qr).transpose(1, 2).reshape(B, Lq, h * 96)
    o.backward(do.float())
    lse_ref = torch.logsumexp(s.detach(), -1) * 1.4426950408889634
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    out = torch.empty(B, Lq, h * 96, dtype=DT, device=DEV)
    lse = torch.empty(B, h, Lq, device=DEV)
    _hip.check(L.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, scale, 1, _hip.BF16, st()), "fwd")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ws = torch.empty(L.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk) // 4, device=DEV)
    _hip.check(L.mvit_attention_bwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), _hip.ptr(dod), _hip.ptr(dq), _hip.ptr(dk),
                                    _hip.ptr(dv), _hip.ptr(ws), B, h, Lq, Lk, scale, 1, _hip.BF16, st()), "bwd")
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    print("c=%.1f max score (log2 units) %5.1f | out %.2e lse(abs) %.2e dq %.2e dk %.2e dv %.2e" % (
        c, (s.detach().max() * 1.4427).item(), rel(out, o.detach()), (lse.cpu() - lse_ref).abs().max().item(), rel(dq, qr.grad), rel(dk, kr.grad), rel(dv, vr.grad)), flush=True)
