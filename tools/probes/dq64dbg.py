import torch, sys, os
sys.path.insert(0, "/root/repo")
from aicity_action_amd import _hip
L = _hip.lib(); dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
def run(B, h, Lq, Lk, add_q=0):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, h, Lq, 96, generator=g).bfloat16(); k = torch.randn(B, h, Lk, 96, generator=g).bfloat16(); v = torch.randn(B, h, Lk, 96, generator=g).bfloat16()
    do = torch.randn(B, Lq, h * 96, generator=g).bfloat16()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    o = ((qf @ kf.transpose(-2, -1)) * 96 ** -0.5).softmax(-1) @ vf
    if add_q: o = o + qf
    o.transpose(1, 2).reshape(B, Lq, h * 96).backward(do.float())
    qd, kd, vd, dod = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    out = torch.empty(B, Lq, h * 96, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, h, Lq, device=dev)
    _hip.check(L.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, 96 ** -0.5, add_q, _hip.BF16, st))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ws = torch.empty(L.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk) // 4, device=dev)
    _hip.check(L.mvit_attention_bwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), _hip.ptr(dod), _hip.ptr(dq), _hip.ptr(dk), _hip.ptr(dv), _hip.ptr(ws), B, h, Lq, Lk, 96 ** -0.5, add_q, _hip.BF16, st))
    torch.cuda.synchronize()
    # which keys does the kernel's dq account for?  reference restricted to the full tiles / to the ragged tail only
    nfull = (Lk // 64) * 64
    for name, sl in (("full tiles only", slice(0, nfull)), ("ragged tail only", slice(nfull, Lk))):
        if nfull == Lk: break
        q2 = q.float().requires_grad_(True)
        sc = (q2 @ k.float().transpose(-2, -1)) * 96 ** -0.5
        p = sc.softmax(-1)                       # full softmax, but only the selected keys' terms of dq = scale * sum_k dS K
        dp = (do.float().reshape(B, Lq, h, 96).transpose(1, 2)) @ v.float().transpose(-2, -1)
        dlt = (p * dp).sum(-1, keepdim=True)
        ds = p * (dp - dlt)
        dq_part = (ds[..., sl] @ k.float()[..., sl, :]) * 96 ** -0.5
        print("   vs %-17s: max diff %.3e" % (name, (dq.float().cpu() - dq_part).abs().max().item()))
    d = dq.float().cpu() - qf.grad
    bad = ~torch.isfinite(d)
    rows = sorted(set(bad.any(-1).nonzero()[:, 2].tolist()))
    print("Lq %5d Lk %5d: dq max err %.3e (scale %.2f) non-finite %d rows %s | lse finite %s" % (Lq, Lk, d[~bad].abs().max().item(), qf.grad.abs().max().item(), int(bad.sum()), rows[:8], bool(torch.isfinite(lse).all())))
for Lq, Lk in ((256, 200), (256, 392), (256, 224), (256, 1568), (256, 256), (257, 392), (5000, 200)):
    run(1, 1, Lq, Lk)
