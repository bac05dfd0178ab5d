import torch, sys
sys.path.insert(0, "/root/repo")
from aicity_action_amd import _hip
L = _hip.lib(); dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
def run(B, h, Lq, Lk):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, h, Lq, 96, generator=g).bfloat16(); k = torch.randn(B, h, Lk, 96, generator=g).bfloat16(); v = torch.randn(B, h, Lk, 96, generator=g).bfloat16()
    ref = ((q.float() @ k.float().transpose(-2, -1)) * 96 ** -0.5).softmax(-1) @ v.float()
    ref = ref.transpose(1, 2).reshape(B, Lq, h * 96)
    o = torch.empty(B, Lq, h * 96, device=dev, dtype=torch.bfloat16)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    _hip.check(L.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(o), None, B, h, Lq, Lk, 96 ** -0.5, 0, _hip.BF16, st))
    torch.cuda.synchronize()
    d = (o.float().cpu() - ref)
    bad = ~torch.isfinite(d)
    print("Lq %5d Lk %5d nkt %3d: max err %.3e, non-finite %d, rows with error > 0.05: %s" % (Lq, Lk, (Lk + 63) // 64, d[~bad].abs().max().item(), int(bad.sum()),
          sorted(set((d.abs() > 0.05).any(-1).nonzero()[:, 1].tolist()))[:12]))
for Lq, Lk in ((256, 448), (256, 392), (256, 320), (256, 384), (256, 192), (256, 128), (256, 200), (256, 1568), (256, 1600)):
    run(1, 1, Lq, Lk)
