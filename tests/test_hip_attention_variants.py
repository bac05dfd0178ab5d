"""GPU: the alternative attention-forward kernels (selected by environment variables that the library reads once) against the
fp32 oracle op, each in its own process: MVIT_ATT_PIPE=0 (tile-by-tile 4-wave kernel) and MVIT_ATT_W64=1 (64 queries per
wave, accumulators in asm-owned ACC registers)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from aicity_action_amd import _hip
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
worst = 0.0
for (B, h, Lq, Lk, add_q) in [(2, 2, 300, 200, 1), (1, 1, 1000, 1569, 0), (2, 4, 257, 64, 1), (1, 2, 64, 1, 1)]:
    g = torch.Generator().manual_seed(Lq + Lk)
    q = (torch.randn(B, h, Lq, 96, generator=g) * 1.5).bfloat16(); k = (torch.randn(B, h, Lk, 96, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, h, Lk, 96, generator=g).bfloat16()
    # slowfast/models/attention.py:267-279: softmax((q k^T) * scale) v (+ q), heads merged to [B, Lq, h*96]
    a = ((q.float() @ k.float().transpose(-1, -2)) * 96 ** -0.5).softmax(-1) @ v.float()
    if add_q: a = a + q.float()
    ref = a.transpose(1, 2).reshape(B, Lq, h * 96)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    out = torch.empty(B, Lq, h * 96, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * h * Lq, dtype=torch.float32, device="cuda")
    _hip.check(L.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, 96 ** -0.5,
                                    add_q, _hip.BF16, st))
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    s = (q.float() @ k.float().transpose(-1, -2)) * 96 ** -0.5
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    lerr = (lse.cpu().view(B, h, Lq) - lse_ref).abs().max().item()
    worst = max(worst, err)
    assert err <= 2e-2, (B, h, Lq, Lk, err)
    assert lerr <= 2e-2, (B, h, Lq, Lk, lerr)
print("OK worst %%.2e" %% worst)
"""


@pytest.mark.parametrize("env", [{"MVIT_ATT_PIPE": "0"}, {"MVIT_ATT_W64": "1"}, {"MVIT_ATT_SLOT": "1"}, {}])
def test_attention_forward_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK worst" in r.stdout
