"""GPU: every backward / training-step C-ABI operator against torch autograd of the oracle's op on the CPU."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mvit_oracle as O
from aicity_action_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _act(t, act):
    return t.to(torch.bfloat16) if act == _hip.BF16 else t.float()


def _close(got, ref, tol):
    got = got.float().cpu()
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "max err %.3e > %.3e (scale %.3f)" % (err, tol * scale, scale)


@pytest.mark.parametrize("dyt", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("C,rows", [(96, 1000), (192, 37), (384, 392), (768, 65)])
def test_layernorm_bwd(hip_lib, dyt, C, rows):
    x = (_rnd(rows, C, seed=1) * 2 + 0.5).requires_grad_(True)
    g = (1 + 0.1 * _rnd(C, seed=2)).requires_grad_(True)
    b = (0.1 * _rnd(C, seed=3)).requires_grad_(True)
    dy = _act(_rnd(rows, C, seed=4), dyt)
    F.layer_norm(x, (C,), g, b, 1e-6).backward(dy.float())
    base = _rnd(rows, C, seed=5)
    dx = base.clone().to(DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ws = torch.empty(hip_lib.mvit_layernorm_bwd_workspace_bytes(C) // 4, device=DEV)
    xd, gd, dyd = x.detach().to(DEV), g.detach().to(DEV), dy.to(DEV)
    _hip.check(hip_lib.mvit_layernorm_bwd(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(dyd), dyt, 1, 1.0, _hip.ptr(dx), _hip.ptr(dx), _hip.ptr(dg),
                                          _hip.ptr(db), 0, _hip.ptr(ws), rows, C, 1e-6, None, None, 0, _st()))     # dx_base = dx: accumulate in place
    _close(dx, x.grad + base, 2e-5)
    _close(dg, g.grad, 2e-5)
    _close(db, b.grad, 2e-5)


def test_layernorm_bwd_broadcast_mode(hip_lib):
    B, N, C = 3, 50, 384
    x = (_rnd(B * N, C, seed=6) + 0.2).requires_grad_(True)
    g = (1 + 0.1 * _rnd(C, seed=7)).requires_grad_(True)
    b = torch.zeros(C, requires_grad=True)
    dz = _rnd(B, C, seed=8)
    (F.layer_norm(x, (C,), g, b, 1e-6).reshape(B, N, C).mean(1) * dz).sum().backward()
    dx = torch.empty(B * N, C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ws = torch.empty(hip_lib.mvit_layernorm_bwd_workspace_bytes(C) // 4, device=DEV)
    xd, gd, dzd = x.detach().to(DEV), g.detach().to(DEV), dz.to(DEV)
    _hip.check(hip_lib.mvit_layernorm_bwd(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(dzd), _hip.F32, N, 1.0 / N, None, _hip.ptr(dx),
                                          _hip.ptr(dg), _hip.ptr(db), 0, _hip.ptr(ws), B * N, C, 1e-6, None, None, 0, _st()))
    _close(dx, x.grad, 2e-5)
    _close(dg, g.grad, 2e-5)
    _close(db, b.grad, 2e-5)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
def test_gelu_fwd_bwd(hip_lib, act):
    n = 4096 * 3
    x = _act(_rnd(n, seed=9) * 2, act)
    dy = _act(_rnd(n, seed=10), act)
    xr = x.float().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(dy.float())
    y = torch.empty_like(x, device=DEV)
    dx = torch.empty_like(x, device=DEV)
    xd, dyd = x.to(DEV), dy.to(DEV)
    _hip.check(hip_lib.mvit_gelu_fwd(_hip.ptr(xd), _hip.ptr(y), n, act, _st()))
    _hip.check(hip_lib.mvit_gelu_bwd(_hip.ptr(xd), _hip.ptr(dyd), _hip.ptr(dx), n, act, _st()))
    tol = 1e-2 if act else 1e-6
    _close(y, yr.detach(), tol)
    _close(dx, xr.grad, tol)


@pytest.mark.parametrize("rows,cols,rps", [(1000, 96, 0), (777, 384, 100), (64, 768, 64)])
def test_cast_rows_with_drop_path_scale(hip_lib, rows, cols, rps):
    x = _rnd(rows, cols, seed=31)
    sc = torch.rand((rows + rps - 1) // rps, generator=torch.Generator().manual_seed(5)) * 2 if rps else None
    ref = (x * (sc.repeat_interleave(rps)[:rows, None] if rps else 1.0)).to(torch.bfloat16)
    xd = x.to(DEV)
    scd = sc.to(DEV) if rps else None
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_cast_rows_f32_to_bf16(_hip.ptr(xd), _hip.ptr(out), rows, cols, _hip.ptr(scd), rps, _st()))
    assert torch.equal(out.cpu(), ref)       # same fp32 product, same round-to-nearest-even
    assert hip_lib.mvit_cast_rows_f32_to_bf16(_hip.ptr(xd), _hip.ptr(out), rows, 100, None, 0, _st()) == -4


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("M,N,K,scaled", [(1000, 288, 96, False), (2500, 96, 384, True), (129, 384, 192, False),
                                           (3000, 192, 768, True), (700, 576, 192, False), (300, 768, 96, False),
                                           (300, 1152, 96, False), (4096, 384, 384, False), (6336, 384, 192, False), (1024, 288, 96, False), (1280, 96, 384, False),
                                           (640, 576, 192, False), (2048, 96, 96, False), (1920, 192, 768, False),
                                           (64, 384, 192, False)])
def test_linear_wgrad_and_colsum(hip_lib, act, M, N, K, scaled):
    a = _act(_rnd(M, K, seed=11), act)
    for dy_f32 in ([True] if act == _hip.F32 else [True, False]):
        dy = _rnd(M, N, seed=12) if dy_f32 else _rnd(M, N, seed=12).to(torch.bfloat16)
        rps = 500
        sc = torch.tensor([0.0, 1.5, 2.0, 1.0, 0.5, 1.25, 3.0])[: (M + rps - 1) // rps] if scaled else None
        dys = dy.float() * (sc.repeat_interleave(rps)[:M, None] if scaled else 1.0)
        if act == _hip.BF16:
            dys_m = (dy.float() * (sc.repeat_interleave(rps)[:M, None] if scaled else 1.0)).to(torch.bfloat16).float()
        else:
            dys_m = dys
        ref = dys_m.t() @ a.float()
        base = _rnd(N, K, seed=13)
        dW = base.clone().to(DEV)
        dbf = torch.zeros(N, device=DEV)
        ad, dyd = a.to(DEV), dy.to(DEV)
        scd = sc.to(DEV) if scaled else None
        ddt = _hip.F32 if dy_f32 else _hip.BF16
        nb = hip_lib.mvit_linear_wgrad_workspace_bytes(act, K, ddt, N, 1 if scaled else 0, M, N, K, act)
        wsw = torch.empty(max(nb // 4, 1), device=DEV)
        _hip.check(hip_lib.mvit_linear_wgrad(_hip.ptr(ad), act, K, _hip.ptr(dyd), ddt, N, _hip.ptr(scd), rps if scaled else 0,
                                             _hip.ptr(dW), _hip.ptr(dbf), M, N, K, act, _hip.ptr(wsw), nb, _st()))
        assert hip_lib.mvit_linear_wgrad(_hip.ptr(ad), act, K, _hip.ptr(dyd), ddt, N, _hip.ptr(scd), rps if scaled else 0,
                                         _hip.ptr(dW), _hip.ptr(dbf), M, N, K, act, None, 0, _st()) == -1      # no workspace, no atomics form
        _close(dW, ref + base, 2e-3 if act else 2e-5)
        _close(dbf, dys_m.sum(0), 2e-3 if act else 2e-5)
        ws = torch.empty(hip_lib.mvit_colsum_workspace_bytes(N) // 4, device=DEV)
        db = torch.zeros(N, device=DEV)
        _hip.check(hip_lib.mvit_colsum(_hip.ptr(dyd), _hip.F32 if dy_f32 else _hip.BF16, M, N, _hip.ptr(scd),
                                       rps if scaled else 0, _hip.ptr(db), 0, _hip.ptr(ws), _st()))
        _close(db, dys.sum(0), 2e-5)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,h,Lq,Lk,add_q", [(1, 1, 128, 64, 1), (2, 2, 100, 33, 1), (1, 2, 257, 392, 0), (1, 4, 392, 1568, 1),
                                               (2, 1, 64, 200, 1), (1, 1, 5000, 200, 1)])
def test_attention_bwd(hip_lib, act, B, h, Lq, Lk, add_q):
    scale = 96 ** -0.5
    q = _act(_rnd(B, h, Lq, 96, seed=14), act)
    k = _act(_rnd(B, h, Lk, 96, seed=15), act)
    v = _act(_rnd(B, h, Lk, 96, seed=16), act)
    do = _act(_rnd(B, Lq, h * 96, seed=17), act)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    p = ((qr @ kr.transpose(-2, -1)) * scale).softmax(-1)
    o = p @ vr
    if add_q:
        o = o + qr
    o = o.transpose(1, 2).reshape(B, Lq, h * 96)
    o.backward(do.float())
    adt = q.dtype
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    out = torch.empty(B, Lq, h * 96, dtype=adt, device=DEV)
    lse = torch.empty(B, h, Lq, device=DEV)
    _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk,
                                          scale, add_q, act, _st()))
    lse_ref = torch.logsumexp((qr.detach() @ kr.detach().transpose(-2, -1)) * scale, -1) * 1.4426950408889634
    _close(lse, lse_ref, 1e-2 if act else 1e-5)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ws = torch.empty(hip_lib.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk) // 4, device=DEV)
    _hip.check(hip_lib.mvit_attention_bwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), _hip.ptr(dod),
                                          _hip.ptr(dq), _hip.ptr(dk), _hip.ptr(dv), _hip.ptr(ws), B, h, Lq, Lk, scale, add_q,
                                          act, _st()))
    tol = 2e-2 if act else 2e-5
    _close(dq, qr.grad, tol)
    _close(dk, kr.grad, tol)
    _close(dv, vr.grad, tol)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,h,T,H,W,s", [(2, 1, 2, 16, 16, 1), (1, 2, 2, 14, 14, 2), (2, 2, 3, 7, 7, 2), (1, 1, 2, 14, 14, 4),
                                         (1, 2, 2, 5, 9, 1), (1, 2, 3, 14, 14, 1), (2, 1, 4, 28, 28, 1), (1, 1, 2, 28, 14, 1)])
def test_pool_conv_ln_bwd(hip_lib, act, B, h, T, H, W, s):
    C = 96 * h
    N = T * H * W
    qkv = _act(_rnd(B, N, 3 * C, seed=18), act)
    w = _rnd(96, 1, 3, 3, 3, seed=19, scale=0.3).requires_grad_(True)
    g = (1 + 0.1 * _rnd(96, seed=20)).requires_grad_(True)
    b = (0.1 * _rnd(96, seed=21)).requires_grad_(True)
    which = 1
    qkvr = qkv.float().requires_grad_(True)
    x = qkvr[:, :, which * C:(which + 1) * C].reshape(B, N, h, 96).permute(0, 2, 1, 3)
    out, thw = O._pool_conv_ln(x, (T, H, W), w, (1, s, s), g, b)
    Lo = thw[0] * thw[1] * thw[2]
    dout = _act(_rnd(B, h, Lo, 96, seed=22), act)
    out.backward(dout.float())
    adt = qkv.dtype
    dqkv = torch.zeros(B, N, 3 * C, dtype=adt, device=DEV)
    dconv = torch.empty(B, h, Lo, 96, dtype=adt, device=DEV)
    dw = torch.zeros(96, 27, device=DEV)
    dg, db = torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
    ws = torch.empty(hip_lib.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, s) // 4, device=DEV)
    qd, wd, gd, dod = qkv.to(DEV), w.detach().to(DEV), g.detach().to(DEV), dout.to(DEV)
    _hip.check(hip_lib.mvit_pool_conv_ln_bwd(_hip.ptr(qd), 3 * C, which * C, _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(dod),
                                             _hip.ptr(dconv), _hip.ptr(dqkv), _hip.ptr(dw), _hip.ptr(dg), _hip.ptr(db), 0,
                                             _hip.ptr(ws), B, h, T, H, W, s, 1e-5, act, _st()))
    tol = 2e-2 if act else 3e-5
    _close(dqkv[:, :, which * C:(which + 1) * C], qkvr.grad[:, :, which * C:(which + 1) * C], tol)
    assert dqkv[:, :, :C].abs().max().item() == 0            # other slices untouched
    _close(dw, w.grad.reshape(96, 27), tol)
    _close(dg, g.grad, tol)
    _close(db, b.grad, tol)
    # the training pair: forward keeps xhat / rstd, backward runs the row-wise LayerNorm backward from them (no second conv)
    bd = b.detach().to(DEV)
    out2 = torch.empty(B, h, Lo, 96, dtype=adt, device=DEV)
    xh = torch.empty(B, h, Lo, 96, dtype=adt, device=DEV)
    rs = torch.empty(B * h * Lo, device=DEV)
    _hip.check(hip_lib.mvit_pool_conv_ln_fwd_train(_hip.ptr(qd), 3 * C, which * C, _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd),
                                                   _hip.ptr(out2), _hip.ptr(xh), _hip.ptr(rs), B, h, T, H, W, s, 1e-5, act, _st()))
    _close(out2, out.detach(), 1e-2 if act else 2e-5)
    dqkv2 = torch.zeros_like(dqkv)
    dw2, dg2, db2 = torch.zeros(96, 27, device=DEV), torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
    _hip.check(hip_lib.mvit_pool_conv_ln_bwd_saved(_hip.ptr(qd), 3 * C, which * C, _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(xh), _hip.ptr(rs),
                                                   _hip.ptr(dod), _hip.ptr(dconv), _hip.ptr(dqkv2), _hip.ptr(dw2), _hip.ptr(dg2),
                                                   _hip.ptr(db2), 0, _hip.ptr(ws), B, h, T, H, W, s, 1e-5, act, _st()))
    _close(dqkv2[:, :, which * C:(which + 1) * C], qkvr.grad[:, :, which * C:(which + 1) * C], tol)
    _close(dw2, w.grad.reshape(96, 27), tol)
    _close(dg2, g.grad, tol)
    _close(db2, b.grad, tol)


@pytest.mark.parametrize("c", [0.0, 1.0, 3.0])
def test_attention_bwd_rows_with_a_dominant_key(hip_lib, c):
    """Every second query has one key c * q planted (scores up to ~60 in the exp2 domain for c = 3: a trained model's peaked heads).
    The 64-query forward kernel exponentiates the scores of the pre-scaled 16-bit queries and saves their lse; the backward must rebuild
    its probabilities from the SAME rounded queries (dQ pass: in registers; dK/dV pass: the copy the delta kernel leaves in the
    workspace) -- on the unscaled queries P is off by 2^(score * 2^-9) in exactly these rows and dV came out 1-2 % wrong
    (tools/probes/attn_peaked.py).  Reference: fp32 autograd on the same 16-bit operands (attention.py:267-279)."""
    B, h, Lq, Lk = 1, 2, 512, 1568
    scale = 96 ** -0.5
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, h, n, 96, generator=g) for n in (Lq, Lk, Lk))
    do = torch.randn(B, Lq, h * 96, generator=g)
    if c:
        for i in range(0, Lq, 2):
            k[0, :, (37 * i) % Lk] = q[0, :, i] * c
    q, k, v, do = (t.to(torch.bfloat16) for t in (q, k, v, do))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    o = (((qr @ kr.transpose(-2, -1)) * scale).softmax(-1) @ vr + qr).transpose(1, 2).reshape(B, Lq, h * 96)
    o.backward(do.float())
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    out = torch.empty(B, Lq, h * 96, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, h, Lq, device=DEV)
    _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, scale, 1,
                                          _hip.BF16, _st()))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ws = torch.empty(hip_lib.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk) // 4, device=DEV)
    _hip.check(hip_lib.mvit_attention_bwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), _hip.ptr(dod), _hip.ptr(dq),
                                          _hip.ptr(dk), _hip.ptr(dv), _hip.ptr(ws), B, h, Lq, Lk, scale, 1, _hip.BF16, _st()))
    _close(out, o.detach(), 5e-3)
    _close(dv, vr.grad, 7e-3)            # 3.2e-3 ... 4.4e-3 measured; 1.1e-2 ... 2.4e-2 with inconsistent scores
    _close(dq, qr.grad, 2e-2)
    _close(dk, kr.grad, 2e-2)


@pytest.mark.parametrize("B,T,H,W,C", [(2, 2, 16, 16, 192), (1, 3, 7, 7, 384), (1, 1, 5, 9, 96)])
def test_maxpool_skip_bwd(hip_lib, B, T, H, W, C):
    x = _rnd(B, T * H * W, C, seed=23).requires_grad_(True)
    t = x.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    y = F.max_pool3d(t, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    Lo = y.shape[2] * y.shape[3] * y.shape[4]
    dy = _rnd(B, Lo, C, seed=24)
    y.reshape(B, C, Lo).transpose(1, 2).backward(dy)
    dx = torch.empty(B, T * H * W, C, device=DEV)
    xd, dyd = x.detach().to(DEV), dy.to(DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_bwd(_hip.ptr(xd), _hip.ptr(dyd), _hip.ptr(dx), B, T, H, W, C, _st()))
    _close(dx, x.grad, 1e-6)
    # training pair: forward records the first-maximum position, backward routes dy by it
    y2 = torch.empty(B, Lo, C, device=DEV)
    idx = torch.empty(B, Lo, C, dtype=torch.uint8, device=DEV)
    dx2 = torch.empty_like(dx)
    _hip.check(hip_lib.mvit_maxpool_skip_fwd_idx(_hip.ptr(xd), _hip.ptr(y2), _hip.ptr(idx), B, T, H, W, C, _st()))
    _hip.check(hip_lib.mvit_maxpool_skip_bwd_idx(_hip.ptr(idx), _hip.ptr(dyd), _hip.ptr(dx2), B, T, H, W, C, _st()))
    assert torch.equal(y2.cpu(), y.detach().reshape(B, C, Lo).transpose(1, 2))
    _close(dx2, x.grad, 1e-6)


def test_maxpool_skip_idx_ties_go_to_the_first_maximum(hip_lib):
    B, T, H, W, C = 1, 2, 9, 7, 8
    x = torch.randint(0, 3, (B, T * H * W, C), generator=torch.Generator().manual_seed(9)).float().requires_grad_(True)
    t = x.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    y = F.max_pool3d(t, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    Lo = y.shape[2] * y.shape[3] * y.shape[4]
    dy = _rnd(B, Lo, C, seed=24)
    y.reshape(B, C, Lo).transpose(1, 2).backward(dy)
    xd, dyd = x.detach().to(DEV), dy.to(DEV)
    y2 = torch.empty(B, Lo, C, device=DEV)
    idx = torch.empty(B, Lo, C, dtype=torch.uint8, device=DEV)
    dx2 = torch.empty(B, T * H * W, C, device=DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_fwd_idx(_hip.ptr(xd), _hip.ptr(y2), _hip.ptr(idx), B, T, H, W, C, _st()))
    _hip.check(hip_lib.mvit_maxpool_skip_bwd_idx(_hip.ptr(idx), _hip.ptr(dyd), _hip.ptr(dx2), B, T, H, W, C, _st()))
    _close(dx2, x.grad, 1e-6)


@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 2, 16, 16, 96, 192), (1, 3, 7, 7, 192, 384), (1, 2, 14, 14, 384, 768), (1, 1, 5, 9, 96, 96),
                                              (1, 2, 28, 28, 192, 384)])
def test_proj_maxpool_fused_skip_path_bwd(hip_lib, B, T, H, W, Cin, Cout):
    """Un-pool + data-gradient GEMM in one kernel: dx bit-identical to mvit_maxpool_skip_bwd_idx followed by mvit_linear_fwd, the
    16-bit copy of the un-pooled gradient equal to the rounded fp32 one, and dx equal to autograd's on the same rounded operands.
    Integer-valued x makes ties (several windows choosing one token, first-maximum rule) frequent."""
    g = torch.Generator().manual_seed(50)
    x = torch.randint(-2, 3, (B, T * H * W, Cout), generator=g).float()          # the widened tensor the forward pooled
    dy_shape_probe = F.max_pool3d(x.reshape(B, T, H, W, Cout).permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    Lo = dy_shape_probe.shape[2] * dy_shape_probe.shape[3] * dy_shape_probe.shape[4]
    dy = _rnd(B, Lo, Cout, seed=51)
    w = _rnd(Cout, Cin, seed=52, scale=Cout ** -0.5)
    xd, dyd = x.to(DEV), dy.to(DEV)
    wt = w.t().contiguous().to(DEV).to(torch.bfloat16)                             # [Cin][Cout]
    M = B * T * H * W
    y = torch.empty(B, Lo, Cout, device=DEV)
    idx = torch.empty(B, Lo, Cout, dtype=torch.uint8, device=DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_fwd_idx(_hip.ptr(xd), _hip.ptr(y), _hip.ptr(idx), B, T, H, W, Cout, _st()))
    grf = torch.empty(M, Cout, device=DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_bwd_idx(_hip.ptr(idx), _hip.ptr(dyd), _hip.ptr(grf), B, T, H, W, Cout, _st()))
    dx0 = torch.empty(M, Cin, device=DEV)
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(grf), _hip.F32, Cout, _hip.ptr(wt), None, None, Cin, None, 0, _hip.ptr(dx0), _hip.F32, Cin, M,
                                       Cin, Cout, 0, _hip.BF16, _st()))
    dx1 = torch.full((M, Cin), float("nan"), device=DEV)
    d16 = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_proj_maxpool_bwd(_hip.ptr(idx), _hip.ptr(dyd), _hip.ptr(wt), _hip.ptr(dx1), _hip.ptr(d16), B, T, H, W, Cin, Cout,
                                             _hip.BF16, _st()))
    assert torch.equal(dx1, dx0)
    assert torch.equal(d16, grf.to(torch.bfloat16))
    dx2 = torch.full((M, Cin), float("nan"), device=DEV)                          # without the 16-bit copy
    _hip.check(hip_lib.mvit_proj_maxpool_bwd(_hip.ptr(idx), _hip.ptr(dyd), _hip.ptr(wt), _hip.ptr(dx2), None, B, T, H, W, Cin, Cout, _hip.BF16,
                                             _st()))
    assert torch.equal(dx2, dx0)
    xr = x.clone().requires_grad_(True)
    F.max_pool3d(xr.reshape(B, T, H, W, Cout).permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1)).reshape(B, Cout, Lo).transpose(
        1, 2).backward(dy)
    ref = xr.grad.reshape(M, Cout).to(torch.bfloat16).float() @ w.to(torch.bfloat16).float()
    _close(dx1, ref, 1e-5)


@pytest.mark.parametrize("B,T,S", [(2, 4, 64), (1, 4, 56), (1, 2, 448), (2, 16, 224)])      # the last: runs of 3-4 output rows per workgroup, across frames (row ring)
def test_stem_bwd(hip_lib, B, T, S):
    clip = _rnd(B, 3, T, S, S, seed=25)
    w = _rnd(96, 3, 3, 7, 7, seed=26, scale=0.05).requires_grad_(True)
    To, So = T // 2, S // 4
    ps = _rnd(1, So * So, 96, seed=27, scale=0.02).requires_grad_(True)
    pt = _rnd(1, To, 96, seed=28, scale=0.02).requires_grad_(True)
    x = F.conv3d(clip, w, None, stride=(2, 4, 4), padding=(1, 3, 3)).flatten(2).transpose(1, 2)
    x = x + (ps.repeat(1, To, 1) + torch.repeat_interleave(pt, So * So, dim=1))
    dx = _rnd(B, To * So * So, 96, seed=29)
    x.backward(dx)
    dW = torch.zeros(96, 441, device=DEV)
    dps, dpt = torch.zeros(So * So, 96, device=DEV), torch.zeros(To, 96, device=DEV)
    cd, dxd = clip.to(DEV), dx.to(DEV)
    nb = hip_lib.mvit_stem_bwd_workspace_bytes(B, T, S, _hip.F32)
    ws = torch.empty(nb // 4, device=DEV)
    _hip.check(hip_lib.mvit_stem_bwd(_hip.ptr(cd), _hip.ptr(dxd), _hip.ptr(dW), _hip.ptr(dps), _hip.ptr(dpt), B, T, S, _hip.F32, _hip.ptr(ws), nb, _st()))
    _close(dW, w.grad.reshape(96, 441), 3e-5)
    _close(dps, ps.grad[0], 2e-5)
    _close(dpt, pt.grad[0], 2e-5)
    # matrix-core weight gradient: same sums over 16-bit roundings of the clip and of the token gradients
    w2 = w.detach().clone().requires_grad_(True)
    x2 = F.conv3d(clip.to(torch.bfloat16).float(), w2, None, stride=(2, 4, 4), padding=(1, 3, 3)).flatten(2).transpose(1, 2)
    x2.backward(dx.to(torch.bfloat16).float())
    dW2 = torch.zeros(96, 441, device=DEV)
    dps2, dpt2 = torch.zeros(So * So, 96, device=DEV), torch.zeros(To, 96, device=DEV)
    nb2 = hip_lib.mvit_stem_bwd_workspace_bytes(B, T, S, _hip.BF16)
    ws2 = torch.empty(nb2 // 4, device=DEV)
    _hip.check(hip_lib.mvit_stem_bwd(_hip.ptr(cd), _hip.ptr(dxd), _hip.ptr(dW2), _hip.ptr(dps2), _hip.ptr(dpt2), B, T, S, _hip.BF16, _hip.ptr(ws2), nb2, _st()))
    _close(dW2, w2.grad.reshape(96, 441), 3e-5)
    _close(dps2, ps.grad[0], 2e-5)


def test_head_train_and_bwd(hip_lib):
    B, N, C = 3, 50, 384
    x = (_rnd(B, N, C, seed=30) + 0.3).requires_grad_(True)
    g = (1 + 0.1 * _rnd(C, seed=31)).requires_grad_(True)
    b = (0.1 * _rnd(C, seed=32)).requires_grad_(True)
    w = _rnd(18, C, seed=33, scale=0.05).requires_grad_(True)
    hb = _rnd(18, seed=34, scale=0.1).requires_grad_(True)
    mask = (torch.rand(B, C, generator=torch.Generator().manual_seed(35)) > 0.5).float() * 2.0
    labels = torch.zeros(B, 18)
    labels[0, 3], labels[1, 5], labels[2, 7], labels[2, 8] = 1.0, 1.0, 0.7, 0.3
    z = F.layer_norm(x, (C,), g, b, 1e-6).mean(1) * mask
    logits = F.linear(z, w, hb)
    loss = O.soft_target_cross_entropy(logits, labels)
    loss.backward()
    nch = (N + 31) // 32
    ws = torch.empty(B * nch * C, device=DEV)
    xd, gd, bd, wd, hbd, md, ld = (t.detach().to(DEV) for t in (x, g, b, w, hb, mask, labels))
    zo = torch.empty(B, C, device=DEV)
    lg = torch.empty(B, 18, device=DEV)
    _hip.check(hip_lib.mvit_head_ln_partial(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(ws), B, N, C, 1e-6, _st()))
    _hip.check(hip_lib.mvit_head_project_train(_hip.ptr(ws), _hip.ptr(wd), _hip.ptr(hbd), _hip.ptr(md), _hip.ptr(zo), _hip.ptr(lg),
                                               B, N, nch, C, 18, _st()))
    _close(lg, logits.detach(), 1e-5)
    lossd = torch.empty(1, device=DEV)
    dl = torch.empty(B, 18, device=DEV)
    _hip.check(hip_lib.mvit_soft_ce(_hip.ptr(lg), _hip.ptr(ld), _hip.ptr(lossd), _hip.ptr(dl), B, 18, 1.0, _st()))
    assert abs(lossd.item() - loss.item()) < 1e-5
    dW, dbh, dz = torch.empty(18, C, device=DEV), torch.empty(18, device=DEV), torch.empty(B, C, device=DEV)
    _hip.check(hip_lib.mvit_head_bwd(_hip.ptr(dl), _hip.ptr(zo), _hip.ptr(wd), _hip.ptr(md), _hip.ptr(dW), _hip.ptr(dbh),
                                     _hip.ptr(dz), B, C, 18, 0, _st()))
    _close(dW, w.grad, 1e-5)
    _close(dbh, hb.grad, 1e-5)
    dx = torch.empty(B * N, C, device=DEV)
    dg, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ws2 = torch.empty(hip_lib.mvit_layernorm_bwd_workspace_bytes(C) // 4, device=DEV)
    _hip.check(hip_lib.mvit_layernorm_bwd(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(dz), _hip.F32, N, 1.0 / N, None, _hip.ptr(dx),
                                          _hip.ptr(dg), _hip.ptr(dbeta), 0, _hip.ptr(ws2), B * N, C, 1e-6, None, None, 0, _st()))
    _close(dx.view(B, N, C), x.grad, 1e-5)
    _close(dg, g.grad, 1e-5)
    _close(dbeta, b.grad, 1e-5)


def test_grad_norm_and_adamw_match_torch(hip_lib):
    torch.manual_seed(0)
    shapes = [(96, 3, 3, 7, 7), (288, 96), (288,), (96,), (5000,), (384, 96)]
    wds = [1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-4]
    ps = [torch.randn(s) for s in shapes]
    gs = [torch.randn(s) * 0.5 for s in shapes]
    ref_p = [p.clone().requires_grad_(True) for p in ps]
    for p, g in zip(ref_p, gs):
        p.grad = g.clone()
    opt = torch.optim.AdamW([{"params": [p for p, w in zip(ref_p, wds) if w > 0], "weight_decay": 1e-4},
                             {"params": [p for p, w in zip(ref_p, wds) if w == 0], "weight_decay": 0.0}], lr=1e-3,
                            betas=(0.9, 0.999), eps=1e-8)
    norm = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
    dp = [p.clone().to(DEV) for p in ps]
    dg = [g.clone().to(DEV) for g in gs]
    dm = [torch.zeros_like(p) for p in dp]
    dv = [torch.zeros_like(p) for p in dp]
    CH = 2048
    rec = []
    for p, g, m, v, wd in zip(dp, dg, dm, dv, wds):
        n = p.numel()
        for off in range(0, n, CH):
            rec.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off,
                        min(CH, n - off), wd))
    assert hip_lib.mvit_mt_chunk_bytes() == 40
    dt = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i4"), ("wd", "f4")])
    table = torch.from_numpy(np.array(rec, dtype=dt).view(np.uint8).copy()).to(DEV)
    partials = torch.empty(len(rec), device=DEV)
    out2 = torch.empty(2, device=DEV)
    for step in (1, 2):
        if step == 2:
            for p, g in zip(ref_p, gs):
                p.grad = g.clone()
            norm = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        opt.step()
        _hip.check(hip_lib.mvit_grad_norm(_hip.ptr(table), len(rec), 1.0, _hip.ptr(partials), _hip.ptr(out2), _st()))
        assert abs(out2[0].item() - norm.item()) <= 1e-4 * norm.item()
        _hip.check(hip_lib.mvit_adamw_step(_hip.ptr(table), len(rec), _hip.ptr(out2), 1e-3, 0.9, 0.999, 1e-8, step, _st()))
        for p, r in zip(dp, ref_p):
            _close(p, r.detach(), 2e-6)


@pytest.mark.parametrize("R,C", [(96, 96), (288, 96), (100, 70), (1152, 384)])
def test_cast_transpose(hip_lib, R, C):
    x = _rnd(R, C, seed=41)
    xd = x.to(DEV)
    d = torch.empty(R, C, dtype=torch.bfloat16, device=DEV)
    dt = torch.empty(C, R, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_cast_transpose_f32_to_bf16(_hip.ptr(xd), _hip.ptr(d), _hip.ptr(dt), R, C, _st()))
    ref = x.to(torch.bfloat16)
    assert torch.equal(d.cpu(), ref) and torch.equal(dt.cpu(), ref.t().contiguous())
    dt2 = torch.zeros_like(dt)
    _hip.check(hip_lib.mvit_cast_transpose_f32_to_bf16(_hip.ptr(xd), None, _hip.ptr(dt2), R, C, _st()))
    assert torch.equal(dt2.cpu(), ref.t().contiguous())


# ---- round-2 entry points: the deterministic / fused forms must agree with the forms they replace -------------------------------

@pytest.mark.parametrize("C,rows,rps", [(96, 1000, 250), (384, 392 * 4, 392), (768, 130, 0)])
def test_layernorm_bwd_emits_the_cast_of_its_own_result(hip_lib, C, rows, rps):
    """mvit_layernorm_bwd's 16-bit side output is bit for bit mvit_cast_rows_f32_to_bf16 of the dx it wrote, and dx / d_gamma /
    d_beta are unchanged by asking for it."""
    x = (_rnd(rows, C, seed=61) * 2 + 0.3).to(DEV)
    g = (1 + 0.1 * _rnd(C, seed=62)).to(DEV)
    dy = _rnd(rows, C, seed=63).to(torch.bfloat16).to(DEV)
    base = _rnd(rows, C, seed=64).to(DEV)
    sc = (torch.rand((rows + rps - 1) // rps, generator=torch.Generator().manual_seed(7)) * 2).to(DEV) if rps else None
    nws = hip_lib.mvit_layernorm_bwd_workspace_bytes(C) // 4
    outs = []
    for emit in (False, True):
        dx = torch.empty(rows, C, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx16 = torch.zeros(rows, C, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(nws, device=DEV)
        _hip.check(hip_lib.mvit_layernorm_bwd(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dy), _hip.BF16, 1, 1.0, _hip.ptr(base), _hip.ptr(dx),
                                               _hip.ptr(dg), _hip.ptr(db), 0, _hip.ptr(ws), rows, C, 1e-6,
                                               _hip.ptr(dx16) if emit else None, _hip.ptr(sc) if emit else None, rps if emit and rps else 0, _st()))
        outs.append((dx, dg, db, dx16))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    ref16 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_cast_rows_f32_to_bf16(_hip.ptr(outs[1][0]), _hip.ptr(ref16), rows, C, _hip.ptr(sc), rps, _st()))
    assert torch.equal(outs[1][3], ref16)


def test_reduce_queue_gives_the_bits_of_the_immediate_reductions(hip_lib):
    """Queued (one launch at the flush) and immediate parameter-gradient reductions of LayerNorm backwards: identical sums; more
    entries than the queue holds flush themselves."""
    torch.manual_seed(3)
    cases = [(384, 3000), (96, 5000), (768, 700), (192, 64)] * 5          # 20 > the queue's 16 slots
    xs = [(torch.randn(r, C, device=DEV) * 2, 1 + 0.1 * torch.randn(C, device=DEV), torch.randn(r, C, device=DEV)) for C, r in cases]

    def run(queued):
        res, keep = [], []
        if queued:
            _hip.check(hip_lib.mvit_reduce_queue_begin())
        for (C, r), (x, g, dy) in zip(cases, xs):
            dx = torch.empty_like(x)
            dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            ws = torch.empty(hip_lib.mvit_layernorm_bwd_workspace_bytes(C) // 4, device=DEV)
            keep.append(ws)                      # the contract: workspaces stay untouched until the flush
            _hip.check(hip_lib.mvit_layernorm_bwd(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dy), _hip.F32, 1, 1.0, None, _hip.ptr(dx), _hip.ptr(dg),
                                                  _hip.ptr(db), 0, _hip.ptr(ws), r, C, 1e-6, None, None, 0, _st()))
            res.append((dg, db))
        if queued:
            _hip.check(hip_lib.mvit_reduce_queue_flush(_st()))
        torch.cuda.synchronize()
        return res
    imm, que = run(False), run(True)
    for (a0, a1), (b0, b1) in zip(imm, que):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
    ref = (xs[0][2] * 1.0).sum(0)               # d_beta = column sums of dy
    _close(imm[0][1], ref.cpu(), 2e-5)


@pytest.mark.parametrize("M,N,K", [(6336, 384, 192), (4096, 1152, 384), (64 * 900, 192, 96), (3000, 96, 96)])
def test_linear_wgrad_is_reproducible_and_matches(hip_lib, M, N, K):
    """The weight gradient (per-chunk slabs added in chunk order): equal to the reference product and bit-identical between two runs."""
    a = _rnd(M, K, seed=71).to(torch.bfloat16).to(DEV)
    dy = _rnd(M, N, seed=72).to(torch.bfloat16).to(DEV)
    ref = (dy.float().t() @ a.float()).cpu()
    nb = hip_lib.mvit_linear_wgrad_workspace_bytes(_hip.BF16, K, _hip.BF16, N, 0, M, N, K, _hip.BF16)
    assert nb > 0
    outs = []
    for _ in range(2):
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ws = torch.empty(nb // 4, device=DEV)
        _hip.check(hip_lib.mvit_linear_wgrad(_hip.ptr(a), _hip.BF16, K, _hip.ptr(dy), _hip.BF16, N, None, 0, _hip.ptr(dW), _hip.ptr(db), M, N, K,
                                              _hip.BF16, _hip.ptr(ws), nb, _st()))
        outs.append((dW, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    _close(outs[0][0], ref, 2e-3)
    _close(outs[0][1], dy.float().sum(0).cpu(), 2e-3)


def test_adamw_step_dev_equals_adamw_step(hip_lib):
    """The captured-step form (lr and bias corrections read from device memory) gives the bits of the host-scalar form, also on
    gradients that are not 16-byte aligned (views into a DistributedDataParallel bucket)."""
    import math
    torch.manual_seed(5)
    n = 70001
    flat = torch.randn(4 * n + 8, device=DEV)
    dt = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i4"), ("wd", "f4")])
    res = []
    for variant in ("host", "dev", "host-unaligned"):
        p = flat[:n].clone()
        gsrc = torch.randn(n + 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
        g = gsrc[1:n + 1] if variant == "host-unaligned" else gsrc[1:n + 1].clone()       # the view starts 4 bytes off a 16-byte boundary
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        CH = 65536
        rec = [(p.data_ptr() + 4 * o, g.data_ptr() + 4 * o, m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, min(CH, n - o), 1e-4) for o in range(0, n, CH)]
        table = torch.from_numpy(np.array(rec, dtype=dt).view(np.uint8).copy()).to(DEV)
        partials, out2 = torch.empty(len(rec), device=DEV), torch.empty(2, device=DEV)
        for step in (1, 2, 3):
            _hip.check(hip_lib.mvit_grad_norm(_hip.ptr(table), len(rec), 1.0, _hip.ptr(partials), _hip.ptr(out2), _st()))
            if variant == "dev":
                f32 = lambda x: ctypes.c_float(x).value
                libm = ctypes.CDLL("libm.so.6")
                libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
                libm.sqrtf.restype = ctypes.c_float; libm.sqrtf.argtypes = [ctypes.c_float]
                hyper = torch.tensor([1e-3, f32(1.0 - libm.powf(0.9, float(step))), libm.sqrtf(f32(1.0 - libm.powf(0.999, float(step))))],
                                     dtype=torch.float32, device=DEV)
                _hip.check(hip_lib.mvit_adamw_step_dev(_hip.ptr(table), len(rec), _hip.ptr(out2), _hip.ptr(hyper), 0.9, 0.999, 1e-8, _st()))
            else:
                _hip.check(hip_lib.mvit_adamw_step(_hip.ptr(table), len(rec), _hip.ptr(out2), 1e-3, 0.9, 0.999, 1e-8, step, _st()))
        res.append((p.clone(), m.clone(), v.clone(), out2.clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,h,T,H,W", [(2, 2, 3, 14, 14), (1, 4, 4, 28, 28), (3, 1, 2, 10, 18)])
def test_pool_kv_pair_form_equals_two_single_calls(hip_lib, act, B, h, T, H, W):
    """mvit_pool_conv_ln_fwd_train_kv / _bwd_saved_kv (k and v of a block in one set of launches, stride 2): every output is bit for
    bit what the single-tensor entry points give."""
    s = 2
    C = 96 * h
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    L = T * Ho * Wo
    dt = torch.float32 if act == _hip.F32 else torch.bfloat16
    qkv = _rnd(B, T * H * W, 3 * C, seed=81).to(dt).to(DEV)
    par = [((_rnd(96, 27, seed=82 + i) * 0.2).to(DEV), (1 + 0.1 * _rnd(96, seed=84 + i)).to(DEV), (0.1 * _rnd(96, seed=86 + i)).to(DEV)) for i in range(2)]
    # ---- forward
    kv = torch.empty(2, B, h, L, 96, dtype=dt, device=DEV)
    xh = torch.empty_like(kv)
    rs = torch.empty(2, B * h * L, device=DEV)
    _hip.check(hip_lib.mvit_pool_conv_ln_fwd_train_kv(_hip.ptr(qkv), 3 * C, C, _hip.ptr(par[0][0]), _hip.ptr(par[0][1]), _hip.ptr(par[0][2]),
                                                      _hip.ptr(par[1][0]), _hip.ptr(par[1][1]), _hip.ptr(par[1][2]), _hip.ptr(kv), _hip.ptr(xh),
                                                      _hip.ptr(rs), B, h, T, H, W, s, 1e-5, act, _st()))
    for i in range(2):
        o1 = torch.empty(B, h, L, 96, dtype=dt, device=DEV)
        x1 = torch.empty_like(o1)
        r1 = torch.empty(B * h * L, device=DEV)
        _hip.check(hip_lib.mvit_pool_conv_ln_fwd_train(_hip.ptr(qkv), 3 * C, (1 + i) * C, _hip.ptr(par[i][0]), _hip.ptr(par[i][1]), _hip.ptr(par[i][2]),
                                                       _hip.ptr(o1), _hip.ptr(x1), _hip.ptr(r1), B, h, T, H, W, s, 1e-5, act, _st()))
        assert torch.equal(kv[i], o1) and torch.equal(xh[i], x1) and torch.equal(rs[i], r1)
    # ---- backward
    dkv = _rnd(2, B, h, L, 96, seed=90).to(dt).to(DEV)
    nb = hip_lib.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, s)
    dconv = torch.empty_like(dkv)
    dqkv = torch.zeros_like(qkv)
    g = [[torch.zeros(96, 27, device=DEV), torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)] for _ in range(2)]
    ws = torch.empty(2 * nb // 4, device=DEV)
    _hip.check(hip_lib.mvit_pool_conv_ln_bwd_saved_kv(_hip.ptr(qkv), 3 * C, C, _hip.ptr(par[0][0]), _hip.ptr(par[0][1]), _hip.ptr(par[1][0]),
                                                      _hip.ptr(par[1][1]), _hip.ptr(xh), _hip.ptr(rs), _hip.ptr(dkv), _hip.ptr(dconv), _hip.ptr(dqkv),
                                                      _hip.ptr(g[0][0]), _hip.ptr(g[0][1]), _hip.ptr(g[0][2]), _hip.ptr(g[1][0]), _hip.ptr(g[1][1]),
                                                      _hip.ptr(g[1][2]), 1, _hip.ptr(ws), B, h, T, H, W, s, act, _st()))
    dqkv1 = torch.zeros_like(qkv)
    for i in range(2):
        dc1 = torch.empty(B, h, L, 96, dtype=dt, device=DEV)
        g1 = [torch.zeros(96, 27, device=DEV), torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)]
        ws1 = torch.empty(nb // 4, device=DEV)
        _hip.check(hip_lib.mvit_pool_conv_ln_bwd_saved(_hip.ptr(qkv), 3 * C, (1 + i) * C, _hip.ptr(par[i][0]), _hip.ptr(par[i][1]), _hip.ptr(xh[i]),
                                                       _hip.ptr(rs[i]), _hip.ptr(dkv[i]), _hip.ptr(dc1), _hip.ptr(dqkv1), _hip.ptr(g1[0]), _hip.ptr(g1[1]),
                                                       _hip.ptr(g1[2]), 1, _hip.ptr(ws1), B, h, T, H, W, s, 1e-5, act, _st()))
        assert torch.equal(dconv[i], dc1)
        for a, b in zip(g[i], g1):
            assert torch.equal(a, b)
    assert torch.equal(dqkv, dqkv1)
    # other strides are refused (the caller falls back to the single form)
    assert hip_lib.mvit_pool_conv_ln_fwd_train_kv(_hip.ptr(qkv), 3 * C, C, _hip.ptr(par[0][0]), _hip.ptr(par[0][1]), _hip.ptr(par[0][2]),
                                                  _hip.ptr(par[1][0]), _hip.ptr(par[1][1]), _hip.ptr(par[1][2]), _hip.ptr(kv), _hip.ptr(xh),
                                                  _hip.ptr(rs), B, h, T, H, W, 1, 1e-5, act, _st()) != 0
