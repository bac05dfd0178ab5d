"""GPU: every C-ABI operator against the oracle's op (same seeded inputs), fp32 (exact path) and bf16 (MFMA path).

fp32 tolerance: 1e-5 relative-to-scale (summation order differs from ATen).  bf16 tolerance: inputs are rounded to
bf16 first so the reference sees the same operands; remaining error is bf16 rounding of outputs / P (2^-8 rel).
"""
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


def _tol(act, fp32=2e-5, bf16=2e-2):
    return bf16 if act == _hip.BF16 else fp32


def _close(got, ref, tol):
    got = got.float().cpu()
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "max err %.3e > %.3e (scale %.3f)" % (err, tol * scale, scale)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("C,rows", [(96, 1000), (192, 37), (384, 392), (768, 65)])
def test_layernorm(hip_lib, act, C, rows):
    x = _rnd(rows, C, seed=1) * 2 + 0.5
    g, b = 1 + 0.1 * _rnd(C, seed=2), 0.1 * _rnd(C, seed=3)
    ref = F.layer_norm(x, (C,), g, b, 1e-6)
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    y = torch.empty(rows, C, dtype=torch.bfloat16 if act else torch.float32, device=DEV)
    _hip.check(hip_lib.mvit_layernorm_fwd(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(y), rows, C, 1e-6, act, _st()))
    _close(y, ref, _tol(act, bf16=8e-3))


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("M,N,K,epi", [(392, 288, 96, "b"), (1000, 96, 384, "br"), (129, 384, 96, "bg"),
                                        (64, 192, 768, "b"), (257, 96, 96, ""), (512, 576, 192, "brg"),
                                        (300, 384, 160, "b"), (1000, 192, 96, "br"), (128 * 70 + 9, 384, 96, "b"),   # K = 64 j + 32 on the 128x192 kernels
                                        (32768 + 40, 192, 1024, "b")])                                               # long K, many rows: the 256-row / 8-wave tile
def test_linear(hip_lib, act, M, N, K, epi):
    a = _act(_rnd(M, K, seed=4), act)
    w = _act(_rnd(N, K, seed=5, scale=0.05), act)
    bias = _rnd(N, seed=6, scale=0.1)
    res = _rnd(M, N, seed=7)
    ref = a.float() @ w.float().t()
    flags = 0
    if "b" in epi:
        ref = ref + bias
        flags |= _hip.EPI_BIAS
    if "g" in epi:
        ref = F.gelu(ref)
        flags |= _hip.EPI_GELU
    if "r" in epi:
        ref = ref + res
        flags |= _hip.EPI_RESIDUAL
    for out_bf16 in ([False, True] if act == _hip.BF16 else [False]):
        y = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
        ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
        _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(ad), act, K, _hip.ptr(wd), _hip.ptr(bd) if "b" in epi else None,
                                           _hip.ptr(rd) if "r" in epi else None, N, None, 0, _hip.ptr(y),
                                           _hip.BF16 if out_bf16 else _hip.F32, N, M, N, K, flags, act, _st()))
        _close(y, ref, 1e-2 if out_bf16 else (2e-3 if act else 2e-5))


def test_linear_bf16_with_fp32_a_and_row_scale(hip_lib):
    M, N, K = 300, 192, 96
    a = _rnd(M, K, seed=8)
    w = _rnd(N, K, seed=9, scale=0.05).to(torch.bfloat16)
    bias = _rnd(N, seed=10, scale=0.1)
    res = _rnd(M, N, seed=11)
    scale = torch.tensor([0.0, 1.25, 2.0])
    ref = (a.to(torch.bfloat16).float() @ w.float().t() + bias) * scale.repeat_interleave(100)[:, None] + res
    y = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ad, wd, bd, rd, sd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), scale.to(DEV)
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(ad), _hip.F32, K, _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(rd), N,
                                       _hip.ptr(sd), 100, _hip.ptr(y), _hip.F32, N, M, N, K,
                                       _hip.EPI_BIAS | _hip.EPI_RESIDUAL, _hip.BF16, _st()))
    _close(y, ref, 2e-3)


@pytest.mark.parametrize("epi", ["b", "bg", "br", "brs"])
def test_linear_persistent_many_tiles(hip_lib, epi):
    """More tiles than resident workgroups (persistent 128x192 kernel: cross-tile prefetch, trailing stores), ragged M."""
    M, N, K = 128 * 300 + 40, 576, 192
    a = _rnd(M, K, seed=21).to(torch.bfloat16)
    w = _rnd(N, K, seed=22, scale=0.05).to(torch.bfloat16)
    bias = _rnd(N, seed=23, scale=0.1)
    res = _rnd(M, N, seed=24)
    rps = 1000
    scale = torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(3)) * 2
    ad, wd, bd, rd, sd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), scale.to(DEV)
    ref = ad.float() @ wd.float().t() + bd
    flags = _hip.EPI_BIAS
    if "g" in epi:
        ref = F.gelu(ref)
        flags |= _hip.EPI_GELU
    if "s" in epi:
        ref = ref * sd.repeat_interleave(rps)[:M, None]
    if "r" in epi:
        ref = ref + rd
        flags |= _hip.EPI_RESIDUAL
    for out_bf16 in ([False] if "r" in epi else [False, True]):
        y = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
        _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(ad), _hip.BF16, K, _hip.ptr(wd), _hip.ptr(bd),
                                           _hip.ptr(rd) if "r" in epi else None, N, _hip.ptr(sd) if "s" in epi else None, rps,
                                           _hip.ptr(y), _hip.BF16 if out_bf16 else _hip.F32, N, M, N, K, flags, _hip.BF16, _st()))
        _close(y, ref.cpu(), 1e-2 if out_bf16 else 2e-3)


@pytest.mark.parametrize("M,N,K", [(128 * 150 + 40, 768, 192), (700, 384, 96), (4096, 1536, 384), (128 * 150 + 40, 384, 96)])
def test_linear_gelu_dual_output(hip_lib, M, N, K):
    """fc1 of a training step: one GEMM pass writes the pre-activation and GELU(pre) (persistent kernel; K=96 takes the
    plain GEMM + element-wise route).  pre must equal the bias-only GEMM bit for bit; y = GELU of the fp32 accumulator."""
    a = _rnd(M, K, seed=31).to(torch.bfloat16).to(DEV)
    w = _rnd(N, K, seed=32, scale=0.08).to(torch.bfloat16).to(DEV)
    bias = _rnd(N, seed=33, scale=0.2).to(DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    y = torch.empty_like(pre)
    _hip.check(hip_lib.mvit_linear_gelu_fwd(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(pre), _hip.ptr(y), M, N, K,
                                            _hip.BF16, _st()))
    plain = torch.empty_like(pre)
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(a), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias), None, N, None, 0, _hip.ptr(plain),
                                       _hip.BF16, N, M, N, K, _hip.EPI_BIAS, _hip.BF16, _st()))
    assert torch.equal(pre, plain)
    ref = a.float() @ w.float().t() + bias
    _close(pre, ref.cpu(), 1e-2)
    _close(y, F.gelu(ref).cpu(), 1e-2)


@pytest.mark.parametrize("M,N,K", [(128 * 40 + 40, 768, 192), (700, 384, 96), (2048, 1536, 384)])
def test_linear_dgelu(hip_lib, M, N, K):
    """fc2 data gradient fused with the GELU backward: y = GELU'(pre) * scale[row] * (a w^T) (K=96 takes the unfused route)."""
    a = _rnd(M, K, seed=41).to(torch.bfloat16).to(DEV)
    w = _rnd(N, K, seed=42, scale=0.08).to(torch.bfloat16).to(DEV)
    pre = (_rnd(M, N, seed=43) * 1.5).to(torch.bfloat16).to(DEV)
    rps = 300
    sc = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(5)) * 2).to(DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_linear_dgelu_fwd(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(sc), rps, _hip.ptr(pre), _hip.ptr(y), M, N, K,
                                             _hip.BF16, _st()))
    x = pre.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    ref = x.grad * (a.float() @ w.float().t()) * sc.repeat_interleave(rps)[:M, None]
    _close(y, ref.cpu(), 1.5e-2)
    y2 = torch.empty_like(y)                 # no drop-path scale
    _hip.check(hip_lib.mvit_linear_dgelu_fwd(_hip.ptr(a), K, _hip.ptr(w), None, 0, _hip.ptr(pre), _hip.ptr(y2), M, N, K,
                                             _hip.BF16, _st()))
    _close(y2, (x.grad * (a.float() @ w.float().t())).cpu(), 1.5e-2)


@pytest.mark.parametrize("N,K,K2", [(768, 192, 192), (384, 96, 96)])
def test_linear_gelu_derivative_pair(hip_lib, N, K, K2):
    """fc1 keeps GELU'(pre) instead of pre; the fc2 data gradient multiplies by it: same result as the pre-activation pair.
    (384, 96, 96) is block 0's MLP: K = 96 runs as one full 64-wide slab plus a half slab."""
    M = 128 * 30 + 40
    a = _rnd(M, K, seed=51).to(torch.bfloat16).to(DEV)
    w = _rnd(N, K, seed=52, scale=0.08).to(torch.bfloat16).to(DEV)
    bias = _rnd(N, seed=53, scale=0.2).to(DEV)
    dact = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    y = torch.empty_like(dact)
    _hip.check(hip_lib.mvit_linear_gelu_fwd_dsave(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(dact), _hip.ptr(y), M, N, K,
                                                  _hip.BF16, _st()))
    x = (a.float() @ w.float().t() + bias).requires_grad_(True)
    g = F.gelu(x)
    g.sum().backward()
    _close(y, g.detach().cpu(), 1e-2)
    _close(dact, x.grad.cpu(), 1e-2)
    # backward side: a2 [M, K2] times w2t [N, K2] (= fc2.weight^T), scaled rows, times the saved derivative
    a2 = _rnd(M, K2, seed=54).to(torch.bfloat16).to(DEV)
    w2t = _rnd(N, K2, seed=55, scale=0.08).to(torch.bfloat16).to(DEV)
    rps = 500
    sc = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(6)) * 2).to(DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_linear_dact_fwd(_hip.ptr(a2), K2, _hip.ptr(w2t), _hip.ptr(sc), rps, _hip.ptr(dact), _hip.ptr(out), M, N, K2,
                                            _hip.BF16, _st()))
    ref = dact.float() * (a2.float() @ w2t.float().t()) * sc.repeat_interleave(rps)[:M, None]
    _close(out, ref.cpu(), 1e-2)
    assert hip_lib.mvit_linear_gelu_fwd_dsave(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(dact), _hip.ptr(y), M, 96, K,
                                              _hip.BF16, _st()) != 0     # N = 96: not a 128x192 shape


@pytest.mark.parametrize("M,N,K", [(8192 + 40, 192, 128), (256 * 70 + 13, 576, 192), (256 * 33, 384, 1536), (8192 + 200, 1536, 384)])
def test_linear_pingpong_all_epilogues(hip_lib, M, N, K):
    """The 256 x 192 ping-pong kernel (linear_pp.hip; M >= 8192 routes to it): every epilogue it implements, ragged M, more tiles
    than workgroups, against fp32 torch on the same 16-bit operands."""
    a = _rnd(M, K, seed=61).to(torch.bfloat16).to(DEV)
    w = _rnd(N, K, seed=62, scale=0.06).to(torch.bfloat16).to(DEV)
    bias = _rnd(N, seed=63, scale=0.2).to(DEV)
    res = _rnd(M, N, seed=64).to(DEV)
    rps = 1000
    sc = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(7)) * 2).to(DEV)
    scr = sc.repeat_interleave(rps)[:M, None]
    acc = a.float() @ w.float().t()

    def lin(flags, out_dt, residual=None, scale=None, with_bias=True):
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16 if out_dt == _hip.BF16 else torch.float32, device=DEV)
        _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(a), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias) if with_bias else None, _hip.ptr(residual), N,
                                           _hip.ptr(scale), rps, _hip.ptr(y), out_dt, N, M, N, K, flags, _hip.BF16, _st()))
        return y
    _close(lin(_hip.EPI_BIAS, _hip.BF16), (acc + bias).cpu(), 1e-2)
    _close(lin(0, _hip.BF16, with_bias=False), acc.cpu(), 1e-2)
    _close(lin(_hip.EPI_BIAS | _hip.EPI_GELU, _hip.BF16), F.gelu(acc + bias).cpu(), 1e-2)      # (routed to the 128 x 192 kernel by default, to the ping-pong one under MVIT_GEMM_PP=1)
    _close(lin(_hip.EPI_BIAS, _hip.F32), (acc + bias).cpu(), 2e-3)
    _close(lin(_hip.EPI_BIAS | _hip.EPI_RESIDUAL, _hip.F32, residual=res), (acc + bias + res).cpu(), 2e-3)
    _close(lin(_hip.EPI_BIAS | _hip.EPI_RESIDUAL, _hip.F32, residual=res, scale=sc), ((acc + bias) * scr + res).cpu(), 2e-3)
    # the MLP pairs of a training step
    pre, y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_linear_gelu_fwd(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(pre), _hip.ptr(y), M, N, K, _hip.BF16, _st()))
    assert torch.equal(pre, lin(_hip.EPI_BIAS, _hip.BF16))
    _close(y, F.gelu(acc + bias).cpu(), 1e-2)
    x = (acc + bias).clone().requires_grad_(True)
    g = F.gelu(x)
    g.sum().backward()
    dact = torch.empty_like(pre)
    _hip.check(hip_lib.mvit_linear_gelu_fwd_dsave(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(dact), _hip.ptr(y), M, N, K, _hip.BF16, _st()))
    _close(y, g.detach().cpu(), 1e-2)
    _close(dact, x.grad.cpu(), 1e-2)
    aux = (_rnd(M, N, seed=65) * 1.5).to(torch.bfloat16).to(DEV)
    xa = aux.float().requires_grad_(True)
    F.gelu(xa).sum().backward()
    out = torch.empty_like(pre)
    _hip.check(hip_lib.mvit_linear_dgelu_fwd(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(sc), rps, _hip.ptr(aux), _hip.ptr(out), M, N, K, _hip.BF16, _st()))
    _close(out, (xa.grad * acc * scr).cpu(), 1.5e-2)
    _hip.check(hip_lib.mvit_linear_dact_fwd(_hip.ptr(a), K, _hip.ptr(w), None, 0, _hip.ptr(aux), _hip.ptr(out), M, N, K, _hip.BF16, _st()))
    _close(out, (aux.float() * acc).cpu(), 1.5e-2)


@pytest.mark.parametrize("M,N", [(128 * 300 + 77, 288), (128 * 600, 384), (128 * 257 + 1, 576), (32768, 384)])
def test_linear_k96_resident_weights_all_epilogues(hip_lib, M, N):
    """csrc/linear_k96.hip (K = 96, N in {288, 384, 576}, M >= 32768 route to it: qkv of blocks 0 / 1 and fc1 of block 0,
    attention.py:231, common.py:27-31): weights resident in LDS, one persistent workgroup per CU streaming 128-row token tiles.  Bias,
    bias + GELU and the two training pairs (GELU + pre-activation, GELU + derivative); ragged M (rows past M never stored: the canary
    rows behind the output stay NaN), more tiles than workgroups, one- and two-pass column layouts; against fp32 torch on the same
    16-bit operands."""
    K = 96
    a = _rnd(M, K, seed=71).to(torch.bfloat16).to(DEV)
    w = _rnd(N, K, seed=72, scale=0.12).to(torch.bfloat16).to(DEV)
    bias = _rnd(N, seed=73, scale=0.3).to(DEV)
    acc = a.float() @ w.float().t() + bias

    def buf():
        return torch.full((M + 64, N), float("nan"), dtype=torch.bfloat16, device=DEV)      # 64 canary rows

    def check_canary(t):
        assert bool(torch.isnan(t[M:].float()).all()), "rows past M were written"
    y = buf()
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(a), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias), None, N, None, 0, _hip.ptr(y), _hip.BF16, N, M, N, K,
                                       _hip.EPI_BIAS, _hip.BF16, _st()))
    _close(y[:M], acc.cpu(), 1e-2)
    check_canary(y)
    yb = y[:M].clone()
    y = buf()
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(a), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias), None, N, None, 0, _hip.ptr(y), _hip.BF16, N, M, N, K,
                                       _hip.EPI_BIAS | _hip.EPI_GELU, _hip.BF16, _st()))
    _close(y[:M], F.gelu(acc).cpu(), 1e-2)
    check_canary(y)
    yg = y[:M].clone()
    pre, y = buf(), buf()
    _hip.check(hip_lib.mvit_linear_gelu_fwd(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(pre), _hip.ptr(y), M, N, K, _hip.BF16, _st()))
    assert torch.equal(pre[:M], yb) and torch.equal(y[:M], yg)
    check_canary(pre), check_canary(y)
    x = acc.clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    dact, y = buf(), buf()
    _hip.check(hip_lib.mvit_linear_gelu_fwd_dsave(_hip.ptr(a), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(dact), _hip.ptr(y), M, N, K, _hip.BF16, _st()))
    assert torch.equal(y[:M], yg)
    _close(dact[:M], x.grad.cpu(), 1e-2)
    check_canary(dact), check_canary(y)


def test_linear_rejects_bad_shapes(hip_lib):
    t = torch.zeros(64, 64, device=DEV)
    assert hip_lib.mvit_linear_fwd(_hip.ptr(t), _hip.BF16, 64, _hip.ptr(t), None, None, 0, None, 0, _hip.ptr(t), _hip.BF16,
                                   64, 64, 64, 64, 0, _hip.BF16, _st()) == -4
    assert hip_lib.mvit_linear_fwd(None, _hip.F32, 64, _hip.ptr(t), None, None, 0, None, 0, _hip.ptr(t), _hip.F32, 64, 64,
                                   64, 64, 0, _hip.F32, _st()) == -1
    assert hip_lib.mvit_layernorm_fwd(_hip.ptr(t), _hip.ptr(t), _hip.ptr(t), _hip.ptr(t), 64, 100, 1e-6, 0, _st()) == -4


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,h,T,H,W,s", [(2, 1, 2, 16, 16, 1), (1, 2, 2, 14, 14, 2), (2, 2, 3, 7, 7, 2),
                                         (1, 4, 2, 14, 14, 4), (1, 1, 8, 28, 28, 8), (1, 2, 2, 5, 9, 1),
                                         # stride-1 grids whose height is a multiple of 14: the full-lane march kernel (16-bit builds)
                                         (1, 2, 3, 14, 14, 1), (2, 1, 4, 28, 28, 1), (1, 1, 2, 28, 14, 1), (1, 1, 1, 14, 7, 1)])
def test_pool_conv_ln(hip_lib, act, B, h, T, H, W, s):
    C = 96 * h
    N = T * H * W
    qkv = _act(_rnd(B, N, 3 * C, seed=12), act)
    w = _rnd(96, 1, 3, 3, 3, seed=13, scale=0.3)
    g, b = 1 + 0.1 * _rnd(96, seed=14), 0.1 * _rnd(96, seed=15)
    for which in range(3):
        x = qkv.float()[:, :, which * C:(which + 1) * C].reshape(B, N, h, 96).permute(0, 2, 1, 3)
        ref, thw = O._pool_conv_ln(x, (T, H, W), w, (1, s, s), g, b)
        Lo = thw[0] * thw[1] * thw[2]
        out = torch.empty(B, h, Lo, 96, dtype=qkv.dtype, device=DEV)
        qd, wd, gd, bd = qkv.to(DEV), w.to(DEV), g.to(DEV), b.to(DEV)
        _hip.check(hip_lib.mvit_pool_conv_ln_fwd(_hip.ptr(qd), 3 * C, which * C, _hip.ptr(wd), _hip.ptr(gd), _hip.ptr(bd),
                                                 _hip.ptr(out), B, h, T, H, W, s, 1e-5, act, _st()))
        _close(out, ref, _tol(act, fp32=2e-5, bf16=1e-2))


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,h,Lq,Lk,add_q", [(1, 1, 128, 64, 1), (2, 2, 392, 392, 1), (1, 4, 100, 33, 0),
                                               (1, 1, 1568, 1568, 1), (2, 8, 49, 200, 1), (1, 2, 257, 6272, 1)])
def test_attention(hip_lib, act, B, h, Lq, Lk, add_q):
    q = _act(_rnd(B, h, Lq, 96, seed=16), act)
    k = _act(_rnd(B, h, Lk, 96, seed=17), act)
    v = _act(_rnd(B, h, Lk, 96, seed=18), act)
    scale = 96 ** -0.5
    p = ((q.float() @ k.float().transpose(-2, -1)) * scale).softmax(-1)
    ref = p @ v.float()
    if add_q:
        ref = ref + q.float()
    ref = ref.transpose(1, 2).reshape(B, Lq, h * 96)
    out = torch.empty(B, Lq, h * 96, dtype=q.dtype, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), None, B, h, Lq, Lk, scale,
                                          add_q, act, _st()))
    _close(out, ref, _tol(act, fp32=2e-5, bf16=1e-2))


def test_attention_bf16_online_softmax_rescale_is_exercised(hip_lib):
    """A key with a much larger score late in the sequence forces the running-max rescale branch."""
    B, h, Lq, Lk = 1, 1, 64, 256
    q = _rnd(B, h, Lq, 96, seed=19)
    k = _rnd(B, h, Lk, 96, seed=20)
    v = _rnd(B, h, Lk, 96, seed=21)
    k[0, 0, 200] = q[0, 0, 5] * 3.0    # spike: q5.k200 is huge, appears in the 4th tile
    k[0, 0, 70] = q[0, 0, 9] * 2.0
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    scale = 96 ** -0.5
    ref = (((q.float() @ k.float().transpose(-2, -1)) * scale).softmax(-1) @ v.float()).transpose(1, 2).reshape(B, Lq, 96)
    out = torch.empty(B, Lq, 96, dtype=torch.bfloat16, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), None, B, h, Lq, Lk, scale, 0,
                                          _hip.BF16, _st()))
    _close(out, ref, 1e-2)


@pytest.mark.parametrize("Lq,Lk", [(256, 448), (300, 549), (128, 64), (513, 1568 + 40)])
def test_attention_64_query_kernel_rescale_paths_and_lse(hip_lib, Lq, Lk):
    """csrc/attention_w64.hip (the default forward for Lk >= 64, Lq >= 128) keeps a LAGGING softmax reference per row: the output is
    rescaled only when a row's maximum jumps by more than 2^8, smaller jumps ride on the old reference.  Keys planted late in the
    sequence make rows take both paths (jumps of ~2^5 and ~2^40 in the exp2 domain), in the first key tile visited (the ragged one), in
    the middle and in the last tile, for odd and even tile counts; checked: the output and the saved row statistic
    lse = log2(sum_k exp(score)) (include/mvit_hip.h) that the backward pass starts from (attention.py:267-279)."""
    B, h = 1, 2
    q = _rnd(B, h, Lq, 96, seed=31)
    k = _rnd(B, h, Lk, 96, seed=32)
    v = _rnd(B, h, Lk, 96, seed=33)
    nt = (Lk + 63) // 64
    plant = [(3, Lk - 1, 3.0), (70, Lk - 5, 0.4), (5, 64 * (nt // 2) + 7, 2.5), (64, 64 * (nt // 2) + 9, 0.45), (17, 1, 3.0),
             (127, min(Lk - 1, 64 * (nt - 1) + 2), 0.5), (Lq - 1, Lk - 2, 2.0), (Lq - 2, 0, 0.4)]
    for qi, kj, c in plant:
        k[0, :, kj] = q[0, :, qi] * c
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    scale = 96 ** -0.5
    s = (q.float() @ k.float().transpose(-2, -1)) * scale
    ref = (s.softmax(-1) @ v.float() + q.float()).transpose(1, 2).reshape(B, Lq, h * 96)
    ref_lse = torch.logsumexp(s, -1) * 1.4426950408889634
    out = torch.empty(B, Lq, h * 96, dtype=torch.bfloat16, device=DEV)
    lse = torch.full((B, h, Lq), float("nan"), dtype=torch.float32, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, scale, 1,
                                          _hip.BF16, _st()))
    _close(out, ref, 1e-2)
    # the kernel's scores are those of the pre-scaled 16-bit queries round16(q * scale * log2e) (include/mvit_hip.h): a row's statistic
    # may sit |score| * 2^-9 away from the exact one (log2 units), on top of the 2e-3 of the diffuse rows
    bound = 2e-3 + (s.abs().amax(-1) * 1.4426950408889634) * 2.0 ** -9
    err = (lse.cpu() - ref_lse).abs()
    assert bool((err <= bound).all()), "lse err %.3e (bound there %.3e)" % (err.max().item(), bound.flatten()[err.argmax()].item())


@pytest.mark.parametrize("B,h,Lq,Lk", [(2, 8, 256 * 40, 200), (3, 1, 256 * 100 - 100, 130), (1, 8, 256 * 33 + 70, 64)])
def test_attention_64_query_kernel_persistent_grid(hip_lib, B, h, Lq, Lk):
    """More (256-query tile, batch x head) items than workgroups of the persistent grid of csrc/attention_w64.hip (one per CU): every
    workgroup then takes several items -- the next item's Q / K / V loads are issued before the current epilogue, the finished output tile
    waits in LDS and leaves behind the next item's operands, the two LDS regions of a wave alternate -- with batch x head a multiple of
    8 (items grouped per XCD) and not (plain order), ragged last query tiles, ragged and exact key tiles.  Output and lse against torch."""
    q = _rnd(B, h, Lq, 96, seed=41).to(torch.bfloat16)
    k = _rnd(B, h, Lk, 96, seed=42).to(torch.bfloat16)
    v = _rnd(B, h, Lk, 96, seed=43).to(torch.bfloat16)
    scale = 96 ** -0.5
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    s = (qd.float() @ kd.float().transpose(-2, -1)) * scale
    ref = (s.softmax(-1) @ vd.float() + qd.float()).transpose(1, 2).reshape(B, Lq, h * 96)
    ref_lse = torch.logsumexp(s, -1) * 1.4426950408889634
    out = torch.full((B, Lq, h * 96), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.full((B, h, Lq), float("nan"), dtype=torch.float32, device=DEV)
    for _ in range(2):      # (second call: same result from a warm cache and a reused LDS state)
        _hip.check(hip_lib.mvit_attention_fwd(_hip.ptr(qd), _hip.ptr(kd), _hip.ptr(vd), _hip.ptr(out), _hip.ptr(lse), B, h, Lq, Lk, scale, 1,
                                              _hip.BF16, _st()))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all()), "rows left unwritten"
        err = (out.float() - ref).abs().max().item()
        assert err <= 1e-2 * max(1.0, ref.abs().max().item()), "output err %.3e" % err
        bound = 2e-3 + (s.abs().amax(-1) * 1.4426950408889634) * 2.0 ** -9
        assert bool(((lse - ref_lse).abs() <= bound).all())


@pytest.mark.parametrize("B,T,H,W,C", [(2, 2, 16, 16, 192), (1, 3, 7, 7, 384), (1, 2, 14, 14, 768), (1, 1, 5, 9, 96)])
def test_maxpool_skip(hip_lib, B, T, H, W, C):
    x = _rnd(B, T * H * W, C, seed=22)
    t = x.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    ref = F.max_pool3d(t, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    Lo = ref.shape[2] * ref.shape[3] * ref.shape[4]
    ref = ref.reshape(B, C, Lo).transpose(1, 2)
    y = torch.empty(B, Lo, C, device=DEV)
    xd = x.to(DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_fwd(_hip.ptr(xd), _hip.ptr(y), B, T, H, W, C, _st()))
    assert torch.equal(y.cpu(), ref.contiguous())   # pure selection: bit-exact


@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 2, 16, 16, 96, 192), (1, 3, 7, 7, 192, 384), (1, 2, 14, 14, 384, 768), (1, 1, 5, 9, 96, 96),
                                              (1, 2, 28, 28, 192, 384)])
def test_proj_maxpool_fused_skip_path(hip_lib, B, T, H, W, Cin, Cout):
    """Widening skip path in one kernel (attention.py:424-432): bit-identical to the GEMM + max-pool pair of calls it replaces
    (values AND recorded arg-max bytes: ragged patches, odd sizes, frame borders), and equal to max_pool3d(linear) on the same
    bf16-rounded operands."""
    x = _rnd(B, T * H * W, Cin, seed=40)
    w = _rnd(Cout, Cin, seed=41, scale=Cin ** -0.5)
    bias = _rnd(Cout, seed=42, scale=0.1)
    xd, wd, bd = x.to(DEV), w.to(DEV).to(torch.bfloat16), bias.to(DEV)
    M = B * T * H * W
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Lo = T * Ho * Wo
    full = torch.empty(M, Cout, device=DEV)
    _hip.check(hip_lib.mvit_linear_fwd(_hip.ptr(xd), _hip.F32, Cin, _hip.ptr(wd), _hip.ptr(bd), None, Cout, None, 0, _hip.ptr(full), _hip.F32,
                                       Cout, M, Cout, Cin, _hip.EPI_BIAS, _hip.BF16, _st()))
    y0 = torch.empty(B, Lo, Cout, device=DEV)
    i0 = torch.empty(B, Lo, Cout, dtype=torch.uint8, device=DEV)
    _hip.check(hip_lib.mvit_maxpool_skip_fwd_idx(_hip.ptr(full), _hip.ptr(y0), _hip.ptr(i0), B, T, H, W, Cout, _st()))
    y1 = torch.full((B, Lo, Cout), float("nan"), device=DEV)
    i1 = torch.full((B, Lo, Cout), 255, dtype=torch.uint8, device=DEV)
    x16 = torch.full((M, Cin), float("nan"), dtype=torch.bfloat16, device=DEV)
    _hip.check(hip_lib.mvit_proj_maxpool_fwd(_hip.ptr(xd), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(y1), _hip.ptr(i1), _hip.ptr(x16), B, T, H, W,
                                             Cin, Cout, _hip.BF16, _st()))
    assert torch.equal(y1, y0) and torch.equal(i1, i0)
    assert torch.equal(x16, xd.reshape(M, Cin).to(torch.bfloat16))             # every token exactly once, rounded like the GEMM operand
    y2 = torch.full((B, Lo, Cout), float("nan"), device=DEV)        # inference form: no index output, no 16-bit copy
    _hip.check(hip_lib.mvit_proj_maxpool_fwd(_hip.ptr(xd), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(y2), None, None, B, T, H, W, Cin, Cout,
                                             _hip.BF16, _st()))
    assert torch.equal(y2, y0)
    ref = F.linear(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), bias)
    ref = F.max_pool3d(ref.reshape(B, T, H, W, Cout).permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    _close(y1, ref.reshape(B, Cout, Lo).transpose(1, 2), 1e-5)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("B,T,S", [(2, 4, 64), (1, 4, 56), (1, 16, 224), (4, 16, 224), (1, 8, 448)])      # the last two: workgroups walk runs of 2-4 tile-frames (frame ring)
def test_stem(hip_lib, act, B, T, S):
    clip = _rnd(B, 3, T, S, S, seed=23)
    w = _rnd(96, 3, 3, 7, 7, seed=24, scale=0.05)
    bias = _rnd(96, seed=25, scale=0.05)
    To, So = T // 2, S // 4
    ps, pt = _rnd(1, So * So, 96, seed=26, scale=0.02), _rnd(1, To, 96, seed=27, scale=0.02)
    ref = F.conv3d(clip, w, bias, stride=(2, 4, 4), padding=(1, 3, 3)).flatten(2).transpose(1, 2)
    ref = ref + (ps.repeat(1, To, 1) + torch.repeat_interleave(pt, So * So, dim=1))
    x = torch.empty(B, To * So * So, 96, device=DEV)
    args = [t.to(DEV) for t in (clip, w, bias, ps, pt)]
    _hip.check(hip_lib.mvit_stem_fwd(*[_hip.ptr(t) for t in args], _hip.ptr(x), B, T, S, act, _st()))
    _close(x, ref, _tol(act, fp32=2e-5, bf16=5e-3))


@pytest.mark.parametrize("B,N,C", [(2, 392, 768), (3, 32, 384), (1, 1568, 768), (2, 50, 96)])
def test_head(hip_lib, B, N, C):
    x = _rnd(B, N, C, seed=28) + 0.3
    g, b = 1 + 0.1 * _rnd(C, seed=29), 0.1 * _rnd(C, seed=30)
    w, hb = _rnd(18, C, seed=31, scale=0.05), _rnd(18, seed=32, scale=0.1)
    z = F.layer_norm(x, (C,), g, b, 1e-6).mean(1)
    ref_logits = F.linear(z, w, hb)
    ws = torch.empty(hip_lib.mvit_head_workspace_bytes(B, N, C) // 4, device=DEV)
    logits = torch.empty(B, 18, device=DEV)
    probs = torch.empty(B, 18, device=DEV)
    args = [t.to(DEV) for t in (x, g, b, w, hb)]
    _hip.check(hip_lib.mvit_head_fwd(*[_hip.ptr(t) for t in args], _hip.ptr(ws), _hip.ptr(logits), _hip.ptr(probs), B, N, C,
                                     18, 1e-6, _st()))
    _close(logits, ref_logits, 1e-5)
    _close(probs, ref_logits.softmax(1), 1e-5)


@pytest.mark.parametrize("act", [_hip.F32, _hip.BF16])
def test_head_split_round_trip(hip_lib, act):
    B, h, N = 2, 3, 77
    C = 96 * h
    qkv = _act(_rnd(B, N, 3 * C, seed=51), act).to(DEV)
    out = torch.empty(B, h, N, 96, dtype=qkv.dtype, device=DEV)
    _hip.check(hip_lib.mvit_head_split_fwd(_hip.ptr(qkv), 3 * C, C, _hip.ptr(out), B, h, N, act, _st()))
    ref = qkv[:, :, C:2 * C].reshape(B, N, h, 96).permute(0, 2, 1, 3)
    assert torch.equal(out, ref.contiguous())
    back = torch.zeros_like(qkv)
    _hip.check(hip_lib.mvit_head_split_bwd(_hip.ptr(out), _hip.ptr(back), 3 * C, C, B, h, N, act, _st()))
    assert torch.equal(back[:, :, C:2 * C], qkv[:, :, C:2 * C]) and float(back[:, :, :C].abs().max()) == 0.0


def _mlp_fused_case(L, half, M, C, seed, wscale=0.05):
    """out = x + fc2(GELU(fc1(LN(x)))) through mvit_mlp_fused_pack / _fwd of library L (16-bit type `half`) and the fp32 torch reference
    (slowfast/models/attention.py:436-445, common.py:26-34)."""
    hid = 4 * C
    x = _rnd(M, C, seed=seed) * 1.5 + 0.3
    gam, bet = 1 + 0.2 * _rnd(C, seed=seed + 1), 0.1 * _rnd(C, seed=seed + 2)
    w1, b1 = _rnd(hid, C, seed=seed + 3) * wscale, 0.1 * _rnd(hid, seed=seed + 4)
    w2, b2 = _rnd(C, hid, seed=seed + 5) * wscale, 0.1 * _rnd(C, seed=seed + 6)
    ref_mlp = F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), gam, bet, 1e-6), w1, b1)), w2, b2)
    d = [t.to(DEV) for t in (x, gam, bet, w1, b1, w2, b2)]
    nb = L.mvit_mlp_fused_pack_bytes(C, hid)
    assert nb == (hid // 32) * 128 * C + (4 * hid + 1023) // 1024 * 1024
    packed = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _hip.check(L.mvit_mlp_fused_pack(_hip.ptr(d[3]), _hip.ptr(d[4]), _hip.ptr(d[1]), _hip.ptr(d[2]), _hip.ptr(d[5]), _hip.ptr(packed), C, hid, _st()))
    out = torch.full((M, C), float("nan"), device=DEV)
    _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(d[0]), _hip.ptr(packed), _hip.ptr(d[6]), _hip.ptr(out), M, C, hid, 1e-6, _hip.BF16, _st()))
    torch.cuda.synchronize()
    got = out.cpu() - x                     # the MLP branch by itself: the residual would hide its error
    return got, ref_mlp, d, packed


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("M,C", [(392, 384), (128, 384), (1, 384), (6272 + 77, 384), (256, 192), (25088 + 3, 192), (65, 192),
                                 (256, 96), (100352 + 129, 96), (31, 96)])
def test_mlp_fused_forward(half, M, C):
    """Every (rows, width) family of the model (stage 1-3 block tails), ragged row counts on every tile size, both 16-bit builds,
    against LayerNorm -> Linear -> erf-GELU -> Linear in fp32."""
    L = _hip.lib(half)
    got, ref, _, _ = _mlp_fused_case(L, half, M, C, seed=40 + C // 96)
    assert torch.isfinite(got).all()
    tol = 2e-2 if half == "bf16" else 3e-3
    _close(got, ref, tol)
    rel = ((got - ref).norm() / ref.norm()).item()
    print("[mlp_fused %s M=%d C=%d] max|err| %.3e (scale %.2f)  relative L2 %.2e" % (half, M, C, (got - ref).abs().max().item(), ref.abs().max().item(), rel))
    assert rel <= (6e-3 if half == "bf16" else 8e-4)


@pytest.mark.parametrize("C", [96, 192, 384])
def test_mlp_fused_rows_do_not_depend_on_the_batch(C):
    """A token's result depends on its own row only: the same rows inside a short and inside a long launch (different tiles,
    different positions in a tile) are bit-identical -- what the window pipeline's batch-invariance tests rely on; out may alias x."""
    L = _hip.lib("fp16")
    hid = 4 * C
    got_a, _, d, packed = _mlp_fused_case(L, "fp16", 700, C, seed=77)
    xs = d[0][123:123 + 300].contiguous()
    out = torch.empty_like(xs)
    _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(xs), _hip.ptr(packed), _hip.ptr(d[6]), _hip.ptr(out), 300, C, hid, 1e-6, _hip.BF16, _st()))
    full = torch.empty_like(d[0])
    _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(d[0]), _hip.ptr(packed), _hip.ptr(d[6]), _hip.ptr(full), 700, C, hid, 1e-6, _hip.BF16, _st()))
    assert torch.equal(out, full[123:423])
    inpl = d[0].clone()
    _hip.check(L.mvit_mlp_fused_fwd(_hip.ptr(inpl), _hip.ptr(packed), _hip.ptr(d[6]), _hip.ptr(inpl), 700, C, hid, 1e-6, _hip.BF16, _st()))
    assert torch.equal(inpl, full)


def test_mlp_fused_unsupported_shapes_are_refused(hip_lib):
    assert hip_lib.mvit_mlp_fused_pack_bytes(768, 3072) == 0 and hip_lib.mvit_mlp_fused_pack_bytes(384, 1024) == 0
    x = torch.zeros(8, 768, device=DEV)
    rc = hip_lib.mvit_mlp_fused_fwd(_hip.ptr(x), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), 8, 768, 3072, 1e-6, _hip.BF16, _st())
    assert rc == -4
    rc = hip_lib.mvit_mlp_fused_fwd(_hip.ptr(x), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), 8, 384, 1536, 1e-6, _hip.F32, _st())
    assert rc == -4


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("M,C", [(392, 384), (1, 384), (6272 + 77, 384), (256, 192), (25088 + 3, 192), (256, 96), (100352 + 129, 96), (31, 96)])
def test_block_tail_proj_mlp_fused_forward(half, M, C):
    """mvit_block_tail_fwd: y = r + proj(o), out = y + fc2(GELU(fc1(LN(y)))) in one kernel (attention.py:281,434-445) against fp32
    torch on 16-bit-rounded o; the branch (out - r) is compared so the residual does not hide its error; ragged row counts; both builds;
    the same rows inside a longer launch are bit-identical; out may alias resid."""
    L = _hip.lib(half)
    dt = torch.bfloat16 if half == "bf16" else torch.float16
    hid = 4 * C
    sd = 60 + C // 96
    o = (_rnd(M, C, seed=sd) * 1.2).to(dt)
    res = _rnd(M, C, seed=sd + 1) * 1.5 + 0.2
    wp, bpj = _rnd(C, C, seed=sd + 2) * 0.05, 0.1 * _rnd(C, seed=sd + 3)
    gam, bet = 1 + 0.2 * _rnd(C, seed=sd + 4), 0.1 * _rnd(C, seed=sd + 5)
    w1, b1 = _rnd(hid, C, seed=sd + 6) * 0.05, 0.1 * _rnd(hid, seed=sd + 7)
    w2, b2 = _rnd(C, hid, seed=sd + 8) * 0.05, 0.1 * _rnd(C, seed=sd + 9)
    y = res + F.linear(o.float(), wp, bpj)
    ref = y + F.linear(F.gelu(F.linear(F.layer_norm(y, (C,), gam, bet, 1e-6), w1, b1)), w2, b2) - res
    d = {k: t.to(DEV) for k, t in dict(o=o, res=res, wp=wp, bpj=bpj, gam=gam, bet=bet, w1=w1, b1=b1, w2=w2, b2=b2).items()}
    nb = L.mvit_block_tail_pack_bytes(C, hid)
    assert nb == (C // 32) * 64 * C + L.mvit_mlp_fused_pack_bytes(C, hid) + (4 * C + 1023) // 1024 * 1024
    packed = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _hip.check(L.mvit_block_tail_pack(_hip.ptr(d["wp"]), _hip.ptr(d["bpj"]), _hip.ptr(d["w1"]), _hip.ptr(d["b1"]), _hip.ptr(d["gam"]), _hip.ptr(d["bet"]),
                                      _hip.ptr(d["w2"]), _hip.ptr(packed), C, hid, _st()))
    out = torch.full((M, C), float("nan"), device=DEV)
    _hip.check(L.mvit_block_tail_fwd(_hip.ptr(d["o"]), _hip.ptr(d["res"]), _hip.ptr(packed), _hip.ptr(d["b2"]), _hip.ptr(out), M, C, hid, 1e-6, _hip.BF16, _st()))
    got = out.cpu() - res
    assert torch.isfinite(got).all()
    _close(got, ref, 2e-2 if half == "bf16" else 3e-3)
    rel = ((got - ref).norm() / ref.norm()).item()
    print("[block_tail %s M=%d C=%d] max|err| %.3e (scale %.2f) relative L2 %.2e" % (half, M, C, (got - ref).abs().max().item(), ref.abs().max().item(), rel))
    assert rel <= (6e-3 if half == "bf16" else 8e-4)
    if M > 300:
        n = 200
        sub = torch.empty(n, C, device=DEV)
        os_, rs_ = d["o"][77:77 + n].contiguous(), d["res"][77:77 + n].contiguous()
        _hip.check(L.mvit_block_tail_fwd(_hip.ptr(os_), _hip.ptr(rs_), _hip.ptr(packed), _hip.ptr(d["b2"]), _hip.ptr(sub), n, C, hid, 1e-6,
                                         _hip.BF16, _st()))
        assert torch.equal(sub, out[77:77 + n])
        inpl = d["res"].clone()
        _hip.check(L.mvit_block_tail_fwd(_hip.ptr(d["o"]), _hip.ptr(inpl), _hip.ptr(packed), _hip.ptr(d["b2"]), _hip.ptr(inpl), M, C, hid, 1e-6, _hip.BF16, _st()))
        assert torch.equal(inpl, out)
