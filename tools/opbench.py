#!/usr/bin/env python3
"""Micro-benchmarks of single C-ABI operators on cuda:0 (timing with events on the launch stream).

    python tools/opbench.py gemm M N K [epi] [reps]     # bf16 a/w/out
    python tools/opbench.py attn B heads Lq Lk [reps]
    python tools/opbench.py pool B heads T H W stride [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd import _hip  # noqa: E402

L = _hip.lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    op = sys.argv[1]
    a = sys.argv[2:]
    if op == "gemm":
        M, N, K = int(a[0]), int(a[1]), int(a[2])
        epi = a[3] if len(a) > 3 else "b"
        reps = int(a[4]) if len(a) > 4 else 20
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        if os.environ.get("OPB_ZERO"):
            x.zero_(); w.zero_()
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if "r" in epi else None
        out_f32 = "r" in epi
        y = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
        flags = _hip.EPI_BIAS | (_hip.EPI_GELU if "g" in epi else 0) | (_hip.EPI_RESIDUAL if "r" in epi else 0) | int(os.environ.get("DIAG", "0"))

        def fn():
            _hip.check(L.mvit_linear_fwd(_hip.ptr(x), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(res), N, None, 0,
                                         _hip.ptr(y), _hip.F32 if out_f32 else _hip.BF16, N, M, N, K, flags, _hip.BF16, st))
        ms = timeit(fn, reps)
        fl = 2.0 * M * N * K
        by = M * K * 2 + N * K * 2 + M * N * (4 if out_f32 else 2) + (M * N * 4 if res is not None else 0)
        print("gemm M=%d N=%d K=%d epi=%s: %.1f us  %.1f TFLOP/s  %.2f TB/s" % (M, N, K, epi, ms * 1e3, fl / ms / 1e9, by / ms / 1e9))
        if os.environ.get("PP_STAMPS"):      # library built with -DPP_ABL=8 (tools/build_pp_abl.sh 8): loop cycles / clock / epilogue cycles of the ping-pong kernel
            import ctypes
            buf = (ctypes.c_float * (256 * 4))()
            L.mvit_debug_pp_stamps.restype = ctypes.c_int
            torch.cuda.synchronize()
            assert L.mvit_debug_pp_stamps(buf) == 0
            e = torch.tensor(list(buf)).view(256, 4)
            e = e[e[:, 2] > 0]
            big = e[e[:, 2] == e[:, 2].max()]
            print("  loop (longest workgroups, %d of %d): %.0f cycles, %.2f us, clock %.2f GHz, %d K-tiles -> %.0f cycles per K-tile incl. epilogues; epilogue %.0f cycles per tile (wave 0)" % (
                len(big), len(e), big[:, 0].mean(), big[:, 1].mean() / 100, big[:, 0].mean() / big[:, 1].mean() / 10, int(big[0, 2]), (big[:, 0] / big[:, 2]).mean(), big[:, 3].mean()))
    elif op == "gemmdual":      # the MLP pairs of a training step: pre | der (fc1 forward, two outputs), dgpre | dgder (fc2 data gradient x GELU')
        M, N, K = int(a[0]), int(a[1]), int(a[2])
        mode = a[3]
        reps = int(a[4]) if len(a) > 4 else 20
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        aux = torch.randn(M, N, device=dev).bfloat16()
        # OPB_Y_PAD=<bytes>: the second matrix starts this many bytes behind a 2 MiB boundary (the allocator hands out 2 MiB-aligned blocks, so
        # element (m, n) of both matrices otherwise maps to the same memory channel / bank)
        pad = int(os.environ.get("OPB_Y_PAD", "0")) // 2
        ybuf = torch.empty(M * N + pad, device=dev, dtype=torch.bfloat16)
        y = ybuf[pad:pad + M * N].view(M, N)
        sc = torch.rand(8, device=dev)
        rps = (M + 7) // 8

        def fn():
            if mode == "pre":
                _hip.check(L.mvit_linear_gelu_fwd(_hip.ptr(x), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(aux), _hip.ptr(y), M, N, K, _hip.BF16, st))
            elif mode == "der":
                _hip.check(L.mvit_linear_gelu_fwd_dsave(_hip.ptr(x), K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(aux), _hip.ptr(y), M, N, K, _hip.BF16, st))
            elif mode == "dgpre":
                _hip.check(L.mvit_linear_dgelu_fwd(_hip.ptr(x), K, _hip.ptr(w), _hip.ptr(sc), rps, _hip.ptr(aux), _hip.ptr(y), M, N, K, _hip.BF16, st))
            else:
                _hip.check(L.mvit_linear_dact_fwd(_hip.ptr(x), K, _hip.ptr(w), _hip.ptr(sc), rps, _hip.ptr(aux), _hip.ptr(y), M, N, K, _hip.BF16, st))
        ms = timeit(fn, reps)
        print("gemmdual M=%d N=%d K=%d mode=%s: %.1f us  %.1f TFLOP/s" % (M, N, K, mode, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    elif op == "attn":
        B, h, Lq, Lk = (int(v) for v in a[:4])
        reps = int(a[4]) if len(a) > 4 else 20
        q = torch.randn(B, h, Lq, 96, device=dev).bfloat16()
        k = torch.randn(B, h, Lk, 96, device=dev).bfloat16()
        v = torch.randn(B, h, Lk, 96, device=dev).bfloat16()
        if os.environ.get("OPB_ZERO"):       # all-zero operands: same instruction stream, far less switching power
            q.zero_(); k.zero_(); v.zero_()
        o = torch.empty(B, Lq, h * 96, device=dev, dtype=torch.bfloat16)

        def fn():
            _hip.check(L.mvit_attention_fwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), None, B, h, Lq, Lk, 96 ** -0.5, 1,
                                            _hip.BF16, st))
        ms = timeit(fn, reps)
        print("attn B=%d h=%d Lq=%d Lk=%d: %.1f us  %.1f TFLOP/s" % (B, h, Lq, Lk, ms * 1e3, 4.0 * B * h * Lq * Lk * 96 / ms / 1e9))
        if os.environ.get("W_STAMP"):        # library built with -DW_STAMP (attention_w64.hip): per-phase cycles per tile in the LSE buffer
            lse = torch.zeros(B * h * Lq, device=dev)
            _hip.check(L.mvit_attention_fwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), _hip.ptr(lse), B, h, Lq, Lk, 96 ** -0.5, 1,
                                            _hip.BF16, st))
            t8 = lse.view(B * h, Lq)[:, :(Lq // 256) * 256].reshape(B * h, Lq // 256, 256)[..., :32].reshape(-1, 8).float()
            t8 = t8[t8[:, 7] > 0]            # one row per wave of every workgroup's LAST item (averages over its items)
            print("w64 cycles per item (top .. operands landed, rest of the prologue, key loop end .. epilogue end, all):", [round(x, 1) for x in t8[:, 4:].mean(dim=0).tolist()], "rows", t8.shape[0])
            print("w64 cycles per tile (wait+barrier, phase 1, phase 2, rescale+rest):", [round(x, 1) for x in t8[:, :4].mean(dim=0).tolist()],
                  "sum", round(t8[:, :4].mean(dim=0).sum().item(), 1))
        if os.environ.get("ATT_STAMP"):      # library built with -DATT_STAMP: per-phase cycle averages land in the LSE buffer
            lse = torch.zeros(B * h * Lq, device=dev)
            _hip.check(L.mvit_attention_fwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), _hip.ptr(lse), B, h, Lq, Lk, 96 ** -0.5, 1,
                                            _hip.BF16, st))
            t = lse.view(B * h, Lq)[:, :(Lq // 128) * 128].reshape(B * h, Lq // 128, 4, 32)[..., :8].float()
            print("phase cycles/tile (wait, barrier+dma, S+max, softmax, PV issue, -):", [round(x, 1) for x in t.mean(dim=(0, 1, 2)).tolist()], "sum", round(t.mean(dim=(0, 1, 2)).sum().item(), 1))
    elif op == "attnbwd":
        B, h, Lq, Lk = (int(v) for v in a[:4])
        reps = int(a[4]) if len(a) > 4 else 10
        q = torch.randn(B, h, Lq, 96, device=dev).bfloat16()
        k = torch.randn(B, h, Lk, 96, device=dev).bfloat16()
        v = torch.randn(B, h, Lk, 96, device=dev).bfloat16()
        do = torch.randn(B, Lq, h * 96, device=dev).bfloat16()
        o = torch.empty(B, Lq, h * 96, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B, h, Lq, device=dev)
        _hip.check(L.mvit_attention_fwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), _hip.ptr(lse), B, h, Lq, Lk, 96 ** -0.5, 1, _hip.BF16, st))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(L.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk) // 4, device=dev)

        def fn():
            _hip.check(L.mvit_attention_bwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), _hip.ptr(lse), _hip.ptr(do), _hip.ptr(dq),
                                            _hip.ptr(dk), _hip.ptr(dv), _hip.ptr(ws), B, h, Lq, Lk, 96 ** -0.5, 1, _hip.BF16, st))
        ms = timeit(fn, reps)
        print("attnbwd B=%d h=%d Lq=%d Lk=%d: %.1f us  %.1f TFLOP/s credited (2x forward)" % (B, h, Lq, Lk, ms * 1e3, 8.0 * B * h * Lq * Lk * 96 / ms / 1e9))
        if os.environ.get("Y_STAMP"):        # library built with -DY_STAMP (attention_bwd_w64.hip): per-phase cycles per tile in the first floats of each wave's dQ rows
            fn(); torch.cuda.synchronize()
            t = dq.view(B * h, Lq, 96)[:, :(Lq // 256) * 256].reshape(B * h, Lq // 256, 4, 64 * 96)[..., :10].contiguous().view(torch.float32)[..., :5]
            m = t.float().mean(dim=(0, 1, 2)).tolist()
            print("dq w64 cycles per tile (wait+barrier+dma, A, B, C, D):", [round(x, 1) for x in m], "sum", round(sum(m), 1))
    elif op == "projpool":      # fused widening skip path (csrc/skip_pool.hip): B T H W Cin Cout [reps]; prints fused vs the unfused pair, fwd and bwd
        B, T, H, W, Cin, Cout = (int(v) for v in a[:6])
        reps = int(a[6]) if len(a) > 6 else 20
        M = B * T * H * W
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        Mo = B * T * Ho * Wo
        x = torch.randn(M, Cin, device=dev)
        w = (torch.randn(Cout, Cin, device=dev) * Cin ** -0.5).bfloat16()
        wt = w.t().contiguous()
        bias = torch.randn(Cout, device=dev)
        full = torch.empty(M, Cout, device=dev)
        y = torch.empty(Mo, Cout, device=dev)
        idx = torch.empty(Mo, Cout, dtype=torch.uint8, device=dev)
        dy = torch.randn(Mo, Cout, device=dev)
        dx = torch.empty(M, Cin, device=dev)
        d16 = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
        x16 = torch.empty(M, Cin, device=dev, dtype=torch.bfloat16)

        def unfused_fwd():
            _hip.check(L.mvit_linear_fwd(_hip.ptr(x), _hip.F32, Cin, _hip.ptr(w), _hip.ptr(bias), None, Cout, None, 0, _hip.ptr(full), _hip.F32, Cout, M,
                                         Cout, Cin, _hip.EPI_BIAS, _hip.BF16, st))
            _hip.check(L.mvit_maxpool_skip_fwd_idx(_hip.ptr(full), _hip.ptr(y), _hip.ptr(idx), B, T, H, W, Cout, st))

        def fused_fwd():
            _hip.check(L.mvit_proj_maxpool_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(y), _hip.ptr(idx), _hip.ptr(x16), B, T, H, W, Cin, Cout, _hip.BF16, st))

        def unfused_bwd():
            _hip.check(L.mvit_maxpool_skip_bwd_idx(_hip.ptr(idx), _hip.ptr(dy), _hip.ptr(full), B, T, H, W, Cout, st))
            _hip.check(L.mvit_linear_fwd(_hip.ptr(full), _hip.F32, Cout, _hip.ptr(wt), None, None, Cin, None, 0, _hip.ptr(dx), _hip.F32, Cin, M, Cin,
                                         Cout, 0, _hip.BF16, st))

        def fused_bwd():
            _hip.check(L.mvit_proj_maxpool_bwd(_hip.ptr(idx), _hip.ptr(dy), _hip.ptr(wt), _hip.ptr(dx), _hip.ptr(d16), B, T, H, W, Cin, Cout, _hip.BF16, st))
        alg_f = M * Cin * 4 + Mo * Cout * 5
        alg_b = Mo * Cout * 5 + M * Cin * 4 + M * Cout * 2
        for name, fn, alg in (("fwd unfused", unfused_fwd, alg_f), ("fwd fused", fused_fwd, alg_f), ("bwd unfused", unfused_bwd, alg_b), ("bwd fused", fused_bwd, alg_b)):
            ms = timeit(fn, reps)
            print("projpool %s B=%d T=%d %dx%d %d->%d: %.1f us  (fused form's algorithmic bytes %.0f MB -> %.2f TB/s)" % (name, B, T, H, W, Cin, Cout, ms * 1e3, alg / 1e6, alg / ms / 1e9))
    elif op == "stem":
        B = int(a[0]); reps = int(a[1]) if len(a) > 1 else 20
        clip = torch.randn(B, 3, 16, 448, 448, device=dev)
        w = torch.randn(96, 3, 3, 7, 7, device=dev) * 0.05
        bias = torch.randn(96, device=dev)
        ps, pt = torch.randn(112 * 112, 96, device=dev), torch.randn(8, 96, device=dev)
        x = torch.empty(B, 8 * 112 * 112, 96, device=dev)

        def fn():
            _hip.check(L.mvit_stem_fwd(_hip.ptr(clip), _hip.ptr(w), _hip.ptr(bias), _hip.ptr(ps), _hip.ptr(pt), _hip.ptr(x), B, 16, 448, _hip.BF16, st))
        ms = timeit(fn, reps)
        print("stem B=%d: %.1f us  %.1f TFLOP/s  %.2f TB/s" % (B, ms * 1e3, B * 8.5e9 / ms / 1e9, (clip.numel() + x.numel()) * 4 / ms / 1e9))
    elif op == "stembwd":
        B = int(a[0]); reps = int(a[1]) if len(a) > 1 else 20
        clip = torch.randn(B, 3, 16, 448, 448, device=dev)
        dx = torch.randn(B, 8 * 112 * 112, 96, device=dev)
        dW = torch.zeros(96, 441, device=dev)
        dps, dpt = torch.zeros(112 * 112, 96, device=dev), torch.zeros(8, 96, device=dev)
        nb = L.mvit_stem_bwd_workspace_bytes(B, 16, 448, _hip.BF16)
        ws = torch.empty(nb // 4, device=dev)

        def fn():
            _hip.check(L.mvit_stem_bwd(_hip.ptr(clip), _hip.ptr(dx), _hip.ptr(dW), _hip.ptr(dps), _hip.ptr(dpt), B, 16, 448, _hip.BF16, _hip.ptr(ws), nb, st))
        ms = timeit(fn, reps)
        print("stem bwd (weight gradient + slab sum + position-embedding gradients) B=%d: %.1f us" % (B, ms * 1e3))
    elif op == "wgrad":
        M, N, K = int(a[0]), int(a[1]), int(a[2])
        reps = int(a[3]) if len(a) > 3 else 20
        x = torch.randn(M, K, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        dW = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)

        nb = L.mvit_linear_wgrad_workspace_bytes(_hip.BF16, K, _hip.BF16, N, 0, M, N, K, _hip.BF16)     # slab form (what training runs)
        ws = torch.empty(max(nb // 4, 1), device=dev)

        def fn():
            _hip.check(L.mvit_linear_wgrad(_hip.ptr(x), _hip.BF16, K, _hip.ptr(dy), _hip.BF16, N, None, 0, _hip.ptr(dW), _hip.ptr(db), M, N, K,
                                           _hip.BF16, _hip.ptr(ws), nb, st))
        ms = timeit(fn, reps)
        print("wgrad M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    elif op == "pool":
        B, h, T, H, W, s = (int(v) for v in a[:6])
        reps = int(a[6]) if len(a) > 6 else 20
        C = 96 * h
        qkv = torch.randn(B, T * H * W, 3 * C, device=dev).bfloat16()
        w = torch.randn(96, 27, device=dev) * 0.2
        g, bt = torch.ones(96, device=dev), torch.zeros(96, device=dev)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        out = torch.empty(B, h, T * Ho * Wo, 96, device=dev, dtype=torch.bfloat16)

        def fn():
            _hip.check(L.mvit_pool_conv_ln_fwd(_hip.ptr(qkv), 3 * C, 0, _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt), _hip.ptr(out),
                                               B, h, T, H, W, s, 1e-5, _hip.BF16, st))
        ms = timeit(fn, reps)
        ntok = B * h * T * Ho * Wo
        print("pool B=%d h=%d THW=%dx%dx%d s=%d: %.1f us  %.2f Gtok/s  %.2f TB/s(out+in slice)" % (
            B, h, T, H, W, s, ms * 1e3, ntok / ms / 1e6, (ntok * 192 + B * h * T * H * W * 192) / ms / 1e9))
    elif op == "poolbwd":       # backward of one pooling conv + LayerNorm from saved statistics: B h T H W stride [reps]; MVIT_POOL_WGRAD_MARCH=0/1 A/B
        B, h, T, H, W, s = (int(v) for v in a[:6])
        reps = int(a[6]) if len(a) > 6 else 20
        C = 96 * h
        qkv = torch.randn(B, T * H * W, 3 * C, device=dev).bfloat16()
        w = torch.randn(96, 27, device=dev) * 0.2
        g = torch.ones(96, device=dev)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        ntok = B * h * T * Ho * Wo
        xh = torch.randn(B, h, T * Ho * Wo, 96, device=dev).bfloat16()
        rs = torch.rand(ntok, device=dev) + 0.5
        dout = torch.randn_like(xh)
        dconv = torch.empty_like(xh)
        dqkv = torch.zeros_like(qkv)
        dw, dg, db = torch.zeros(96, 27, device=dev), torch.zeros(96, device=dev), torch.zeros(96, device=dev)
        nb = L.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, s)
        ws = torch.empty(nb // 4, device=dev)

        def fn():
            _hip.check(L.mvit_pool_conv_ln_bwd_saved(_hip.ptr(qkv), 3 * C, 0, _hip.ptr(w), _hip.ptr(g), _hip.ptr(xh), _hip.ptr(rs), _hip.ptr(dout),
                                                     _hip.ptr(dconv), _hip.ptr(dqkv), _hip.ptr(dw), _hip.ptr(dg), _hip.ptr(db), 1, _hip.ptr(ws),
                                                     B, h, T, H, W, s, 1e-5, _hip.BF16, st))
        ms = timeit(fn, reps)
        print("pool backward (LN bwd + wgrad + reduces + dgrad) B=%d h=%d THW=%dx%dx%d s=%d: %.1f us" % (B, h, T, H, W, s, ms * 1e3))
    elif op == "poolkv":
        B, h, T, H, W = (int(v) for v in a[:5])
        reps = int(a[5]) if len(a) > 5 else 20
        C = 96 * h
        qkv = torch.randn(B, T * H * W, 3 * C, device=dev).bfloat16()
        w = torch.randn(96, 27, device=dev) * 0.2
        g, bt = torch.ones(96, device=dev), torch.zeros(96, device=dev)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty(2, B, h, T * Ho * Wo, 96, device=dev, dtype=torch.bfloat16)

        def pair():
            _hip.check(L.mvit_pool_conv_ln_fwd_train_kv(_hip.ptr(qkv), 3 * C, C, _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt), _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt),
                                                        _hip.ptr(out), None, None, B, h, T, H, W, 2, 1e-5, _hip.BF16, st))

        def two():
            for i in range(2):
                _hip.check(L.mvit_pool_conv_ln_fwd(_hip.ptr(qkv), 3 * C, C * (1 + i), _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt), _hip.ptr(out[i]),
                                                   B, h, T, H, W, 2, 1e-5, _hip.BF16, st))
        print("pool k+v B=%d h=%d THW=%dx%dx%d stride 2: pair launch %.1f us, two launches %.1f us" % (B, h, T, H, W, timeit(pair, reps) * 1e3, timeit(two, reps) * 1e3))
        # the same launch on inputs that are NOT cache-resident: a ring of buffers larger than the 256 MB memory-side cache
        nbuf = max(2, int(600e6 // (qkv.numel() * 2)) + 1)
        ring = [torch.randn_like(qkv) for _ in range(nbuf)]
        k = [0]

        def pair_cold():
            q_ = ring[k[0] % nbuf]
            k[0] += 1
            _hip.check(L.mvit_pool_conv_ln_fwd_train_kv(_hip.ptr(q_), 3 * C, C, _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt), _hip.ptr(w), _hip.ptr(g), _hip.ptr(bt),
                                                        _hip.ptr(out), None, None, B, h, T, H, W, 2, 1e-5, _hip.BF16, st))
        print("   ... on a ring of %d input buffers (cold reads): pair launch %.1f us" % (nbuf, timeit(pair_cold, reps) * 1e3))


if __name__ == "__main__":
    main()
