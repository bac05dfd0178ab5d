"""Training path: forward that keeps activations + hand-written HIP backward, glued with torch.autograd.

One ``autograd.Function`` per stem / block / head, so the reference's own loop (``loss.backward()``,
``clip_grad_norm_``, ``optimizer.step()``, DDP bucketed all-reduce overlapping with the per-block backward) works
unchanged on the module's ordinary ``nn.Parameter``s (tools/train_net.py:201-246).  Every arithmetic step is a C-ABI
kernel; torch supplies allocation, the RNG draws for drop-path / dropout (slowfast/models/common.py:46-59,
head_helper.py:410-411) and the autograd graph.

Mixed-precision policy of the bf16 path: MFMA operands and stored activations bf16; residual stream, LayerNorm
statistics, softmax statistics, accumulators, parameter gradients and optimizer state fp32 (fp32 masters).
"""
import os

import torch

from . import _hip


def _st():
    return torch.cuda.current_stream().cuda_stream


_REDUCE_QUEUE = os.environ.get("MVIT_REDUCE_QUEUE", "1") != "0"     # A/B switch: 0 = every reduction is its own launch
_LN_EMIT16 = os.environ.get("MVIT_LN_EMIT16", "1") != "0"           # A/B switch: 0 = separate cast passes over the stream gradient
# k / v pooling convs of the training forward beside the q one on the library's side stream: measured -0.15 ms per step when OFF
# (profiles/r2_side_stream_ab.txt: since the kernels lost their long tails, interleaving two of them on the CUs costs more than
# the filled tail returns); MVIT_POOL_FWD_SIDE=1 turns it back on
_POOL_FWD_SIDE = os.environ.get("MVIT_POOL_FWD_SIDE", "0") == "1"
# the k and v pooling convs of a block (and their backward) as ONE set of launches where the library has that form (stride 2:
# mvit_pool_conv_ln_*_kv); 0 = two single-tensor calls
_POOL_KV_BATCH = os.environ.get("MVIT_POOL_KV_BATCH", "1") != "0"


_SKIP_FUSE = os.environ.get("MVIT_SKIP_FUSE", "1") != "0"      # read once


def _skip_fused(g, act, B=1):
    """Widening stage-transition blocks (MViTv2-B: 1, 3, 14) take the fused skip path of csrc/skip_pool.hip on the 16-bit builds.
    The conditions mirror what mvit_proj_maxpool_fwd / _bwd accept (input widths 96 / 192 / 384, output a multiple of 96, 32-bit
    offsets); anything else -- a 768 -> 1536 stage, a very large batch -- takes the unfused linear + max-pool pair."""
    if not (_SKIP_FUSE and g.expand and not g.skip_is_identity and act != _hip.F32):
        return False
    if g.dim_in not in (96, 192, 384) or g.dim_out % 96 != 0:
        return False
    n_in = B * g.thw_in[0] * g.thw_in[1] * g.thw_in[2]
    return n_in * g.dim_in < (1 << 31) and B * g.lq * g.dim_out < (1 << 30)


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=dev)


class _Ctx(object):
    """Per-model helper: dtype policy + cached (transposed) low-precision weights."""

    def __init__(self, model):
        self.m = model
        # "auto" is resolved HERE, once per graph: inside the autograd backward grad mode is off and the property would answer
        # "fp16" for a graph whose forward ran in bf16 (eval-mode model with grad enabled)
        self.prec = model.precision
        self.L = model._lib(self.prec)
        self.act = _hip.F32 if self.prec == "fp32" else _hip.BF16
        self.adt = torch.float32 if self.act == _hip.F32 else model._half_dtype(self.prec)
        self._red_keep = None   # workspaces pinned while a deferred-reduction queue is open (_BlockFn.backward)
        self._last_block = None  # (index, dp2, output data_ptr) of the block that ran last in this chain (forward)
        self._g16_stash = None   # (data_ptr of a stream gradient, its 16-bit scaled copy) handed from one block backward to the next
        self._zpools = {}       # one zero pool per HIP stream (sub-batches of a step run on side streams)
        self._side_out = []

    def zeros(self, *shape):
        """fp32 zeros carved from ONE zero-filled buffer per step (the ~290 small gradient buffers of a backward pass
        would otherwise cost a fill launch each); 256-byte aligned views."""
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        dev = next(self.m.parameters()).device
        key = torch.cuda.current_stream(dev).cuda_stream
        ent = self._zpools.get(key)
        if ent is None or ent[1] + n_al > ent[0].numel():
            total = sum((p.numel() + 63) // 64 * 64 for p in self.m.parameters())
            ent = [torch.zeros(max(total, n_al), dtype=torch.float32, device=dev), 0]
            self._zpools[key] = ent
        t = ent[0][ent[1]:ent[1] + n].view(*shape)
        ent[1] += n_al
        return t

    def w(self, p):
        return self.m._w_pair(p, self.act, self.prec)[0]

    def wt(self, p):
        """[K][N] transposed copy in the activation dtype (operand of the data-gradient GEMM)."""
        return self.m._w_pair(p, self.act, self.prec)[1]

    # y = a . w^T (+bias)(gelu)(*row_scale)(+residual)
    def linear(self, a, w, bias, out_dtype, residual=None, gelu=False, row_scale=None, rps=0):
        M, K = a.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=out_dtype, device=a.device)
        epi = (_hip.EPI_BIAS if bias is not None else 0) | (_hip.EPI_GELU if gelu else 0) | (
            _hip.EPI_RESIDUAL if residual is not None else 0)
        adt = _hip.F32 if a.dtype == torch.float32 else _hip.BF16
        odt = _hip.F32 if out_dtype == torch.float32 else _hip.BF16
        _hip.check(self.L.mvit_linear_fwd(_hip.ptr(a), adt, K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(residual), N,
                                          _hip.ptr(row_scale), rps, _hip.ptr(y), odt, N, M, N, K, epi, self.act, _st()),
                   "linear %dx%dx%d" % (M, N, K))
        return y

    def wgrad(self, a, dy, N, K, row_scale=None, rps=0):
        """(dW, db): weight gradient and the fused bias gradient (column sums of the scaled dy).
        With HIP.WGRAD_STREAM (default on) the kernel is issued on a side stream: nothing in the backward chain consumes dW, so it
        runs beside the data-gradient / attention-backward kernels of the same block and fills their partially occupied last
        waves of workgroups; ``join_side()`` at the end of the block's backward makes the results visible to the caller's stream."""
        adt = _hip.F32 if a.dtype == torch.float32 else _hip.BF16
        ddt = _hip.F32 if dy.dtype == torch.float32 else _hip.BF16
        M = a.shape[0]

        def launch():
            # deterministic form: per-chunk partial slabs in a workspace, added in chunk order (no float atomics)
            dW, db = self.zeros(N, K), self.zeros(N)
            nb = self.L.mvit_linear_wgrad_workspace_bytes(adt, K, ddt, N, 1 if row_scale is not None else 0, M, N, K, self.act)
            ws = _ws(nb, a.device)
            _hip.check(self.L.mvit_linear_wgrad(_hip.ptr(a), adt, K, _hip.ptr(dy), ddt, N, _hip.ptr(row_scale), rps, _hip.ptr(dW),
                                                 _hip.ptr(db), M, N, K, self.act, _hip.ptr(ws), nb, _st()), "wgrad")
            return dW, db
        side = self._side()
        if side is None:
            return launch()
        cur = torch.cuda.current_stream(a.device)
        side.wait_stream(cur)                       # the operands were produced on the caller's stream
        with torch.cuda.stream(side):
            dW, db = launch()
        for t in (a, dy, row_scale):
            if t is not None:
                t.record_stream(side)               # keep the operands' memory until the side stream is done with it
        self._side_out += [dW, db]
        return dW, db

    def _side(self):
        hip = getattr(self.m.cfg, "HIP", None)
        if os.environ.get("MVIT_WGRAD_STREAM", "1") == "0" or not (bool(getattr(hip, "WGRAD_STREAM", True)) if hip is not None else True):
            return None
        dev = next(self.m.parameters()).device
        self.m._wgrad_stream = _hip.shared_streams(dev, 1, "wgrad")[0]
        return self.m._wgrad_stream

    def join_side(self):
        """The caller's stream waits for the side-stream weight gradients issued so far."""
        if not self._side_out:
            return
        dev = self._side_out[0].device
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(self.m._wgrad_stream)
        for t in self._side_out:
            t.record_stream(cur)
        self._side_out = []

    def scaled16(self, dy, row_scale=None, rps=0):
        """fp32 gradient (times its drop-path factor) as the 16-bit GEMM operand; (tensor, row_scale, rps) to pass on."""
        if self.act == _hip.F32:
            return dy, row_scale, rps
        out = torch.empty(dy.shape, dtype=self.adt, device=dy.device)
        _hip.check(self.L.mvit_cast_rows_f32_to_bf16(_hip.ptr(dy), _hip.ptr(out), dy.shape[0], dy.shape[1], _hip.ptr(row_scale), rps,
                                                     _st()), "cast_rows")
        return out, None, 0

    def colsum(self, dy, row_scale=None, rps=0):
        M, N = dy.shape
        out = torch.empty(N, dtype=torch.float32, device=dy.device)
        ws = _ws(self.L.mvit_colsum_workspace_bytes(N), dy.device)
        ddt = _hip.F32 if dy.dtype == torch.float32 else _hip.BF16
        _hip.check(self.L.mvit_colsum(_hip.ptr(dy), ddt, M, N, _hip.ptr(row_scale), rps, _hip.ptr(out), 0, _hip.ptr(ws), _st()),
                   "colsum")
        return out

    def ln_fwd(self, x, norm):
        rows, C = x.shape
        y = torch.empty(rows, C, dtype=self.adt, device=x.device)
        _hip.check(self.L.mvit_layernorm_fwd(_hip.ptr(x), _hip.ptr(norm.weight), _hip.ptr(norm.bias), _hip.ptr(y), rows, C,
                                             norm.eps, self.act, _st()), "ln")
        return y

    def ln_bwd(self, x, norm, dy, dx, accumulate, rows_per_dy=1, dy_scale=1.0, base=None, emit16=None):
        """dx = [base | dx if accumulate | 0] + LN-backward(dy); returns (dgamma, dbeta).  emit16 = (row_scale | None, rows_per_scale):
        also returns dx * row_scale as the 16-bit operand of the GEMMs that consume it next (None on the fp32 path) -- the copy
        mvit_cast_rows_f32_to_bf16 would make, written by the same kernel instead of a second pass over dx."""
        rows, C = x.shape
        dg = self.zeros(C)          # pre-zeroed pool slices + accumulate: the sliced partial reduction needs no memset
        db = self.zeros(C)
        ws = _ws(self.L.mvit_layernorm_bwd_workspace_bytes(C), x.device)
        if self._red_keep is not None:      # a reduce queue is open: the partial table in ws is read at the flush
            self._red_keep.append(ws)
        ddt = _hip.F32 if dy.dtype == torch.float32 else _hip.BF16
        if base is None and accumulate:
            base = dx
        out16, sc16, rps16 = None, None, 0
        if emit16 is not None and self.act != _hip.F32 and _LN_EMIT16:
            out16 = torch.empty(dx.shape, dtype=self.adt, device=dx.device)
            sc16, rps16 = emit16
        _hip.check(self.L.mvit_layernorm_bwd(_hip.ptr(x), _hip.ptr(norm.weight), _hip.ptr(dy), ddt, rows_per_dy, dy_scale,
                                              _hip.ptr(base), _hip.ptr(dx), _hip.ptr(dg), _hip.ptr(db), 1, _hip.ptr(ws),
                                              rows, C, norm.eps, _hip.ptr(out16), _hip.ptr(sc16), rps16 if sc16 is not None else 0,
                                              _st()), "ln_bwd")
        if emit16 is not None:
            return dg, db, out16
        return dg, db


def _block_params(blk, g):
    at = blk.attn
    ps = [blk.norm1.weight, blk.norm1.bias, at.qkv.weight, at.qkv.bias, at.proj.weight, at.proj.bias]
    if g.kernel_q:
        ps += [at.pool_q.weight, at.norm_q.weight, at.norm_q.bias]
    ps += [at.pool_k.weight, at.norm_k.weight, at.norm_k.bias, at.pool_v.weight, at.norm_v.weight, at.norm_v.bias,
           blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]
    if g.expand:
        ps += [blk.proj_max_pool.weight, blk.proj_max_pool.bias]
    return ps


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, w, b, ps, pt, hx):
        m = hx.m
        B, _, T, S, _ = clip.shape
        Tp, Hp, Wp = m.patch_dims
        x = torch.empty(B, Tp * Hp * Wp, 96, dtype=torch.float32, device=clip.device)
        _hip.check(hx.L.mvit_stem_fwd(_hip.ptr(clip), _hip.ptr(w), _hip.ptr(b), _hip.ptr(ps), _hip.ptr(pt), _hip.ptr(x), B, T, S,
                                      hx.act, _st()), "stem")
        ctx.hx, ctx.clip = hx, clip
        return x

    @staticmethod
    def backward(ctx, dx):
        hx, clip = ctx.hx, ctx.clip
        m = hx.m
        dx = dx.contiguous()
        B, _, T, S, _ = clip.shape
        dev = clip.device
        dW = hx.zeros(96, 3, 3, 7, 7)
        dps = hx.zeros(*m.pos_embed_spatial.shape)
        dpt = hx.zeros(*m.pos_embed_temporal.shape)
        nb = hx.L.mvit_stem_bwd_workspace_bytes(B, T, S, hx.act)        # slab form: ordered sums, no float atomics
        ws = _ws(nb, dev)
        _hip.check(hx.L.mvit_stem_bwd(_hip.ptr(clip), _hip.ptr(dx), _hip.ptr(dW), _hip.ptr(dps), _hip.ptr(dpt), B, T, S, hx.act,
                                       _hip.ptr(ws), nb, _st()), "stem_bwd")
        db = hx.colsum(dx.view(-1, 96))
        return None, dW, db, dps, dpt, None


class _BlockFn(torch.autograd.Function):
    """One MultiScaleBlock.  With ``hx.m.use_act_checkpoint`` (MODEL.ACT_CHECKPOINT, video_model_builder.py:1036-1037: the
    reference wraps every block in fairscale's checkpoint_wrapper) only the block input and the two drop-path draws are kept:
    the forward kernels run a second time at the start of ``backward`` to rebuild the activations, exactly the memory / time
    trade of the reference's recipe (README.md:84,96-114).  ``_BlockFn.forward_launches`` counts forward executions (tests)."""
    forward_launches = 0

    @staticmethod
    def forward(ctx, x, hx, g, blk, dp1, dp2, *params):
        out, saved, pool_saved, mlp_dsave = _BlockFn._run_forward(x, hx, g, blk, dp1, dp2)
        ctx.hx, ctx.g, ctx.blk, ctx.addq = hx, g, blk, 1 if hx.m.use_query_residual_pool else 0
        ctx.mlp_dsave = mlp_dsave
        # the block upstream in this chain and its MLP drop-path draw: this block's backward hands that block its incoming gradient
        # already cast (see backward)
        prev = hx._last_block
        ctx.prev_dp2 = (prev[1],) if (prev is not None and prev[0] == g.index - 1 and prev[2] == x.data_ptr()) else None
        hx._last_block = (g.index, dp2, out.data_ptr())
        if hx.m.use_act_checkpoint:
            ctx.saved, ctx.pool_saved = None, None
            ctx.recompute = (x, dp1, dp2)
        else:
            ctx.saved, ctx.pool_saved = saved, pool_saved
            ctx.recompute = None
        return out

    @staticmethod
    def _run_forward(x, hx, g, blk, dp1, dp2):
        _BlockFn.forward_launches += 1
        mlp_dsave = False
        L, act, adt = hx.L, hx.act, hx.adt
        dev = x.device
        at = blk.attn
        B = x.shape[0]
        T, H, W = g.thw_in
        N, Lq, Lk = g.n_in, g.lq, g.lk
        M, Mq = B * N, B * Lq
        Cin, Cout, h = g.dim_in, g.dim_out, g.heads
        x2 = x.view(M, Cin)
        u = hx.ln_fwd(x2, blk.norm1)
        qkv = hx.linear(u, hx.w(at.qkv.weight), at.qkv.bias, adt)
        q = torch.empty(B, h, Lq, 96, dtype=adt, device=dev)
        kv = torch.empty(2, B, h, Lk, 96, dtype=adt, device=dev)      # k and v back to back (the pair form of the pooling kernels)
        k, v = kv[0], kv[1]
        kv_batch = _POOL_KV_BATCH and g.stride_kv[1] == 2 and os.environ.get("MVIT_POOL_RECOMPUTE", "0") != "1"
        pools = [] if kv_batch else [(1, k, at.pool_k, at.norm_k, g.stride_kv[1]), (2, v, at.pool_v, at.norm_v, g.stride_kv[1])]
        if g.kernel_q:
            pools.insert(0, (0, q, at.pool_q, at.norm_q, g.stride_q[1]))
        else:
            _hip.check(L.mvit_head_split_fwd(_hip.ptr(qkv), 3 * Cout, 0, _hip.ptr(q), B, h, N, act, _st()), "head_split")
        forked = _POOL_FWD_SIDE and L.mvit_side_fork(_st()) == 0      # k / v pooling convs beside the q one (independent readers of qkv)
        side = L.mvit_side_stream() if forked else _st()
        pool_saved = {}     # which -> (xhat, rstd): what the LayerNorm backward needs, so the backward runs no second convolution
        save_ln = os.environ.get("MVIT_POOL_RECOMPUTE", "0") != "1"
        for which, buf, conv, norm, stride in pools:
            xh = torch.empty_like(buf) if save_ln else None
            rs = torch.empty(buf.shape[0] * buf.shape[1] * buf.shape[2], dtype=torch.float32, device=dev) if save_ln else None
            pool_saved[which] = (xh, rs)
            _hip.check(L.mvit_pool_conv_ln_fwd_train(_hip.ptr(qkv), 3 * Cout, which * Cout, _hip.ptr(conv.weight), _hip.ptr(norm.weight),
                                                     _hip.ptr(norm.bias), _hip.ptr(buf), _hip.ptr(xh), _hip.ptr(rs), B, h, T, H, W, stride,
                                                     norm.eps, act, _st() if which == 0 else side), "pool")
        if kv_batch:
            xh_kv = torch.empty_like(kv)
            rs_kv = torch.empty(2, B * h * Lk, dtype=torch.float32, device=dev)
            _hip.check(L.mvit_pool_conv_ln_fwd_train_kv(_hip.ptr(qkv), 3 * Cout, Cout, _hip.ptr(at.pool_k.weight), _hip.ptr(at.norm_k.weight),
                                                        _hip.ptr(at.norm_k.bias), _hip.ptr(at.pool_v.weight), _hip.ptr(at.norm_v.weight),
                                                        _hip.ptr(at.norm_v.bias), _hip.ptr(kv), _hip.ptr(xh_kv), _hip.ptr(rs_kv), B, h, T, H, W,
                                                        g.stride_kv[1], at.norm_k.eps, act, side), "pool_kv")
            pool_saved["kv"] = (xh_kv, rs_kv)
        if forked:
            _hip.check(L.mvit_side_join(_st()), "side_join")
        o = torch.empty(Mq, Cout, dtype=adt, device=dev)
        lse = torch.empty(B, h, Lq, dtype=torch.float32, device=dev)
        addq = 1 if hx.m.use_query_residual_pool else 0
        _hip.check(_hip.attention_fwd(L, q, k, v, o, lse, B, h, Lq, Lk, 96 ** -0.5, addq, act, _st()), "attention")
        r = x2
        r_full = None
        if _skip_fused(g, act, B):
            # widen + max-pool in one kernel: the full-resolution widened tensor never reaches HBM (csrc/skip_pool.hip)
            r = torch.empty(Mq, Cout, dtype=torch.float32, device=dev)
            pool_idx = torch.empty(Mq, Cout, dtype=torch.uint8, device=dev)
            x16 = torch.empty(M, Cin, dtype=adt, device=dev)          # the block input rounded once: operand of the weight gradient
            _hip.check(L.mvit_proj_maxpool_fwd(_hip.ptr(x2), _hip.ptr(hx.w(blk.proj_max_pool.weight)), _hip.ptr(blk.proj_max_pool.bias),
                                               _hip.ptr(r), _hip.ptr(pool_idx), _hip.ptr(x16), B, T, H, W, Cin, Cout, act, _st()),
                       "proj_maxpool")
            r_full = (pool_idx, x16)
        elif g.expand:
            r = hx.linear(x2, hx.w(blk.proj_max_pool.weight), blk.proj_max_pool.bias, torch.float32)
        if not g.skip_is_identity and r_full is None:
            r_full = r
            r = torch.empty(Mq, Cout, dtype=torch.float32, device=dev)
            pool_idx = torch.empty(Mq, Cout, dtype=torch.uint8, device=dev)
            _hip.check(L.mvit_maxpool_skip_fwd_idx(_hip.ptr(r_full), _hip.ptr(r), _hip.ptr(pool_idx), B, T, H, W, Cout, _st()), "maxpool")
            r_full = pool_idx        # the backward only needs the argmax positions
        y = hx.linear(o, hx.w(at.proj.weight), at.proj.bias, torch.float32, residual=r, row_scale=dp1, rps=Lq)
        vn = hx.ln_fwd(y, blk.norm2)
        if act == _hip.BF16:      # fc1 + GELU in one GEMM pass: the pre-activation (kept for the backward) and the activation
            w1 = hx.w(blk.mlp.fc1.weight)
            pre = torch.empty(Mq, w1.shape[0], dtype=adt, device=dev)
            hid = torch.empty_like(pre)
            # `pre` holds GELU'(fc1 output) when both MLP GEMMs fit the 128x192 kernels (the backward then only multiplies), else the
            # pre-activation itself
            n1, k1 = w1.shape
            mlp_dsave = (n1 % 192 == 0 and (k1 % 64 == 0 or (k1 % 64 == 32 and k1 >= 64)) and os.environ.get("MVIT_GELU_DSAVE", "1") != "0")
            fc1 = L.mvit_linear_gelu_fwd_dsave if mlp_dsave else L.mvit_linear_gelu_fwd
            _hip.check(fc1(_hip.ptr(vn), k1, _hip.ptr(w1), _hip.ptr(blk.mlp.fc1.bias), _hip.ptr(pre), _hip.ptr(hid), Mq, n1, k1, act,
                           _st()), "fc1+gelu")
        else:
            pre = hx.linear(vn, hx.w(blk.mlp.fc1.weight), blk.mlp.fc1.bias, adt)
            hid = torch.empty_like(pre)
            _hip.check(L.mvit_gelu_fwd(_hip.ptr(pre), _hip.ptr(hid), pre.numel(), act, _st()), "gelu")
        out = hx.linear(hid, hx.w(blk.mlp.fc2.weight), blk.mlp.fc2.bias, torch.float32, residual=y, row_scale=dp2, rps=Lq)
        saved = (x2, u, qkv, q, k, v, o, lse, r_full, y, vn, pre, hid, dp1, dp2)
        return out.view(B, Lq, Cout), saved, pool_saved, mlp_dsave

    @staticmethod
    def backward(ctx, d_out):
        # the deferred-reduction queue of the library is process-global state: whatever the body raises (an allocation failure, a
        # kernel error code), the queue is closed again and the pinned workspaces are dropped before the exception travels on
        hx = ctx.hx
        try:
            return _BlockFn._backward_body(ctx, d_out)
        except BaseException:
            if hx._red_keep is not None:
                try:
                    hx.L.mvit_reduce_queue_flush(_st())      # closes the queue; what it launches writes into gradients nobody will read
                finally:
                    hx._red_keep = None
            raise

    @staticmethod
    def _backward_body(ctx, d_out):
        hx, g, blk = ctx.hx, ctx.g, ctx.blk
        L, act, adt = hx.L, hx.act, hx.adt
        if ctx.recompute is not None:          # activation checkpointing: rebuild the block's activations from its input
            xin, rdp1, rdp2 = ctx.recompute
            ctx.recompute = None
            _, saved, pool_saved, _ = _BlockFn._run_forward(xin, hx, g, blk, rdp1, rdp2)
            del xin
        else:
            saved, pool_saved = ctx.saved, ctx.pool_saved
        x2, u, qkv, q, k, v, o, lse, r_full, y, vn, pre, hid, dp1, dp2 = saved
        del saved
        ctx.saved = None
        ctx.pool_saved = None
        at = blk.attn
        dev = x2.device
        B = q.shape[0]
        T, H, W = g.thw_in
        N, Lq, Lk = g.n_in, g.lq, g.lk
        M, Mq = B * N, B * Lq
        Cin, Cout, h = g.dim_in, g.dim_out, g.heads
        d_out = d_out.contiguous().view(Mq, Cout)
        # the block's eight small parameter-gradient reductions (2 LayerNorms, 3 pooling convs x {LayerNorm, conv weights}) are
        # queued by the library and go out as one launch at the end (mvit_reduce_queue_*); their partial tables stay alive in
        # hx._red_keep until then
        defer = _REDUCE_QUEUE
        if defer:
            _hip.check(L.mvit_reduce_queue_begin(), "reduce_queue_begin")
            hx._red_keep = []
        # ---- MLP branch: out = y + dp2 * (fc2(gelu(fc1(LN2(y))))) ------------------------------------------
        stash = hx._g16_stash
        hx._g16_stash = None
        if stash is not None and stash[0] == d_out.data_ptr() and stash[1].shape == d_out.shape:
            g16, gs, grps = stash[1], None, 0       # written by the downstream block's LayerNorm backward, drop-path factor applied
        else:
            g16, gs, grps = hx.scaled16(d_out, dp2, Lq)
        del stash
        dW2, db2 = hx.wgrad(hid, g16, Cout, 4 * Cout, gs, grps)
        if act == _hip.BF16:      # fc2 data gradient and the GELU backward in one GEMM pass
            w2t = hx.wt(blk.mlp.fc2.weight)
            d_pre = torch.empty_like(pre)
            fc2d = L.mvit_linear_dact_fwd if getattr(ctx, "mlp_dsave", False) else L.mvit_linear_dgelu_fwd
            _hip.check(fc2d(_hip.ptr(g16), g16.shape[1], _hip.ptr(w2t), _hip.ptr(gs), grps, _hip.ptr(pre), _hip.ptr(d_pre),
                            Mq, w2t.shape[0], w2t.shape[1], act, _st()), "fc2 dgrad + gelu_bwd")
            del g16
        else:
            d_hid = hx.linear(g16, hx.wt(blk.mlp.fc2.weight), None, adt, row_scale=gs, rps=grps)
            del g16
            d_pre = torch.empty_like(d_hid)
            _hip.check(L.mvit_gelu_bwd(_hip.ptr(pre), _hip.ptr(d_hid), _hip.ptr(d_pre), pre.numel(), act, _st()), "gelu_bwd")
            del d_hid
        dW1, db1 = hx.wgrad(vn, d_pre, 4 * Cout, Cout)
        d_vn = hx.linear(d_pre, hx.wt(blk.mlp.fc1.weight), None, adt)
        del d_pre
        d_y = torch.empty_like(d_out)
        # d_y = d_out + LN2-backward (no clone); the same kernel writes d_y * dp1 as the 16-bit operand of the proj GEMMs
        dg2, dbe2, y16 = hx.ln_bwd(y, blk.norm2, d_vn, d_y, False, base=d_out, emit16=(dp1, Lq))
        del d_vn
        # ---- attention branch: y = r + dp1 * proj(o) ------------------------------------------------------
        if y16 is not None:
            g16, gs, grps = y16, None, 0
        else:
            g16, gs, grps = hx.scaled16(d_y, dp1, Lq)
        del y16
        dWp, dbp = hx.wgrad(o, g16, Cout, Cout, gs, grps)
        d_o = hx.linear(g16, hx.wt(at.proj.weight), None, adt, row_scale=gs, rps=grps)
        del g16
        dq = torch.empty_like(q)
        dkv = torch.empty(2, B, h, Lk, 96, dtype=k.dtype, device=dev)
        dk, dv = dkv[0], dkv[1]
        ws = _ws(L.mvit_attention_bwd_workspace_bytes(B, h, Lq, Lk), dev)
        _hip.check(_hip.attention_bwd(L, q, k, v, o, lse, d_o, dq, dk, dv, ws, B, h, Lq, Lk, 96 ** -0.5, ctx.addq, act, _st()),
                   "attention_bwd")
        del d_o
        d_qkv = torch.empty(M, 3 * Cout, dtype=adt, device=dev)
        pws_bytes = max(L.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, g.stride_q[1] if g.stride_q else 1),
                        L.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, g.stride_kv[1]))
        pws = None if defer else _ws(pws_bytes, dev)      # queued reductions: one workspace per pooling conv, kept until the flush
        pool_grads = []
        kv_batch = "kv" in pool_saved
        bpools = [] if kv_batch else [(1, dk, at.pool_k, at.norm_k, g.stride_kv[1]), (2, dv, at.pool_v, at.norm_v, g.stride_kv[1])]
        if g.kernel_q:
            bpools.insert(0, (0, dq, at.pool_q, at.norm_q, g.stride_q[1]))
        else:
            _hip.check(L.mvit_head_split_bwd(_hip.ptr(dq), _hip.ptr(d_qkv), 3 * Cout, 0, B, h, N, act, _st()), "head_split_bwd")
        for which, dbuf, conv, norm, stride in bpools:
            dconv = torch.empty_like(dbuf)
            if defer:
                pws = _ws(pws_bytes, dev)
                hx._red_keep.append(pws)
            dw = hx.zeros(96, 1, 3, 3, 3)
            dgm = hx.zeros(96)
            dbt = hx.zeros(96)
            xh, rs = pool_saved.get(which, (None, None))
            _hip.check(L.mvit_pool_conv_ln_bwd_saved(_hip.ptr(qkv), 3 * Cout, which * Cout, _hip.ptr(conv.weight), _hip.ptr(norm.weight),
                                                     _hip.ptr(xh), _hip.ptr(rs), _hip.ptr(dbuf), _hip.ptr(dconv), _hip.ptr(d_qkv),
                                                     _hip.ptr(dw), _hip.ptr(dgm), _hip.ptr(dbt), 1, _hip.ptr(pws), B, h, T, H, W, stride,
                                                     norm.eps, act, _st()), "pool_bwd")
            pool_grads += [dw, dgm, dbt]
        if kv_batch:      # k and v chains as one set of launches
            xh_kv, rs_kv = pool_saved["kv"]
            dconv_kv = torch.empty_like(dkv)
            pws2 = _ws(2 * L.mvit_pool_bwd_workspace_bytes(B, h, T, H, W, g.stride_kv[1]), dev)
            if defer:
                hx._red_keep.append(pws2)
            gk = [hx.zeros(96, 1, 3, 3, 3), hx.zeros(96), hx.zeros(96)]
            gv = [hx.zeros(96, 1, 3, 3, 3), hx.zeros(96), hx.zeros(96)]
            _hip.check(L.mvit_pool_conv_ln_bwd_saved_kv(_hip.ptr(qkv), 3 * Cout, Cout, _hip.ptr(at.pool_k.weight), _hip.ptr(at.norm_k.weight),
                                                        _hip.ptr(at.pool_v.weight), _hip.ptr(at.norm_v.weight), _hip.ptr(xh_kv), _hip.ptr(rs_kv),
                                                        _hip.ptr(dkv), _hip.ptr(dconv_kv), _hip.ptr(d_qkv), _hip.ptr(gk[0]), _hip.ptr(gk[1]),
                                                        _hip.ptr(gk[2]), _hip.ptr(gv[0]), _hip.ptr(gv[1]), _hip.ptr(gv[2]), 1, _hip.ptr(pws2),
                                                        B, h, T, H, W, g.stride_kv[1], act, _st()), "pool_bwd_kv")
            pool_grads += gk + gv
        dWqkv, dbqkv = hx.wgrad(u, d_qkv, 3 * Cout, Cin)
        d_u = hx.linear(d_qkv, hx.wt(at.qkv.weight), None, adt)
        del d_qkv
        # ---- skip path ------------------------------------------------------------------------------------
        d_r = d_y
        extra = []
        if _skip_fused(g, act, B):
            # un-pool + data gradient in one kernel; the un-pooled gradient leaves once, 16 bit, for the weight-gradient GEMM
            d_x = torch.empty(M, Cin, dtype=torch.float32, device=dev)
            d16 = torch.empty(M, Cout, dtype=adt, device=dev)
            pool_idx, x16 = r_full
            _hip.check(L.mvit_proj_maxpool_bwd(_hip.ptr(pool_idx), _hip.ptr(d_y), _hip.ptr(hx.wt(blk.proj_max_pool.weight)), _hip.ptr(d_x),
                                               _hip.ptr(d16), B, T, H, W, Cin, Cout, act, _st()), "proj_maxpool_bwd")
            extra = list(hx.wgrad(x16, d16, Cout, Cin))
            del x16
            del d16
        elif r_full is not None:
            d_rf = torch.empty(M, Cout, dtype=torch.float32, device=dev)
            _hip.check(L.mvit_maxpool_skip_bwd_idx(_hip.ptr(r_full), _hip.ptr(d_r), _hip.ptr(d_rf), B, T, H, W, Cout, _st()), "maxpool_bwd")
            d_r = d_rf
        if _skip_fused(g, act, B):
            pass
        elif g.expand:
            dWm, dbm = hx.wgrad(x2, d_r, Cout, Cin)
            d_x = hx.linear(d_r, hx.wt(blk.proj_max_pool.weight), None, torch.float32)
            extra = [dWm, dbm]
        else:
            d_x = d_r
        ln1_base = None
        if act == _hip.F32 and d_x is d_y:
            # exact-fp32 path: the proj weight-gradient GEMM on the side stream reads d_y ITSELF (no 16-bit copy of it exists), so the
            # LayerNorm backward must not accumulate into it in place -- under load the side stream can still be reading when this
            # stream gets here (seen as one wrong blocks.0.attn.proj.weight gradient in ~1 of 10 runs of the two-rank test)
            ln1_base, d_x = d_y, torch.empty_like(d_y)
        if ln1_base is not None:
            dg1, dbe1 = hx.ln_bwd(x2, blk.norm1, d_u, d_x, False, base=ln1_base)
        elif ctx.prev_dp2 is not None:    # d_x is the upstream block's incoming gradient: cast it for that block's fc2 GEMMs right here
            dg1, dbe1, x16 = hx.ln_bwd(x2, blk.norm1, d_u, d_x, True, emit16=(ctx.prev_dp2[0], N))
            if x16 is not None:
                hx._g16_stash = (d_x.data_ptr(), x16)
        else:
            dg1, dbe1 = hx.ln_bwd(x2, blk.norm1, d_u, d_x, True)
        if defer:
            _hip.check(L.mvit_reduce_queue_flush(_st()), "reduce_queue_flush")
            hx._red_keep = None
        grads = [dg1, dbe1, dWqkv, dbqkv, dWp, dbp] + pool_grads + [dg2, dbe2, dW1, db1, dW2, db2] + extra
        hx.join_side()
        return (d_x.view(B, N, Cin), None, None, None, None, None) + tuple(grads)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, w, b, mask, hx):
        L = hx.L
        B, N, C = x.shape
        x = x.contiguous()
        nch = (N + 31) // 32
        ws = torch.empty(B * nch * C, dtype=torch.float32, device=x.device)
        eps = hx.m.norm.eps
        _hip.check(L.mvit_head_ln_partial(_hip.ptr(x), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(ws), B, N, C, eps, _st()), "head1")
        z = torch.empty(B, C, dtype=torch.float32, device=x.device)
        logits = torch.empty(B, w.shape[0], dtype=torch.float32, device=x.device)
        _hip.check(L.mvit_head_project_train(_hip.ptr(ws), _hip.ptr(w), _hip.ptr(b), _hip.ptr(mask), _hip.ptr(z), _hip.ptr(logits), B, N,
                                             nch, C, w.shape[0], _st()), "head2")
        ctx.hx = hx
        ctx.saved = (x, z, mask)
        return logits

    @staticmethod
    def backward(ctx, dl):
        hx = ctx.hx
        L, m = hx.L, hx.m
        x, z, mask = ctx.saved
        ctx.saved = None
        B, N, C = x.shape
        dl = dl.contiguous().float()
        ncls = dl.shape[1]
        dW = torch.empty(ncls, C, dtype=torch.float32, device=x.device)
        db = torch.empty(ncls, dtype=torch.float32, device=x.device)
        dz = torch.empty(B, C, dtype=torch.float32, device=x.device)
        _hip.check(L.mvit_head_bwd(_hip.ptr(dl), _hip.ptr(z), _hip.ptr(m.head.projection.weight), _hip.ptr(mask), _hip.ptr(dW),
                                   _hip.ptr(db), _hip.ptr(dz), B, C, ncls, 0, _st()), "head_bwd")
        dx = torch.empty(B * N, C, dtype=torch.float32, device=x.device)
        dg, dbe = hx.ln_bwd(x.view(B * N, C), m.norm, dz, dx, False, rows_per_dy=N, dy_scale=1.0 / N)
        return dx.view(B, N, C), dg, dbe, dW, db, None, None


def _draw_train_noise(model, B, C_last, dev):
    """(drop-path factors [depth,2,B] or None, head-dropout mask [B,C] or None) for a whole batch, one draw each."""
    dp_all = None
    if model.training and any(g.drop_path > 0.0 for g in model.geoms):
        # common.py:46-59 draws per DropPath call: floor(keep + U[B]) / keep
        keep_all = getattr(model, "_keep_all", None)       # device constant: a per-step H2D copy from pageable memory would block the
        if keep_all is None or keep_all.device != torch.device(dev):     # host until the previous step has drained (a full sync per step)
            keep_all = torch.tensor([1.0 - g.drop_path for g in model.geoms], device=dev, dtype=torch.float32).view(-1, 1, 1)
            model._keep_all = keep_all
        dp_all = torch.floor(keep_all + torch.rand(len(model.geoms), 2, B, device=dev, dtype=torch.float32)) / keep_all
    mask = None
    if model.training and model.head_dropout > 0.0:
        p = model.head_dropout                             # head_helper.py:410-411
        mask = (torch.rand(B, C_last, device=dev) >= p).float() / (1.0 - p)
    return dp_all, mask


def noise_from_keep(model, dp_keep, head_keep, dev):
    """The (drop-path factors, head-dropout mask) pair of ``_draw_train_noise`` from GIVEN Bernoulli outcomes instead of fresh draws:
    dp_keep [depth, 2, B] of {0,1} (the binarised floor(keep + U) of every DropPath call, common.py:46-59, in call order: attention
    branch then MLP branch of each block), head_keep [B, C] of {0,1} (the elements nn.Dropout kept, head_helper.py:410-411).  The same
    arithmetic as the drawing path: factor = kept / keep_i, mask = kept / (1 - p).  This is how a recorded step of the reference is
    replayed (tests/golden/mvit_*_stoch.npz) -- the arithmetic downstream of the draws is what parity pins."""
    dp_all = mask = None
    if dp_keep is not None:
        keep_all = torch.tensor([1.0 - g.drop_path for g in model.geoms], device=dev, dtype=torch.float32).view(-1, 1, 1)
        dp_all = torch.as_tensor(dp_keep, device=dev).to(torch.float32) / keep_all
    if head_keep is not None:
        mask = torch.as_tensor(head_keep, device=dev).to(torch.float32) / (1.0 - model.head_dropout)
    return dp_all, mask


def _forward_train_one(model, clip, hx, dp_all, mask):
    pe = model.patch_embed.proj
    x = _StemFn.apply(clip, pe.weight, pe.bias, model.pos_embed_spatial, model.pos_embed_temporal, hx)
    for i, (g, blk) in enumerate(zip(model.geoms, model.blocks)):
        dp1 = dp2 = None
        if dp_all is not None and g.drop_path > 0.0:
            dp1, dp2 = dp_all[i, 0].contiguous(), dp_all[i, 1].contiguous()
        # MODEL.ACT_CHECKPOINT is handled inside _BlockFn (input + drop-path draws kept, forward re-run in backward)
        x = _BlockFn.apply(x, hx, g, blk, dp1, dp2, *_block_params(blk, g))
    hp = model.head.projection
    return _HeadFn.apply(x, model.norm.weight, model.norm.bias, hp.weight, hp.bias, mask, hx)


def forward_train(model, clip, noise=None):
    """Forward in training mode (drop-path + head dropout active when model.training); returns raw logits.
    ``noise`` = (drop-path factors [depth, 2, B] | None, head-dropout mask [B, C] | None) replaces the fresh draws (see
    ``noise_from_keep``): used to replay a recorded stochastic step of the reference.
    Builds the autograd graph when grad mode is on.  With HIP.TRAIN_STREAMS > 1 (default 1: measured slower at B=8 @448, 78.7
    vs 72.4 ms in round 1 and 57.5 vs 49.2 ms in round 3 (profiles/r3_train_streams.txt) -- twice the launches and half-size weight-gradient GEMMs outweigh the filled tails) and >= 2 clips per stream the batch
    runs as sub-batches on side streams (forward and, through autograd's stream bookkeeping, backward): the kernels of one
    sub-batch fill the last partial wave of workgroups of the other; parameter gradients of the chains are summed by autograd."""
    hx = _Ctx(model)
    clip = clip.contiguous().float()
    B = clip.shape[0]
    dev = clip.device
    assert list(clip.shape[2:]) == model.input_dims and clip.shape[1] == 3, "clip shape %s" % (tuple(clip.shape),)
    if noise is not None:
        dp_all, mask = noise
        assert dp_all is None or tuple(dp_all.shape) == (len(model.geoms), 2, B), "drop-path factors must be [depth, 2, B]"
        assert mask is None or tuple(mask.shape) == (B, model.geoms[-1].dim_out), "head-dropout mask must be [B, C]"
    else:
        dp_all, mask = _draw_train_noise(model, B, model.geoms[-1].dim_out, dev)
    ns = model.train_streams
    if ns <= 1 or B < 2 * ns:
        return _forward_train_one(model, clip, hx, dp_all, mask)
    for m in model.modules():                      # weight copies are (re)built once, on the caller's stream
        if isinstance(m, torch.nn.Linear) and m.weight.is_cuda and m.weight.dim() == 2 and m is not model.head.projection:
            model._w_pair(m.weight, hx.act, hx.prec)
    model._side_streams = _hip.shared_streams(dev, ns)
    cur = torch.cuda.current_stream(dev)
    bounds = [(B * i) // ns for i in range(ns + 1)]
    outs = []
    for st_, b0, b1 in zip(model._side_streams, bounds[:-1], bounds[1:]):
        st_.wait_stream(cur)
        with torch.cuda.stream(st_):
            outs.append(_forward_train_one(model, clip[b0:b1], hx, None if dp_all is None else dp_all[:, :, b0:b1],
                                           None if mask is None else mask[b0:b1]))
    for st_ in model._side_streams:
        cur.wait_stream(st_)
    for o in outs:
        o.record_stream(cur)
    return torch.cat(outs, 0)


def forward_with_grad(model, clip, return_logits=False, noise=None):
    logits = forward_train(model, clip, noise)
    if return_logits:
        return torch.softmax(logits, 1), logits
    if model.training and not model.use_act_in_train:
        return logits
    return torch.softmax(logits, dim=1)      # eval-mode call with grad enabled: head act as in head_helper.py:415-416
