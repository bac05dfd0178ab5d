"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain PyTorch fp32 ops, functional, token-major) of the reference's MViTv2
forward path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package (``aicity_action_amd``) never does.

Parity status: PINNED.  ``oracle/make_golden.py`` runs this restatement next to the real
reference (imported from /root/reference in the build container) and asserts <=1e-5 on logits,
per-block activations and gradients; the fixtures it writes to ``tests/golden/`` are then
checked again by ``tests/test_oracle_golden.py`` on any box.

Each function cites the reference lines it restates (paths relative to the reference root).
Written from the math spec in SURVEY.md Appendix A -- no reference code is copied.
"""
import math
from collections import namedtuple

import torch
import torch.nn.functional as F

BlockSpec = namedtuple(
    "BlockSpec", "dim_in dim_out heads stride_q stride_kv pool_q pool_kv skip_pool drop_path")


def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:8-22."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def derive_specs(mv, depth=None):
    """Stage arithmetic of MViT.__init__ (slowfast/models/video_model_builder.py:922-1038).

    ``mv`` is a plain dict of the MVIT.* cfg keys. Only the branches the Aicity configs take
    plus their on/off switches (Q_POOL_ALL, CHANNEL_EXPAND_FRONT) are restated.
    """
    depth = depth or mv["DEPTH"]
    dim_mul = [1.0] * (depth + 1)
    head_mul = [1.0] * (depth + 1)
    for i, m in mv["DIM_MUL"]:
        dim_mul[i] = m
    for i, m in mv["HEAD_MUL"]:
        head_mul[i] = m
    kern = mv["POOL_KVQ_KERNEL"]
    pool_q = [[] for _ in range(depth)]
    stride_q = [[] for _ in range(depth)]
    for ent in mv["POOL_Q_STRIDE"]:
        stride_q[ent[0]] = list(ent[1:])
        pool_q[ent[0]] = list(kern) if kern is not None else [s + 1 if s > 1 else s for s in ent[1:]]
    if mv.get("Q_POOL_ALL", False):
        for i in range(depth):
            if not pool_q[i]:
                pool_q[i] = list(kern)
                stride_q[i] = [1, 1, 1]
    if mv.get("POOL_KV_STRIDE_ADAPTIVE") is not None:
        skv = list(mv["POOL_KV_STRIDE_ADAPTIVE"])
        kv_list = []
        for i in range(depth):
            if len(stride_q[i]) > 0:
                skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
            kv_list.append([i] + skv)
    else:
        kv_list = mv.get("POOL_KV_STRIDE") or []
    pool_kv = [[] for _ in range(depth)]
    stride_kv = [[] for _ in range(depth)]
    for ent in kv_list:
        stride_kv[ent[0]] = list(ent[1:])
        pool_kv[ent[0]] = list(kern) if kern is not None else [s + 1 if s > 1 else s for s in ent[1:]]
    dpr = [x.item() for x in torch.linspace(0, mv["DROPPATH_RATE"], depth)]
    heads = mv["NUM_HEADS"]
    embed = mv["EMBED_DIM"]
    dim_out = embed
    specs = []
    for i in range(depth):
        heads = round_width(heads, head_mul[i])
        if mv.get("CHANNEL_EXPAND_FRONT", False):
            mul = 1.0 if i == 0 else dim_mul[i - 1]
            embed = round_width(embed, mul, divisor=heads)
            dim_out = round_width(dim_out, dim_mul[i], divisor=heads)
        else:
            embed = round_width(embed, dim_mul[i], divisor=heads)
            dim_out = round_width(embed, dim_mul[i + 1], divisor=round_width(heads, head_mul[i + 1]))
        sq = stride_q[i]
        # attention.py:316-318: kernel_skip = s+1 if s>1 else s; pool_skip exists iff stride_q non-empty
        skip = None
        if len(sq) > 0:
            skip = ([s + 1 if s > 1 else s for s in sq], list(sq), [int((s + 1 if s > 1 else s) // 2) for s in sq])
        specs.append(BlockSpec(embed, dim_out, heads, tuple(sq), tuple(stride_kv[i]),
                               tuple(pool_q[i]), tuple(pool_kv[i]), skip, dpr[i]))
    return specs


def _pool_conv_ln(x, thw, w, stride, ln_w, ln_b):
    """attention_pool, conv variant (slowfast/models/attention.py:12-83) on [B,h,L,D]:
    depthwise Conv3d(k, stride, pad=k//2, groups=D, no bias) over the (T,H,W) token grid, weight
    shared by all heads, then LayerNorm(D, eps=1e-5) (attention.py:185,199,213 -> nn.LayerNorm default)."""
    B, h, L, D = x.shape
    T, H, W = thw
    k = w.shape[2:]
    t = x.reshape(B * h, T, H, W, D).permute(0, 4, 1, 2, 3)
    t = F.conv3d(t, w, None, stride=stride, padding=[kk // 2 for kk in k], groups=D)
    thw2 = list(t.shape[2:])
    t = t.reshape(B, h, D, -1).transpose(2, 3)
    t = F.layer_norm(t, (D,), ln_w, ln_b, 1e-5)
    return t, thw2


def attention(xn, thw, sd, pre, spec, taps=None):
    """MultiScaleAttention.forward (slowfast/models/attention.py:222-284)."""
    B, N, _ = xn.shape
    h, C = spec.heads, spec.dim_out
    D = C // h
    qkv = F.linear(xn, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, h, D).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    out_thw = list(thw)
    if len(spec.pool_q) > 0:
        q, out_thw = _pool_conv_ln(q, thw, sd[pre + "pool_q.weight"], spec.stride_q,
                                   sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"])
    if len(spec.pool_kv) > 0:
        k, _ = _pool_conv_ln(k, thw, sd[pre + "pool_k.weight"], spec.stride_kv,
                             sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"])
        v, _ = _pool_conv_ln(v, thw, sd[pre + "pool_v.weight"], spec.stride_kv,
                             sd[pre + "norm_v.weight"], sd[pre + "norm_v.bias"])
    scale = D ** -0.5
    attn = (q @ k.transpose(-2, -1)) * scale        # attention.py:267 (scale applied after the product)
    attn = attn.softmax(dim=-1)                     # :269
    Lq = q.shape[2]
    o = (attn @ v).transpose(1, 2).reshape(B, Lq, C)  # :276
    if spec.q_residual:
        o = o + q.transpose(1, 2).reshape(B, Lq, C)   # :277-279
    if taps is not None:
        taps.update(q=q, k=k, v=v, attn_out=o)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), out_thw  # :281


def drop_path(z, p, training, gen=None, keep_mask=None):
    """slowfast/models/common.py:46-59.  ``keep_mask`` ([B] of {0,1}): the binarised draw floor(keep + U[B]) of one DropPath call,
    injected (recovered from the reference by oracle/make_golden.py) instead of drawn here."""
    if p == 0.0 or not training:
        return z
    keep = 1 - p
    if keep_mask is not None:
        mask = keep_mask.to(z.dtype).reshape((z.shape[0],) + (1,) * (z.ndim - 1))
    else:
        mask = keep + torch.rand((z.shape[0],) + (1,) * (z.ndim - 1), dtype=z.dtype, generator=gen)
        mask.floor_()
    return z.div(keep) * mask


def block(x, thw, sd, pre, spec, training=False, taps=None, dp_keep=None):
    """MultiScaleBlock.forward (slowfast/models/attention.py:412-446), CHANNEL_EXPAND_FRONT variant.
    dp_keep ([2, B] of {0,1}, optional): the block's two DropPath draws (attention branch :434, MLP branch :445), injected."""
    B, N, _ = x.shape
    xn = F.layer_norm(x, (spec.dim_in,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)  # :421
    xb, thw_new = attention(xn, thw, sd, pre + "attn.", spec, taps)
    if spec.dim_in != spec.dim_out:            # :424-426 proj_max_pool on the UN-normed x
        x = F.linear(x, sd[pre + "proj_max_pool.weight"], sd[pre + "proj_max_pool.bias"])
    if spec.skip_pool is not None:             # :427-432 MaxPool3d on the token grid (identity when k=1,s=1)
        k, s, p = spec.skip_pool
        C = x.shape[-1]
        t = x.reshape(B, thw[0], thw[1], thw[2], C).permute(0, 4, 1, 2, 3)
        t = F.max_pool3d(t, k, s, p)
        x = t.reshape(B, C, -1).transpose(1, 2)
    x = x + drop_path(xb, spec.drop_path, training, keep_mask=None if dp_keep is None else dp_keep[0])    # :434
    xn2 = F.layer_norm(x, (spec.dim_out,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-6)  # :436
    hmid = F.gelu(F.linear(xn2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))  # common.py:27-28 (erf GELU)
    m = F.linear(hmid, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])           # common.py:31
    x = x + drop_path(m, spec.drop_path, training, keep_mask=None if dp_keep is None else dp_keep[1])     # :445
    return x, thw_new


class _Spec(object):
    """BlockSpec + the q-residual flag."""

    def __init__(self, bs, q_residual):
        self.__dict__.update(bs._asdict())
        self.q_residual = q_residual


def forward(sd, clip, mv, patch=((3, 7, 7), (2, 4, 4), (1, 3, 3)), training=False,
            head_act=True, taps=None, head_dropout=0.0, dp_keep=None, head_keep=None):
    """MViT.forward (slowfast/models/video_model_builder.py:1161-1335), non-cls, sep-pos-embed,
    TransformerBasicHead (slowfast/models/head_helper.py:409-417).

    sd: state-dict (350-key layout of SURVEY section 2.2), clip: [B,3,T,H,W] fp32, mv: MVIT cfg dict.
    Returns (output, logits): output = softmax(logits) in eval (head_act) or logits in train.
    taps (optional dict) is filled with intermediate tensors for golden fixtures.
    Injected stochastic draws (training only): dp_keep [depth, 2, B] of {0,1} = the binarised DropPath draws in call order
    (common.py:46-59; block i calls its DropPath twice), head_keep [B, C] of {0,1} = the kept elements of the head's
    nn.Dropout(head_dropout) (head_helper.py:410-411: kept elements are scaled by 1/(1-p)).
    """
    kern, stride, pad = patch
    x = F.conv3d(clip, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                 stride=stride, padding=pad)                       # stem_helper.py:335-338
    B, C0, T, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    pos = sd["pos_embed_spatial"].repeat(1, T, 1) + torch.repeat_interleave(
        sd["pos_embed_temporal"], H * W, dim=1)                    # video_model_builder.py:1206-1223
    x = x + pos
    thw = [T, H, W]
    if taps is not None:
        taps["stem"] = x
    specs = [_Spec(s, mv.get("Q_POOL_RESIDUAL", False)) for s in derive_specs(mv)]
    for i, spec in enumerate(specs):
        bt = {} if taps is not None else None
        x, thw = block(x, thw, sd, "blocks.%d." % i, spec, training, bt, None if dp_keep is None else dp_keep[i])
        if taps is not None:
            taps["block%d" % i] = x
            taps["thw%d" % i] = list(thw)
            for k_, v_ in bt.items():
                taps["block%d.%s" % (i, k_)] = v_
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-6)   # :1248-1249
    if taps is not None:
        taps["final_norm"] = x
    z = x.mean(1)                                                   # :1310
    if training and head_dropout > 0.0:
        if head_keep is not None:
            z = z * head_keep.to(z.dtype) * (1.0 / (1.0 - head_dropout))   # what F.dropout computes for a given Bernoulli mask
        else:
            z = F.dropout(z, head_dropout, True)                    # head_helper.py:410-411
    logits = F.linear(z, sd["head.projection.weight"], sd["head.projection.bias"])
    out = logits
    if head_act and not training:
        out = logits.softmax(dim=1)                                 # head_helper.py:415-416
    return out, logits


def soft_target_cross_entropy(logits, y):
    """SoftTargetCrossEntropy.forward (slowfast/models/losses.py:133-142), reduction=mean."""
    return torch.sum(-y * F.log_softmax(logits, dim=-1), dim=-1).mean()


def lr_at_epoch(sol, cur_epoch):
    """get_lr_at_epoch with the cosine policy (slowfast/utils/lr_policy.py:9-53)."""
    def cosine(e):
        off = sol["WARMUP_EPOCHS"] if sol.get("COSINE_AFTER_WARMUP", False) else 0.0
        return sol["COSINE_END_LR"] + (sol["BASE_LR"] - sol["COSINE_END_LR"]) * (
            math.cos(math.pi * (e - off) / (sol["MAX_EPOCH"] - off)) + 1.0) * 0.5
    lr = cosine(cur_epoch)
    if cur_epoch < sol["WARMUP_EPOCHS"]:
        lr_start = sol["WARMUP_START_LR"]
        lr_end = cosine(sol["WARMUP_EPOCHS"])
        alpha = (lr_end - lr_start) / sol["WARMUP_EPOCHS"]
        lr = cur_epoch * alpha + lr_start
    return lr
