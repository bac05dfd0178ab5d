"""Per-block geometry of an MViTv2 model, derived from cfg.

Restates the stage arithmetic of ``MViT.__init__`` (reference
slowfast/models/video_model_builder.py:922-1038) and ``round_width``
(slowfast/models/utils.py:8-22) as a pure function cfg -> list of ``BlockGeom`` so both the
``nn.Module`` (parameter shapes) and the HIP forward plan (launch geometry) read one table.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple


def round_width(width, multiplier, min_width=1, divisor=1):
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    rounded = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if rounded < 0.9 * width:
        rounded += divisor
    return int(rounded)


@dataclass
class BlockGeom:
    index: int
    dim_in: int
    dim_out: int
    heads: int
    kernel_q: Tuple[int, ...]      # () = no q pooling module
    stride_q: Tuple[int, ...]
    kernel_kv: Tuple[int, ...]
    stride_kv: Tuple[int, ...]
    skip_kernel: Optional[Tuple[int, int, int]]   # MaxPool3d on the skip path (None = absent)
    skip_stride: Optional[Tuple[int, int, int]]
    skip_pad: Optional[Tuple[int, int, int]]
    drop_path: float
    thw_in: Tuple[int, int, int] = (0, 0, 0)
    thw_q: Tuple[int, int, int] = (0, 0, 0)
    thw_kv: Tuple[int, int, int] = (0, 0, 0)

    @property
    def head_dim(self):
        return self.dim_out // self.heads

    @property
    def expand(self):
        return self.dim_in != self.dim_out

    @property
    def skip_is_identity(self):
        return self.skip_kernel is None or all(s == 1 for s in self.skip_stride)

    @property
    def n_in(self):
        return self.thw_in[0] * self.thw_in[1] * self.thw_in[2]

    @property
    def lq(self):
        return self.thw_q[0] * self.thw_q[1] * self.thw_q[2]

    @property
    def lk(self):
        return self.thw_kv[0] * self.thw_kv[1] * self.thw_kv[2]


def _conv_out(n, k, s):
    return (n + 2 * (k // 2) - k) // s + 1


def derive_block_geoms(cfg) -> List[BlockGeom]:
    mv = cfg.MVIT
    depth = mv.DEPTH
    dim_mul = [1.0] * (depth + 1)
    head_mul = [1.0] * (depth + 1)
    for idx, m in mv.DIM_MUL:
        dim_mul[idx] = m
    for idx, m in mv.HEAD_MUL:
        head_mul[idx] = m

    def adaptive_kernel(strides):
        return [s + 1 if s > 1 else s for s in strides]

    kq = [[] for _ in range(depth)]
    sq = [[] for _ in range(depth)]
    for ent in mv.POOL_Q_STRIDE:
        sq[ent[0]] = list(ent[1:])
        kq[ent[0]] = list(mv.POOL_KVQ_KERNEL) if mv.POOL_KVQ_KERNEL is not None else adaptive_kernel(ent[1:])
    if mv.Q_POOL_ALL:
        for i in range(depth):
            if not kq[i]:
                kq[i] = list(mv.POOL_KVQ_KERNEL)
                sq[i] = [1, 1, 1]

    if mv.POOL_KV_STRIDE_ADAPTIVE is not None:
        # NOTE: the reference also writes this list back into cfg.MVIT.POOL_KV_STRIDE
        # (video_model_builder.py:960-967); MViT.__init__ keeps that side effect.
        cur = list(mv.POOL_KV_STRIDE_ADAPTIVE)
        kv_entries = []
        for i in range(depth):
            if sq[i]:
                cur = [max(cur[d] // sq[i][d], 1) for d in range(len(cur))]
            kv_entries.append([i] + cur)
    else:
        kv_entries = mv.POOL_KV_STRIDE or []
    kkv = [[] for _ in range(depth)]
    skv = [[] for _ in range(depth)]
    for ent in kv_entries:
        skv[ent[0]] = list(ent[1:])
        kkv[ent[0]] = list(mv.POOL_KVQ_KERNEL) if mv.POOL_KVQ_KERNEL is not None else adaptive_kernel(ent[1:])

    # torch.linspace(0, rate, depth) in fp32 (video_model_builder.py:880-882)
    import torch
    dpr = [v.item() for v in torch.linspace(0, mv.DROPPATH_RATE, depth)]

    stride = list(mv.PATCH_STRIDE)
    if mv.PATCH_2D:
        stride = [1] + stride
    dims_in = [cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.TRAIN_CROP_SIZE]
    thw = tuple(dims_in[i] // stride[i] for i in range(3))

    heads = mv.NUM_HEADS
    embed = mv.EMBED_DIM
    dim_out = embed
    geoms = []
    for i in range(depth):
        heads = round_width(heads, head_mul[i])
        if mv.CHANNEL_EXPAND_FRONT:
            embed = round_width(embed, 1.0 if i == 0 else dim_mul[i - 1], divisor=heads)
            dim_out = round_width(dim_out, dim_mul[i], divisor=heads)
        else:
            embed = round_width(embed, dim_mul[i], divisor=heads)
            dim_out = round_width(embed, dim_mul[i + 1], divisor=round_width(heads, head_mul[i + 1]))
        # attention.py:131-134 -- a (1,1,1)/(1,1,1) pool is dropped
        def live(k, s):
            if not k:
                return (), ()
            if all(v == 1 for v in k) and all(v == 1 for v in s):
                return (), ()
            return tuple(k), tuple(s)
        kq_i, sq_i = live(kq[i], sq[i])
        kkv_i, skv_i = live(kkv[i], skv[i])
        # attention.py:316-318,389-395 (pool_skip built from the *configured* stride_q, even 1,1,1)
        if len(sq[i]) > 0:
            sk = tuple(s + 1 if s > 1 else s for s in sq[i])
            ss = tuple(sq[i])
            sp = tuple(int(k // 2) for k in sk)
        else:
            sk = ss = sp = None
        g = BlockGeom(i, embed, dim_out, heads, kq_i, sq_i, kkv_i, skv_i, sk, ss, sp, dpr[i])
        g.thw_in = thw
        g.thw_q = tuple(_conv_out(thw[d], kq_i[d], sq_i[d]) for d in range(3)) if kq_i else thw
        g.thw_kv = tuple(_conv_out(thw[d], kkv_i[d], skv_i[d]) for d in range(3)) if kkv_i else thw
        geoms.append(g)
        thw = g.thw_q
    return geoms, kv_entries
