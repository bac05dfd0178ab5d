"""Optimizer / LR schedule of the train step, on multi-tensor HIP kernels.

Mirrors the reference's ``construct_optimizer`` parameter grouping and AdamW hyper-parameters
(slowfast/models/optimizer.py:26-210: 1-D tensors get weight decay 0 when SOLVER.ZERO_WD_1D_PARAM, names in
``model.no_weight_decay()`` too), ``set_lr`` (:224-236), the cosine + linear warm-up policy
(slowfast/utils/lr_policy.py:9-53) and ``clip_grad_norm_(params, SOLVER.CLIP_GRAD_L2NORM)``
(tools/train_net.py:239-243).  The arithmetic is two kernels per step (global grad norm, fused AdamW) over a device
table of chunk descriptors instead of ~350 small kernels.
"""
import math

import numpy as np
import torch

from . import _hip

_CHUNK = 65536
_DT = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i4"), ("wd", "f4")])


def lr_func_cosine(cfg, cur_epoch):
    s = cfg.SOLVER
    offset = s.WARMUP_EPOCHS if s.COSINE_AFTER_WARMUP else 0.0
    assert s.COSINE_END_LR < s.BASE_LR
    return s.COSINE_END_LR + (s.BASE_LR - s.COSINE_END_LR) * (
        math.cos(math.pi * (cur_epoch - offset) / (s.MAX_EPOCH - offset)) + 1.0) * 0.5


def get_lr_at_epoch(cfg, cur_epoch):
    if cfg.SOLVER.LR_POLICY != "cosine":
        raise NotImplementedError("Unknown LR policy: {}".format(cfg.SOLVER.LR_POLICY))
    lr = lr_func_cosine(cfg, cur_epoch)
    if cur_epoch < cfg.SOLVER.WARMUP_EPOCHS:
        lr_start = cfg.SOLVER.WARMUP_START_LR
        lr_end = lr_func_cosine(cfg, cfg.SOLVER.WARMUP_EPOCHS)
        alpha = (lr_end - lr_start) / cfg.SOLVER.WARMUP_EPOCHS
        lr = cur_epoch * alpha + lr_start
    return lr


def param_groups(model, cfg):
    """(decay, no_decay) lists of (name, param) following optimizer.py:56-75."""
    mod = model.module if hasattr(model, "module") else model
    skip = mod.no_weight_decay() if hasattr(mod, "no_weight_decay") else {}
    decay, no_decay = [], []
    for name, m in mod.named_modules():
        for pname, p in m.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = (name + "." if name else "") + pname
            if name in skip:                 # the reference tests the MODULE name only (optimizer.py:69): with
                no_decay.append((full, p))   # MVIT.ZERO_DECAY_POS_CLS the root-level pos_embed_* parameters still decay there
            elif cfg.SOLVER.ZERO_WD_1D_PARAM and (p.ndim == 1 or name.endswith(".bias")):
                no_decay.append((full, p))
            else:
                decay.append((full, p))
    return decay, no_decay


class HipAdamW(object):
    """AdamW(betas (0.9, 0.999), eps 1e-8) with fused global-norm clipping, fp32 state."""

    def __init__(self, model, cfg):
        if cfg.SOLVER.OPTIMIZING_METHOD not in ("adamw", "zero_adamw"):
            raise NotImplementedError("Does not support {} optimizer".format(cfg.SOLVER.OPTIMIZING_METHOD))
        self.cfg = cfg
        self.lr = cfg.SOLVER.BASE_LR
        self.betas = (0.9, 0.999)
        self.eps = 1e-8
        self.step_count = 0
        self.groups = self._make_groups(model, cfg)
        self._owned = self._ownership()          # None: this rank updates every parameter; else the set it owns (HipZeroAdamW)
        self.state = {}
        for grp in self.groups:
            for p in grp["params"]:
                if self._owned is None or p in self._owned:
                    self.state[p] = (torch.zeros_like(p, memory_format=torch.contiguous_format),
                                     torch.zeros_like(p, memory_format=torch.contiguous_format))
        self._table = None
        self._grad_ptrs = None
        self.last_grad_norm = None

    def _make_groups(self, model, cfg):
        decay, no_decay = param_groups(model, cfg)
        return [{"params": [p for _, p in decay], "names": [n for n, _ in decay], "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
                {"params": [p for _, p in no_decay], "names": [n for n, _ in no_decay], "weight_decay": 0.0}]

    def _ownership(self):
        return None

    def _after_update(self):
        pass

    def set_lr(self, lr):
        self.lr = float(lr)

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad (default set_to_none=True as in torch >= 2.0: the gradient buffers are dropped, the next
        backward writes fresh ones instead of accumulating into zero-filled ones)."""
        for grp in self.groups:
            for p in grp["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def _build(self):
        self.table_builds = getattr(self, "table_builds", 0) + 1      # each build is an H2D copy = a host sync (see step())
        rec, rec_other, ptrs = [], [], []
        for grp in self.groups:
            for p in grp["params"]:
                if p.grad is None:            # torch.optim skips parameters that received no gradient this step
                    ptrs.append(0)
                    continue
                assert p.grad.is_contiguous() and p.is_contiguous() and p.dtype == torch.float32
                mine = self._owned is None or p in self._owned
                m, v = self.state[p] if mine else (None, None)
                n = p.numel()
                ptrs.append(p.grad.data_ptr())
                for off in range(0, n, _CHUNK):
                    (rec if mine else rec_other).append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off,
                                                         m.data_ptr() + 4 * off if mine else 0, v.data_ptr() + 4 * off if mine else 0,
                                                         min(_CHUNK, n - off), grp["weight_decay"]))
        # the chunks this rank UPDATES come first: the norm kernel walks the whole table (every rank holds every gradient after DDP's
        # all-reduce, so the global norm and the clip coefficient are the same everywhere), the AdamW kernel only the first _n_upd
        self._n_upd = len(rec)
        rec = rec + rec_other
        dev = self.groups[0]["params"][0].device
        assert _hip.lib().mvit_mt_chunk_bytes() == _DT.itemsize
        host = np.array(rec, dtype=_DT).view(np.uint8)
        # With set_to_none the gradient addresses change every step, so this runs every step: the table goes up through a small
        # ring of PINNED host buffers with a non-blocking copy (a copy from pageable memory blocks the host until the stream has
        # drained -- one full sync per training step).  The kernels that read the table are ordered behind the copy on the stream.
        if not rec:
            self._n = 0
            self._grad_ptrs = ptrs
            return
        if torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture (graph_step.GraphedTrainStep): no event waits allowed, and the copy node re-reads its host
            # source at every replay -- so the table goes up from a pinned buffer of its own that is never written again (the
            # gradients of a captured step live at fixed addresses in the graph's memory pool)
            if getattr(self, "_cap_pinned", None) is None or self._cap_pinned.numel() != host.size:
                raise RuntimeError("HipAdamW: call prepare_capture() before capturing a step (pinned memory cannot be allocated "
                                   "while a stream is capturing)")
            self._cap_pinned.numpy()[:] = host
            self._table = torch.empty(host.size, dtype=torch.uint8, device=dev)
            self._partials = torch.empty(len(rec), dtype=torch.float32, device=dev)
            self._out2 = torch.empty(2, dtype=torch.float32, device=dev)
            self._table.copy_(self._cap_pinned, non_blocking=True)
            self._pin_ring = None
            self._n = len(rec)
            self._grad_ptrs = ptrs
            return
        ring = self.__dict__.get("_pin_ring")
        if ring is None or ring[0][0].numel() != host.size or self._table is None or self._table.device != dev:
            ring = [[torch.empty(host.size, dtype=torch.uint8).pin_memory(), None] for _ in range(3)]
            self._pin_ring, self._pin_i = ring, 0
            self._table = torch.empty(host.size, dtype=torch.uint8, device=dev)
            self._partials = torch.empty(len(rec), dtype=torch.float32, device=dev)
            self._out2 = torch.empty(2, dtype=torch.float32, device=dev)
        self._pin_i = (self._pin_i + 1) % len(ring)
        buf, ev = ring[self._pin_i]
        if ev is not None:
            ev.synchronize()                                # copy issued three builds ago: long done
        buf.numpy()[:] = host
        self._table.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring[self._pin_i][1] = ev
        self._n = len(rec)
        self._grad_ptrs = ptrs

    def prepare_capture(self):
        """Allocate, OUTSIDE the capture, the pinned buffer the captured step's chunk table is uploaded from (needs one eager
        step before: the table size is the chunk count)."""
        assert self._table is not None, "run one eager step first"
        self._cap_pinned = torch.empty(self._table.numel(), dtype=torch.uint8).pin_memory()

    def hyper_values(self, step=None):
        """[lr, 1 - beta1^t, sqrt(1 - beta2^t)] of step t (default: the next one): what mvit_adamw_step_dev reads from device memory."""
        t = self.step_count + 1 if step is None else step
        # single-precision libm, exactly as mvit_adamw_step computes them on the host (so a captured step and an eager step agree
        # to the last bit)
        import ctypes
        libm = getattr(HipAdamW, "_libm", None)
        if libm is None:
            libm = ctypes.CDLL("libm.so.6")
            libm.powf.restype = ctypes.c_float
            libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
            libm.sqrtf.restype = ctypes.c_float
            libm.sqrtf.argtypes = [ctypes.c_float]
            HipAdamW._libm = libm
        f32 = lambda v: ctypes.c_float(v).value
        bc1 = f32(1.0 - libm.powf(self.betas[0], float(t)))
        bc2s = libm.sqrtf(f32(1.0 - libm.powf(self.betas[1], float(t))))
        return [self.lr, bc1, bc2s]

    def step(self, max_norm=None, grad_scale=None, hyper=None):
        """One clipped AdamW step.  max_norm None -> cfg.SOLVER.CLIP_GRAD_L2NORM (None/0 disables clipping).
        grad_scale (float): the gradients carry this loss-scale factor (HipGradScaler): they are unscaled inside the fused
        update, the clip acts on the unscaled norm, and the step is SKIPPED (returns None) when the norm is inf / nan.
        Without a scaler the same guard sits in the AdamW kernel: a non-finite global norm leaves parameters and moments
        untouched (the loop raises on the NaN / infinite loss when the iteration's scalars reach the host, at most
        HIP.STAT_QUEUE_DEPTH iterations later -- engine.train_epoch, meters._check_nan; ``step_count`` has advanced for those
        skipped steps, which is moot because the run ends there)."""
        cur = [p.grad.data_ptr() if p.grad is not None else 0 for grp in self.groups for p in grp["params"]]
        if self._table is None or cur != self._grad_ptrs:
            self._build()                                   # grads were re-allocated (e.g. set_to_none): refresh the table
        if self._n == 0:
            return None
        if max_norm is None:
            max_norm = self.cfg.SOLVER.CLIP_GRAD_L2NORM or 0.0
        L = _hip.lib()
        st = torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        if grad_scale is not None:
            _hip.check(L.mvit_grad_norm(_hip.ptr(self._table), self._n, 0.0, _hip.ptr(self._partials), _hip.ptr(self._out2), st),
                       "grad_norm")
            norm = self._out2[0] / grad_scale                       # unscaled global norm (inf / nan on overflow)
            if not bool(torch.isfinite(norm)):                      # the one host sync torch's GradScaler.step also makes
                self.step_count -= 1
                self.last_grad_norm = self._out2
                return None
            coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0) if max_norm else torch.ones_like(norm)
            self._out2[0] = norm
            self._out2[1] = coef / grad_scale                       # clip and unscale in the kernel's single gradient factor
        else:
            _hip.check(L.mvit_grad_norm(_hip.ptr(self._table), self._n, float(max_norm), _hip.ptr(self._partials),
                                        _hip.ptr(self._out2), st), "grad_norm")
        n_upd = self._n_upd
        if hyper is not None:       # per-iteration scalars from device memory (captured step)
            if n_upd:
                _hip.check(L.mvit_adamw_step_dev(_hip.ptr(self._table), n_upd, _hip.ptr(self._out2), _hip.ptr(hyper), self.betas[0],
                                                 self.betas[1], self.eps, st), "adamw")
        elif n_upd:
            _hip.check(L.mvit_adamw_step(_hip.ptr(self._table), n_upd, _hip.ptr(self._out2), self.lr, self.betas[0], self.betas[1],
                                         self.eps, self.step_count, st), "adamw")
        self._after_update()
        self.last_grad_norm = self._out2      # device tensor [norm, coef]; no host sync here
        # the kernel wrote the parameters behind torch's back: bump their version counters so every version-keyed cache (the
        # 16-bit / transposed GEMM weight copies of MViT._w and autograd._Ctx.wt) is refreshed on the next use
        bump = torch.autograd.graph.increment_version
        for grp in self.groups:
            for p in grp["params"]:
                bump(p)
        return self._out2

    def state_dict(self):
        """``torch.optim.AdamW.state_dict()`` layout (what the reference writes into ``optimizer_state`` of a .pyth checkpoint,
        slowfast/utils/checkpoint.py:127-134): parameters indexed in group order [decay..., no-decay...] -- the order
        ``construct_optimizer`` builds them in (optimizer.py:140-206) -- so the two optimizers' checkpoints interchange."""
        state, groups, idx = {}, [], 0
        for grp in self.groups:
            ids = []
            for p in grp["params"]:
                if p in self.state:          # (HipZeroAdamW: only this rank's shard, until consolidate_state_dict() has run)
                    m, v = self.state[p]
                    state[idx] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": grp["weight_decay"], "amsgrad": False,
                           "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                           "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:           # layout of the first revision of this class: {"step", "lr", "state": {name: (m, v)}}
            self.step_count = sd["step"]
            self.lr = sd["lr"]
            for grp in self.groups:
                for n, p in zip(grp["names"], grp["params"]):
                    m, v = sd["state"][n]
                    self.state[p][0].copy_(m)
                    self.state[p][1].copy_(v)
            return
        idx = 0
        # group by group, as torch.optim.Optimizer.load_state_dict checks it ("loaded state dict contains a parameter group that doesn't
        # match the size of optimizer's group"): `adamw` writes [decay..., no_decay...] in two groups, `zero_adamw` one group in
        # model.parameters() order -- a checkpoint of one must not be paired index by index with the parameters of the other
        have, want = [len(g["params"]) for g in sd["param_groups"]], [len(g["params"]) for g in self.groups]
        if have != want:
            raise ValueError("loaded optimizer state has parameter groups of sizes %s, this optimizer (SOLVER.OPTIMIZING_METHOD %s) has %s"
                             % (have, self.cfg.SOLVER.OPTIMIZING_METHOD, want))
        self.lr = float(sd["param_groups"][0]["lr"])
        for grp in self.groups:
            for n, p in zip(grp["names"], grp["params"]):
                ent = sd["state"].get(idx)
                if ent is not None and p in self.state:     # torch omits entries of parameters that never received a gradient
                    for key in ("exp_avg", "exp_avg_sq"):
                        if tuple(ent[key].shape) != tuple(p.shape):      # Tensor.copy_ would broadcast a [C] moment into a [N, C] slot
                            raise ValueError("optimizer state %d (%s): %s has shape %s, the parameter %s" % (
                                idx, n, key, tuple(ent[key].shape), tuple(p.shape)))
                    self.state[p][0].copy_(ent["exp_avg"])
                    self.state[p][1].copy_(ent["exp_avg_sq"])
                    self.step_count = int(float(ent["step"]))
                idx += 1


class HipGradScaler(object):
    """Dynamic loss scaling for fp16 training, the role ``torch.cuda.amp.GradScaler`` plays in the reference's loop
    (tools/train_net.py:126,231-246): ``scale(loss).backward(); step(optimizer); update()``.  Unscaling, the inf / nan check and
    the global-norm clip are folded into ``HipAdamW.step(grad_scale=...)``; ``state_dict`` uses GradScaler's keys so the
    ``scaler_state`` entry of a .pyth checkpoint interchanges."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = bool(enabled)
        self._scale = float(init_scale)
        self._growth_factor, self._backoff_factor, self._growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._growth_tracker = 0
        self._skipped = False

    def is_enabled(self):
        return self._enabled

    def get_scale(self):
        return self._scale if self._enabled else 1.0

    def scale(self, loss):
        return loss * self._scale if self._enabled else loss

    def step(self, optimizer, max_norm=None):
        if not self._enabled:
            return optimizer.step(max_norm)
        out = optimizer.step(max_norm, grad_scale=self._scale)
        self._skipped = out is None
        return out

    def update(self):
        if not self._enabled:
            return
        if self._skipped:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0
        self._skipped = False

    def state_dict(self):
        if not self._enabled:
            return {}
        return {"scale": self._scale, "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, sd):
        if not self._enabled or not sd:
            return
        self._scale = float(sd["scale"])
        self._growth_factor, self._backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self._growth_interval, self._growth_tracker = int(sd["growth_interval"]), int(sd["_growth_tracker"])


class HipZeroAdamW(HipAdamW):
    """``SOLVER.OPTIMIZING_METHOD zero_adamw`` (slowfast/models/optimizer.py:189-199: ``ZeroRedundancyOptimizer(model.parameters(),
    optimizer_class=AdamW, ...)``): ZeRO stage 1.  Every rank keeps the AdamW moments of ITS shard of the parameters only (greedy
    partition by size, as torch's ZeroRedundancyOptimizer does), updates that shard with the same fused kernels, and the owners
    broadcast their updated parameters (one flat buffer per owner).  Gradients are complete on every rank (DDP all-reduce), so the
    global-norm clip needs no extra collective.  As in the reference, this path hands ``model.parameters()`` to the optimizer in ONE
    group: the weight decay applies to every parameter, 1-D ones included (optimizer.py:190-191 prints exactly that warning).
    ``state_dict()`` holds this rank's shard until ``consolidate_state_dict()`` (a collective: every rank calls it) has gathered the
    rest on rank ``to``; ``engine.save_checkpoint`` calls it."""

    def _make_groups(self, model, cfg):
        mod = model.module if hasattr(model, "module") else model
        named = [(n, p) for n, p in mod.named_parameters() if p.requires_grad]
        if cfg.SOLVER.ZERO_WD_1D_PARAM:
            import logging
            logging.getLogger(__name__).warning("warning, not setting zero parameters weight decay to zero")     # optimizer.py:190-191
        return [{"params": [p for _, p in named], "names": [n for n, _ in named], "weight_decay": cfg.SOLVER.WEIGHT_DECAY}]

    def _ownership(self):
        from . import distributed as du
        self.rank, self.world = du.get_rank(), du.get_world_size()
        params = [p for g in self.groups for p in g["params"]]
        load = [0] * self.world
        self.owner = {}
        for i in sorted(range(len(params)), key=lambda i: (-params[i].numel(), i)):     # largest first onto the least loaded rank
            r = min(range(self.world), key=lambda r_: (load[r_], r_))
            self.owner[params[i]] = r
            load[r] += params[i].numel()
        self.shards = [[p for p in params if self.owner[p] == r] for r in range(self.world)]
        return set(self.shards[self.rank])

    def _after_update(self):
        """The owners hand out their updated shards: one persistent flat buffer per owner (allocated once), the owner packs its
        shard into it (one pass over 1 / world of the parameters), all broadcasts are issued asynchronously and waited for once,
        the other ranks copy the received values into their parameters (one multi-tensor copy per owner)."""
        if self.world == 1:
            return
        import torch.distributed as dist
        if getattr(self, "_flat", None) is None:
            self._flat = [None if not shard else torch.empty(sum(p.numel() for p in shard), dtype=shard[0].dtype, device=shard[0].device)
                          for shard in self.shards]
        works = []
        for r, shard in enumerate(self.shards):
            if not shard:
                continue
            if r == self.rank:
                torch.cat([p.data.reshape(-1) for p in shard], out=self._flat[r])
            works.append(dist.broadcast(self._flat[r], src=r, async_op=True))
        for w in works:
            w.wait()
        for r, shard in enumerate(self.shards):
            if shard and r != self.rank:
                views = [v.view_as(p) for v, p in zip(self._flat[r].split([p.numel() for p in shard]), shard)]
                torch._foreach_copy_([p.data for p in shard], views)

    def consolidate_state_dict(self, to=0):
        """Gathers every shard's moments on rank ``to`` (collective: every rank calls it).  The gathered moments live in a temporary
        table that the NEXT ``state_dict()`` on that rank consumes (full ``torch.optim.AdamW`` layout) and drops: rank ``to`` goes
        back to holding its own shard only, and a later ``state_dict()`` without a fresh consolidation is the shard again, never a
        stale copy of the other ranks' moments under the current step count."""
        if self.world == 1:
            return
        import torch.distributed as dist
        gathered = {}
        for r, shard in enumerate(self.shards):
            for p in shard:
                if r == to:
                    continue
                if self.rank == r:
                    dist.send(torch.stack(list(self.state[p])), dst=to)
                elif self.rank == to:
                    buf = torch.empty((2,) + tuple(p.shape), dtype=p.dtype, device=p.device)
                    dist.recv(buf, src=r)
                    gathered[p] = (buf[0], buf[1])
        if self.rank == to:
            self._gathered = gathered

    def state_dict(self):
        gathered = getattr(self, "_gathered", None)
        if not gathered:
            return super().state_dict()                 # this rank's shard
        own = self.state
        try:
            self.state = dict(own)
            self.state.update(gathered)
            return super().state_dict()
        finally:
            self.state = own
            self._gathered = None                       # one-shot: the other ranks' moments are not kept


def construct_optimizer(model, cfg):
    """slowfast/models/optimizer.py:14-206 for the methods the MViT recipes use: ``adamw`` (the README recipe; parameter groups of
    :56-75) and ``zero_adamw`` (:189-199); anything else raises NotImplementedError as the reference does for unknown names."""
    if cfg.SOLVER.OPTIMIZING_METHOD == "zero_adamw":
        return HipZeroAdamW(model, cfg)
    return HipAdamW(model, cfg)


def soft_target_cross_entropy(logits, labels):
    """SoftTargetCrossEntropy(reduction='mean') (slowfast/models/losses.py:133-142) with a fused HIP fwd+bwd kernel."""
    return _SoftCE.apply(logits, labels)


class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        L = _hip.lib()
        logits = logits.contiguous().float()
        labels = labels.contiguous().float()
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        _hip.check(L.mvit_soft_ce(_hip.ptr(logits), _hip.ptr(labels), _hip.ptr(loss), _hip.ptr(dl), B, C, 1.0,
                                  torch.cuda.current_stream().cuda_stream), "soft_ce")
        ctx.save_for_backward(dl)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None
