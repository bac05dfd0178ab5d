"""A whole training step as ONE replayed hipGraph.

The reference's loop (tools/train_net.py:83-324) enqueues a step kernel by kernel from Python; on this path that is ~720
launches + ~130 stream / event calls per step = ~27 ms of host work next to ~55 ms of GPU time, per process, eight processes
per node.  Every shape of the step is fixed by the cfg, so after two eager iterations the step

    forward (drop-path / head-dropout draws included) -> soft-target CE / CE -> zero_grad -> backward (side streams and all) ->
    global-norm clip + AdamW -> [loss, top1_err, top5_err]

is captured once (``torch.cuda.graph``: HIP stream capture; the C-ABI kernels are ordinary launches on the capturing stream, the
library's side stream and the weight-gradient stream join the capture through their fork / join events) and replayed with
one ``hipGraphLaunch`` per iteration.  What changes between iterations lives in device memory that the host refreshes before
the replay: the clip and labels (copied into static tensors) and the optimizer's ``[lr, 1 - beta1^t, sqrt(1 - beta2^t)]``
(``mvit_adamw_step_dev``); the RNG offsets of the captured draws are advanced by torch's graph-safe generator.

``GraphedTrainStep.run(inputs, labels, lr)`` has the semantics of the eager sequence in ``engine.train_epoch`` and returns the
device tensor ``[loss, top1_err, top5_err]``; batches whose shape differs from the captured one (a ragged last batch) and
multi-rank runs (DistributedDataParallel hooks are not captured here) take the eager path.
"""
import torch

from . import solver


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


class GraphedTrainStep(object):
    def __init__(self, model, optimizer, cfg, loss_fn, topk_fn, warmup=2):
        self.model, self.opt, self.cfg = model, optimizer, cfg
        self.loss_fn, self.topk_fn = loss_fn, topk_fn
        self.warmup = max(int(warmup), 1)        # >= 1: the capture must see parameters whose 16-bit copies are stale
        self.seen = 0
        self.graph = None
        self.static_clip = self.static_labels = self.static_stats = None
        self.hyper = None                         # device [lr, bc1, bc2_sqrt]
        self._hyper_pin = None
        self._hyper_ev = None
        self.replays = 0

    # -- the step, eager ---------------------------------------------------------------------------------------------------
    def _step(self, clip, labels, hyper=None):
        preds = self.model([clip])
        loss = self.loss_fn(self.cfg, preds, labels)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step(hyper=hyper)
        lab_idx = labels if labels.dim() == 1 else labels.argmax(1)
        n1, n5 = self.topk_fn(preds.detach(), lab_idx, (1, 5))
        return torch.stack([loss.detach().float().reshape(()), (1.0 - n1 / preds.size(0)) * 100.0, (1.0 - n5 / preds.size(0)) * 100.0])

    def _capturable(self, clip, labels):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        if self.static_clip is None:
            return True
        return clip.shape == self.static_clip.shape and labels.shape == self.static_labels.shape and labels.dtype == self.static_labels.dtype

    def _set_hyper(self):
        vals = self.opt.hyper_values(step=self.opt.step_count)
        if self._hyper_pin is None:
            # a ring of pinned triples: the host may be a few iterations ahead of the stream (engine.train_epoch's statistics
            # queue bounds that to HIP.STAT_QUEUE_DEPTH), and a slot must not be rewritten before its copy has executed
            self._hyper_pin = [torch.empty(3, dtype=torch.float32).pin_memory() for _ in range(8)]
            self._hyper_ev = [None] * 8
        i = self.replays % 8
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()      # eight iterations old: long done
        buf = self._hyper_pin[i]
        buf[0], buf[1], buf[2] = vals
        self.hyper.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[i] = ev

    def _capture(self, clip, labels):
        dev = clip.device
        self.static_clip = clip.detach().clone()
        self.static_labels = labels.detach().clone()
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self.opt.prepare_capture()
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        count0 = self.opt.step_count
        with torch.cuda.graph(self.graph):
            self.static_stats = self._step(self.static_clip, self.static_labels, hyper=self.hyper)
        self.opt.step_count = count0              # nothing has executed yet: the capture only recorded the launches

    def run(self, clip, labels, lr):
        """One train step on (clip, labels) at learning rate lr; returns the device tensor [loss, top1_err, top5_err]."""
        self.opt.set_lr(lr)
        self.seen += 1
        if not self._capturable(clip, labels) or self.seen <= self.warmup:
            return self._step(clip, labels)
        if self.graph is None:
            self._capture(clip, labels)
        self.static_clip.copy_(clip, non_blocking=True)
        self.static_labels.copy_(labels, non_blocking=True)
        self.opt.step_count += 1
        self._set_hyper()
        self.graph.replay()
        self.replays += 1
        # the fused optimizer wrote the parameters behind torch's back inside the graph: keep the version counters moving for
        # anything outside the graph that keys a cache on them (eval between epochs)
        bump = torch.autograd.graph.increment_version
        for grp in self.opt.groups:
            for p in grp["params"]:
                bump(p)
        return self.static_stats.clone()
