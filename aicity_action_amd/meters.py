"""Statistics bookkeeping of the train / eval / test loops: what ``slowfast/utils/meters.py`` provides to
``tools/train_net.py`` / ``tools/test_net.py`` (TrainMeter :529-676, ValMeter :694-933, TestMeter :277-482, ScalarMeter
:482-527), re-designed for a GPU loop that must not stall on the host.

Only the public surface is kept (class names, ``iter_tic / data_toc / iter_toc``, ``update_stats``, ``log_iter_stats``,
``log_epoch_stats``, ``finalize_metrics``, ``reset``) and the ``json_stats`` lines those calls emit, which
tests/golden/train_loop.json pins byte for byte.  The internals are different:

* one ``_Window`` object per meter holds EVERY tracked series as a column of a fixed numpy ring (the last ``LOG_PERIOD`` rows)
  plus sample-weighted running sums -- no per-series deque objects;
* per-iteration scalars that live on the device go through ``DeviceScalarQueue``: each iteration issues ONE non-blocking
  device-to-pinned-host copy of the stacked scalars and an event; the host absorbs a row when its event has completed, i.e.
  normally one iteration late, and blocks only at the end of the epoch or when it is ``depth`` iterations ahead of the GPU; a
  ``LOG_PERIOD`` line whose last row is still in flight is emitted (unchanged) the moment that row lands.  The reference's ``.item()`` x3 per iteration
  (train_net.py:290-294) is a full pipeline drain per step;
* ``TestMeter`` accumulates a whole batch of clip scores with one ``index_add_`` / ``scatter_reduce_`` instead of a Python loop
  per clip.
"""
import datetime
import logging
import math
import time

import numpy as np
import torch

from . import distributed as du

logger = logging.getLogger(__name__)


def _log(stats):
    from .engine import log_json_stats
    return log_json_stats(stats) if du.get_rank() == 0 else None


def _mem_fields():
    from .engine import cpu_mem_usage, gpu_mem_usage
    return "{:.2f}G".format(gpu_mem_usage()), "{:.2f}/{:.2f}G".format(*cpu_mem_usage())


class _Window(object):
    """``width`` series side by side: ring of the last ``period`` rows + weighted running sums of all rows."""

    def __init__(self, period, width):
        self.ring = np.zeros((max(int(period), 1), width), np.float64)
        self.clear()

    def clear(self):
        self.n = 0                                   # rows pushed since clear()
        self.wsum = np.zeros(self.ring.shape[1], np.float64)
        self.weight = 0.0

    def push(self, row, weight=1.0):
        self.ring[self.n % len(self.ring)] = row
        self.n += 1
        self.wsum += np.asarray(row, np.float64) * weight
        self.weight += weight

    def recent(self):
        return self.ring[:min(self.n, len(self.ring))]

    def median(self, col):
        return float(np.median(self.recent()[:, col]))

    def mean(self, col):
        return float(self.wsum[col] / self.weight)


class ScalarMeter(object):
    """Single-series view of ``_Window`` with the reference's accessor names (meters.py:482-527)."""

    def __init__(self, window_size):
        self._w = _Window(window_size, 1)

    def reset(self):
        self._w.clear()

    def add_value(self, value):
        self._w.push([value])

    def get_win_median(self):
        return self._w.median(0)

    def get_win_avg(self):
        return float(self._w.recent()[:, 0].mean())

    def get_global_avg(self):
        return self._w.mean(0)

    @property
    def count(self):
        return self._w.n

    @property
    def total(self):
        return float(self._w.wsum[0])


class DeviceScalarQueue(object):
    """Rows of device scalars on their way to the host without a per-iteration sync.

    ``put(stacked_device_tensor, tag)`` copies the tensor into a pinned slot (non-blocking) and records an event;
    ``ready()`` yields ``(row, tag)`` for every entry whose copy has landed, oldest first; ``flush()`` waits for all of them.
    ``put`` itself waits for the entry that used the slot ``depth`` iterations ago, which bounds how far the host runs ahead.
    CPU tensors (tests, NUM_GPUS=0 host logic) pass straight through."""

    def __init__(self, width, depth=2):
        self.width, self.depth = width, max(int(depth), 1)
        self._slots = None
        self._pending = []                           # [(slot, event, tag)] oldest first
        self._done = []                              # absorbed rows waiting to be handed out
        self._next = 0

    def _alloc(self):
        self._slots = [torch.empty(self.width, dtype=torch.float64).pin_memory() for _ in range(self.depth)]

    def put(self, t, tag):
        t = t.detach().reshape(-1).to(torch.float64)
        if not t.is_cuda:
            self._done.append((t.numpy().copy(), tag))
            return
        if self._slots is None:
            self._alloc()
        if len(self._pending) == self.depth:         # the slot about to be reused: its copy must have landed
            self._retire(wait=True, count=1)
        slot = self._slots[self._next][:t.numel()]   # (rows may be shorter than the slot: the ragged last batch of a test loop)
        self._next = (self._next + 1) % self.depth
        slot.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((slot, ev, tag))

    def _retire(self, wait, count=None):
        while self._pending and (count is None or count > 0):
            slot, ev, tag = self._pending[0]
            if not ev.query():
                if not wait:
                    break
                ev.synchronize()
            self._done.append((slot.numpy().copy(), tag))
            self._pending.pop(0)
            if count is not None:
                count -= 1

    def ready(self, wait=False):
        self._retire(wait)
        out, self._done = self._done, []
        return out


class _IterClock(object):
    """The three wall-clock spans the reference's meters expose (whole iteration, data wait, network time)."""

    def __init__(self):
        self._t_iter = self._t_data_end = self._t_end = time.perf_counter()

    def iter_tic(self):
        self._t_iter = time.perf_counter()
        self._t_data_end = self._t_end = None

    def data_toc(self):
        self._t_data_end = time.perf_counter()

    def iter_toc(self):
        self._t_end = time.perf_counter()

    def iter_seconds(self):
        return (self._t_end if self._t_end is not None else time.perf_counter()) - self._t_iter

    def data_seconds(self):
        return (self._t_data_end if self._t_data_end is not None else time.perf_counter()) - self._t_iter

    def net_seconds(self):
        if self._t_data_end is None:
            return 0.0
        return (self._t_end if self._t_end is not None else time.perf_counter()) - self._t_data_end


def _check_nan(loss):
    # NaN: the reference's check (train_net.py:221-223).  An infinite loss is raised too: here the scalars reach the host up to
    # HIP.STAT_QUEUE_DEPTH iterations late, and what protects the weights meanwhile is the AdamW kernel skipping steps whose
    # gradient norm is not finite -- with inf (not NaN) the loop would otherwise keep running on skipped updates without a word
    if math.isnan(loss):
        raise RuntimeError("ERROR: Got NaN losses {}".format(datetime.datetime.now()))
    if math.isinf(loss):
        raise RuntimeError("ERROR: Got infinite losses {}".format(datetime.datetime.now()))


# column order of the train window
_LOSS, _TOP1, _TOP5 = 0, 1, 2


class TrainMeter(_IterClock):
    """Training statistics: window medians for the ``train_iter`` lines, sample-weighted epoch means for ``train_epoch``."""

    def __init__(self, epoch_iters, cfg):
        super().__init__()
        self._cfg = cfg
        self.epoch_iters = epoch_iters
        self.overall_iters = epoch_iters
        self.MAX_EPOCH = cfg.SOLVER.MAX_EPOCH * epoch_iters
        self._single = not cfg.DATA.MULTI_LABEL
        self._win = _Window(cfg.LOG_PERIOD, 3)
        self._queue = DeviceScalarQueue(3, depth=int(getattr(getattr(cfg, "HIP", None), "STAT_QUEUE_DEPTH", 2) or 2))
        self._fed = 0                                # rows handed in (absorbed or still in flight)
        self._due = []                               # log lines waiting for their last row: (rows needed, epoch, iter)
        self.lr = None

    def reset(self):
        self._absorb(wait=True)
        self._win.clear()
        self._fed, self._due = 0, []
        self.lr = None

    # -- feeding -------------------------------------------------------------------------------------------------------
    def update_stats(self, top1_err, top5_err, loss, lr, mb_size):
        """Host floats (the reference's signature)."""
        self._fed += 1
        self._push(top1_err, top5_err, loss, lr, mb_size)

    def update_stats_async(self, stats_dev, lr, mb_size):
        """``stats_dev`` = device tensor ``[loss, top1_err, top5_err]``: queued, absorbed when its copy has landed."""
        self._fed += 1
        self._queue.put(stats_dev, (lr, mb_size))
        self._absorb(wait=False)

    def _push(self, top1_err, top5_err, loss, lr, mb_size):
        _check_nan(loss)
        self.lr = lr
        self._win.push([loss, top1_err if self._single else 0.0, top5_err if self._single else 0.0], mb_size)
        while self._due and self._due[0][0] == self._win.n:      # a log line was waiting for exactly this row
            _, ep, it = self._due.pop(0)
            self._emit_iter(ep, it)

    def _absorb(self, wait):
        for row, (lr, mb) in self._queue.ready(wait):
            self._push(float(row[_TOP1]), float(row[_TOP5]), float(row[_LOSS]), lr, mb)

    # -- reporting -----------------------------------------------------------------------------------------------------
    @property
    def num_samples(self):
        return self._win.weight

    def _emit_iter(self, cur_epoch, cur_iter):
        gpu, _ = _mem_fields()
        stats = {"_type": "train_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                 "iter": "{}/{}".format(cur_iter + 1, self.epoch_iters), "loss": self._win.median(_LOSS), "lr": self.lr,
                 "gpu_mem": gpu}
        if self._single:
            stats["top1_err"] = self._win.median(_TOP1)
            stats["top5_err"] = self._win.median(_TOP5)
        return _log(stats)

    def log_iter_stats(self, cur_epoch, cur_iter):
        """The ``train_iter`` line of every ``LOG_PERIOD``-th iteration.  Returns the line when all its rows are on the host
        already; otherwise None, and the (identical) line is logged as soon as the iteration's scalars arrive."""
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return None
        self._absorb(wait=False)
        if self._win.n >= self._fed:
            return self._emit_iter(cur_epoch, cur_iter)
        self._due.append((self._fed, cur_epoch, cur_iter))
        return None

    def log_epoch_stats(self, cur_epoch):
        self._absorb(wait=True)
        gpu, ram = _mem_fields()
        stats = {"_type": "train_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH), "lr": self.lr,
                 "gpu_mem": gpu, "RAM": ram, "loss": self._win.mean(_LOSS)}
        if self._single:
            stats["top1_err"] = self._win.mean(_TOP1)
            stats["top5_err"] = self._win.mean(_TOP5)
        return _log(stats)


class ValMeter(_IterClock):
    """Validation statistics (single-label branch): window medians per ``val_iter`` line, epoch means + running minima."""

    def __init__(self, max_iter, cfg):
        super().__init__()
        self._cfg = cfg
        self.max_iter = max_iter
        self.overall_iters = max_iter
        self._win = _Window(cfg.LOG_PERIOD, 2)
        self._queue = DeviceScalarQueue(2, depth=2)
        self.min_top1_err = 100.0
        self.min_top5_err = 100.0
        self.all_preds, self.all_labels = [], []

    def reset(self):
        self._queue.ready(wait=True)
        self._win.clear()
        self.all_preds, self.all_labels = [], []

    def update_stats(self, top1_err, top5_err, mb_size):
        self._win.push([top1_err, top5_err], mb_size)

    def update_stats_async(self, stats_dev, mb_size):
        self._queue.put(stats_dev, mb_size)
        self._absorb(wait=False)

    def _absorb(self, wait):
        for row, mb in self._queue.ready(wait):
            self.update_stats(float(row[0]), float(row[1]), mb)

    def update_predictions(self, preds, labels):
        self.all_preds.append(preds)
        self.all_labels.append(labels)

    @property
    def num_samples(self):
        return self._win.weight

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return None
        self._absorb(wait=True)
        gpu, _ = _mem_fields()
        return _log({"_type": "val_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                     "iter": "{}/{}".format(cur_iter + 1, self.max_iter), "gpu_mem": gpu,
                     "top1_err": self._win.median(0), "top5_err": self._win.median(1)})

    def log_epoch_stats(self, cur_epoch):
        """Logs the ``val_epoch`` line; returns the epoch's top-5 error (what the reference's eval loop hands back)."""
        self._absorb(wait=True)
        top1, top5 = self._win.mean(0), self._win.mean(1)
        self.min_top1_err = min(self.min_top1_err, top1)
        self.min_top5_err = min(self.min_top5_err, top5)
        gpu, ram = _mem_fields()
        _log({"_type": "val_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH), "gpu_mem": gpu, "RAM": ram,
              "top1_err": top1, "top5_err": top5, "min_top1_err": self.min_top1_err, "min_top5_err": self.min_top5_err})
        return top5


class TestMeter(_IterClock):
    """Multi-view test ensemble: the ``num_clips`` score vectors of a video are summed (or max-ed) into ``video_preds``;
    clip ``i`` belongs to video ``i // num_clips`` (the loader's index layout, meters.py:354-390)."""

    def __init__(self, num_videos, num_clips, num_cls, overall_iters, ensemble_method="sum"):
        super().__init__()
        if ensemble_method not in ("sum", "max"):
            raise NotImplementedError("Ensemble Method {} is not supported".format(ensemble_method))
        self.num_clips = num_clips
        self.overall_iters = overall_iters
        self.ensemble_method = ensemble_method
        self.video_preds = torch.zeros((num_videos, num_cls))
        self.video_labels = torch.zeros((num_videos,)).long()
        self.clip_count = torch.zeros((num_videos,)).long()
        self.stats = {}
        self._queue = None

    def reset(self):
        self.flush()
        for t in (self.video_preds, self.video_labels, self.clip_count):
            t.zero_()

    def update_stats(self, preds, labels, clip_ids):
        """The reference moves the three tensors to the host here (meters.py:354-390 ``.cpu()``), a sync per iteration that lets the
        GPU run dry between forwards.  Device tensors travel through a DeviceScalarQueue instead (one packed non-blocking copy;
        absorbed into ``video_preds`` at most two iterations later, all of them by ``finalize_metrics`` / ``flush``)."""
        if not preds.is_cuda:
            return self._apply(preds.detach().float(), labels.detach().long(), clip_ids.detach().long())
        B, C = preds.shape
        row = torch.cat([preds.detach().reshape(-1).double(), labels.detach().reshape(-1).double(), clip_ids.detach().reshape(-1).double()])
        if self._queue is None or self._queue.width < row.numel():
            self.flush()
            self._queue = DeviceScalarQueue(row.numel(), depth=2)
        self._queue.put(row, (B, C))
        self._absorb(wait=False)

    def _absorb(self, wait):
        if self._queue is None:
            return
        for row, (B, C) in self._queue.ready(wait=wait):
            row = torch.from_numpy(row)
            self._apply(row[:B * C].view(B, C).float(), row[B * C:B * C + B].long(), row[B * C + B:].long())

    def flush(self):
        """Waits for every queued batch and folds it into ``video_preds`` / ``video_labels`` / ``clip_count``."""
        self._absorb(wait=True)

    def _apply(self, preds, labels, clip_ids):
        if not bool(torch.isfinite(preds).all()):       # host data by now: the check is free, and a NaN score must not be summed into a video
            raise FloatingPointError("TestMeter: non-finite clip scores (fp16 overflow under HIP.PRECISION auto? pin HIP.PRECISION bf16)")
        vid = torch.div(clip_ids, self.num_clips, rounding_mode="floor")
        seen = self.video_labels[vid] > 0                                   # a label already on file must not change
        assert torch.equal(self.video_labels[vid][seen], labels[seen]), "clips of one video disagree on its label"
        self.video_labels[vid] = labels
        if self.ensemble_method == "sum":
            self.video_preds.index_add_(0, vid, preds)
        else:
            self.video_preds.scatter_reduce_(0, vid.view(-1, 1).expand_as(preds), preds, reduce="amax", include_self=True)
        self.clip_count.index_add_(0, vid, torch.ones_like(vid))

    def log_iter_stats(self, cur_iter):
        dt = self.iter_seconds()
        eta = datetime.timedelta(seconds=int(dt * (self.overall_iters - cur_iter)))
        return _log({"split": "test_iter", "cur_iter": "{}".format(cur_iter + 1), "overall_iters": self.overall_iters,
                     "eta": str(eta), "time_diff": dt})

    def finalize_metrics(self, ks=(1, 5)):
        from .engine import topks_correct
        self.flush()
        short = [(i, int(c)) for i, c in enumerate(self.clip_count.tolist()) if c != self.num_clips]
        if short:
            logger.warning("clip count {} != num clips {}".format(", ".join("{}: {}".format(i, c) for i, c in short), self.num_clips))
        self.stats = {"split": "test_final"}
        n = self.video_preds.size(0)
        for k, hit in zip(ks, topks_correct(self.video_preds, self.video_labels, ks)):
            self.stats["top{}_acc".format(k)] = "{:.2f}".format(float(hit) / n * 100.0)
        _log(self.stats)
        return self.stats
