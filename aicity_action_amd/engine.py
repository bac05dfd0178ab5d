"""Train / eval / test loops around the MI355X MViT path.

Same step order, logging lines and checkpoint files as the reference's ``tools/train_net.py`` / ``tools/test_net.py`` so the
model is a drop-in for ``tools/run_net.py``:

* ``train_epoch``  (train_net.py:35-324): per-iteration LR (``get_epoch_lr(cur_epoch + cur_iter / data_size)``, :118-120),
  forward, loss, NaN check (:221-223), ``zero_grad`` / backward / clip / step (:228-246), top-1/top-5 errors, ONE all-reduce of
  the three scalars (:284-287), ``TrainMeter`` bookkeeping and ``json_stats`` lines.
* ``eval_epoch``   (train_net.py:337-470), ``perform_test`` (test_net.py:27-170) with the view-sum ensemble of ``TestMeter``
  (meters.py:277-482).
* ``save_checkpoint`` / ``load_checkpoint`` / ``load_train_checkpoint`` (checkpoint.py:107-139,190-347,504-532): ``.pyth`` dict
  ``{"epoch", "model_state", "optimizer_state", "cfg"[, "scaler_state"]}`` under ``OUTPUT_DIR/checkpoints/
  checkpoint_epoch_%05d.pyth``, auto-resume from the lexicographically last file, shape-matched non-strict model load.

Data loaders are whatever the caller provides (the reference's dataset code is out of scope): any iterable with ``len()`` that
yields ``(inputs, labels, index, meta)`` with ``inputs`` a list of one ``[B,3,T,H,W]`` tensor, as the reference's loaders do.
Pinned by tests/golden/train_loop.json (generated from the reference's metrics / logging / checkpoint / lr_policy modules); the
meter classes follow meters.py's text (that module does not import without the dataset stack).
"""
import datetime
import decimal
import json
import logging
import math
import os
import time
from collections import deque

import numpy as np
import torch

from . import distributed as du
from . import solver

logger = logging.getLogger(__name__)


# ----------------------------------------------------------------------------------------------------------------------
# metrics / logging
# ----------------------------------------------------------------------------------------------------------------------
def topks_correct(preds, labels, ks):
    """Number of top-k correct predictions for each k (slowfast/utils/metrics.py:11-50)."""
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    _, top_max_k_inds = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)
    top_max_k_inds = top_max_k_inds.t()                         # (max_k, batch)
    rep = labels.view(1, -1).expand_as(top_max_k_inds)
    correct = top_max_k_inds.eq(rep)
    return [correct[:k, :].float().sum() for k in ks]


def json_stats_line(stats):
    """The ``json_stats: {...}`` line of slowfast/utils/logging.py:87-99: floats as 5-decimal literals, keys sorted."""
    def enc(v):
        if isinstance(v, bool):
            return json.dumps(v)
        if isinstance(v, float):
            return str(decimal.Decimal("{:.5f}".format(v)))
        if isinstance(v, dict):
            return "{" + ", ".join(json.dumps(str(k)) + ": " + enc(x) for k, x in sorted(v.items())) + "}"
        if isinstance(v, (list, tuple)):
            return "[" + ", ".join(enc(x) for x in v) + "]"
        return json.dumps(v)
    return "json_stats: " + enc(dict(stats))


def log_json_stats(stats):
    line = json_stats_line(stats)
    logger.info(line)
    return line


def gpu_mem_usage():
    """Peak GPU memory in GB (slowfast/utils/misc.py:37-46)."""
    if torch.cuda.is_available():
        return torch.cuda.max_memory_allocated() / 1024 ** 3
    return 0.0


def cpu_mem_usage():
    """(used, total) host RAM in GB (slowfast/utils/misc.py:49-60)."""
    import psutil
    vm = psutil.virtual_memory()
    return (vm.total - vm.available) / 1024 ** 3, vm.total / 1024 ** 3


class _Timer(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self._t0 = time.perf_counter()
        self._paused = None

    def pause(self):
        if self._paused is None:
            self._paused = time.perf_counter()

    def seconds(self):
        end = self._paused if self._paused is not None else time.perf_counter()
        return end - self._t0


class ScalarMeter(object):
    """Windowed scalar tracker (meters.py:482-527): median / mean of the last ``window_size`` values, global mean."""

    def __init__(self, window_size):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0

    def reset(self):
        self.deque.clear()
        self.total = 0.0
        self.count = 0

    def add_value(self, value):
        self.deque.append(value)
        self.count += 1
        self.total += value

    def get_win_median(self):
        return float(np.median(self.deque))

    def get_win_avg(self):
        return float(np.mean(self.deque))

    def get_global_avg(self):
        return self.total / self.count


class _IterTimers(object):
    def __init__(self):
        self.iter_timer, self.data_timer, self.net_timer = _Timer(), _Timer(), _Timer()

    def iter_tic(self):
        self.iter_timer.reset()
        self.data_timer.reset()

    def iter_toc(self):
        self.iter_timer.pause()
        self.net_timer.pause()

    def data_toc(self):
        self.data_timer.pause()
        self.net_timer.reset()


class TrainMeter(_IterTimers):
    """Training stats (meters.py:529-676)."""

    def __init__(self, epoch_iters, cfg):
        super().__init__()
        self._cfg = cfg
        self.epoch_iters = epoch_iters
        self.overall_iters = epoch_iters
        self.MAX_EPOCH = cfg.SOLVER.MAX_EPOCH * epoch_iters
        self.loss = ScalarMeter(cfg.LOG_PERIOD)
        self.mb_top1_err = ScalarMeter(cfg.LOG_PERIOD)
        self.mb_top5_err = ScalarMeter(cfg.LOG_PERIOD)
        self.reset()

    def reset(self):
        self.loss.reset()
        self.loss_total = 0.0
        self.lr = None
        self.mb_top1_err.reset()
        self.mb_top5_err.reset()
        self.num_top1_mis = 0
        self.num_top5_mis = 0
        self.num_samples = 0

    def update_stats(self, top1_err, top5_err, loss, lr, mb_size):
        self.loss.add_value(loss)
        self.lr = lr
        self.loss_total += loss * mb_size
        self.num_samples += mb_size
        if not self._cfg.DATA.MULTI_LABEL:
            self.mb_top1_err.add_value(top1_err)
            self.mb_top5_err.add_value(top5_err)
            self.num_top1_mis += top1_err * mb_size
            self.num_top5_mis += top5_err * mb_size

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return None
        stats = {"_type": "train_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                 "iter": "{}/{}".format(cur_iter + 1, self.epoch_iters), "loss": self.loss.get_win_median(), "lr": self.lr,
                 "gpu_mem": "{:.2f}G".format(gpu_mem_usage())}
        if not self._cfg.DATA.MULTI_LABEL:
            stats["top1_err"] = self.mb_top1_err.get_win_median()
            stats["top5_err"] = self.mb_top5_err.get_win_median()
        return log_json_stats(stats) if du.get_rank() == 0 else None

    def log_epoch_stats(self, cur_epoch):
        stats = {"_type": "train_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH), "lr": self.lr,
                 "gpu_mem": "{:.2f}G".format(gpu_mem_usage()), "RAM": "{:.2f}/{:.2f}G".format(*cpu_mem_usage()),
                 "loss": self.loss_total / self.num_samples}
        if not self._cfg.DATA.MULTI_LABEL:
            stats["top1_err"] = self.num_top1_mis / self.num_samples
            stats["top5_err"] = self.num_top5_mis / self.num_samples
        return log_json_stats(stats) if du.get_rank() == 0 else None


class ValMeter(_IterTimers):
    """Validation stats (meters.py:694-933), single-label branch."""

    def __init__(self, max_iter, cfg):
        super().__init__()
        self._cfg = cfg
        self.max_iter = max_iter
        self.overall_iters = max_iter
        self.mb_top1_err = ScalarMeter(cfg.LOG_PERIOD)
        self.mb_top5_err = ScalarMeter(cfg.LOG_PERIOD)
        self.min_top1_err = 100.0
        self.min_top5_err = 100.0
        self.reset()

    def reset(self):
        self.iter_timer.reset()
        self.mb_top1_err.reset()
        self.mb_top5_err.reset()
        self.num_top1_mis = 0
        self.num_top5_mis = 0
        self.num_samples = 0
        self.all_preds = []
        self.all_labels = []

    def update_stats(self, top1_err, top5_err, mb_size):
        self.mb_top1_err.add_value(top1_err)
        self.mb_top5_err.add_value(top5_err)
        self.num_top1_mis += top1_err * mb_size
        self.num_top5_mis += top5_err * mb_size
        self.num_samples += mb_size

    def update_predictions(self, preds, labels):
        self.all_preds.append(preds)
        self.all_labels.append(labels)

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return None
        stats = {"_type": "val_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                 "iter": "{}/{}".format(cur_iter + 1, self.max_iter), "gpu_mem": "{:.2f}G".format(gpu_mem_usage()),
                 "top1_err": self.mb_top1_err.get_win_median(), "top5_err": self.mb_top5_err.get_win_median()}
        return log_json_stats(stats) if du.get_rank() == 0 else None

    def log_epoch_stats(self, cur_epoch):
        """Logs the ``val_epoch`` line and returns the top-5 error (the reference's ``eval_result``)."""
        top1_err = self.num_top1_mis / self.num_samples
        top5_err = self.num_top5_mis / self.num_samples
        self.min_top1_err = min(self.min_top1_err, top1_err)
        self.min_top5_err = min(self.min_top5_err, top5_err)
        stats = {"_type": "val_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                 "gpu_mem": "{:.2f}G".format(gpu_mem_usage()), "RAM": "{:.2f}/{:.2f}G".format(*cpu_mem_usage()),
                 "top1_err": top1_err, "top5_err": top5_err, "min_top1_err": self.min_top1_err, "min_top5_err": self.min_top5_err}
        if du.get_rank() == 0:
            log_json_stats(stats)
        return top5_err


class TestMeter(_IterTimers):
    """Multi-view ensemble for testing (meters.py:277-482): ``num_clips`` predictions per video are summed (or max-ed)."""

    def __init__(self, num_videos, num_clips, num_cls, overall_iters, ensemble_method="sum"):
        super().__init__()
        self.num_clips = num_clips
        self.overall_iters = overall_iters
        self.ensemble_method = ensemble_method
        self.video_preds = torch.zeros((num_videos, num_cls))
        self.video_labels = torch.zeros((num_videos)).long()
        self.clip_count = torch.zeros((num_videos)).long()
        self.stats = {}

    def reset(self):
        self.clip_count.zero_()
        self.video_preds.zero_()
        self.video_labels.zero_()

    def update_stats(self, preds, labels, clip_ids):
        for ind in range(preds.shape[0]):
            vid_id = int(clip_ids[ind]) // self.num_clips
            if self.video_labels[vid_id].sum() > 0:
                assert torch.equal(self.video_labels[vid_id].float(), labels[ind].float())
            self.video_labels[vid_id] = labels[ind]
            if self.ensemble_method == "sum":
                self.video_preds[vid_id] += preds[ind]
            elif self.ensemble_method == "max":
                self.video_preds[vid_id] = torch.max(self.video_preds[vid_id], preds[ind])
            else:
                raise NotImplementedError("Ensemble Method {} is not supported".format(self.ensemble_method))
            self.clip_count[vid_id] += 1

    def log_iter_stats(self, cur_iter):
        eta_sec = self.iter_timer.seconds() * (self.overall_iters - cur_iter)
        stats = {"split": "test_iter", "cur_iter": "{}".format(cur_iter + 1), "overall_iters": self.overall_iters,
                 "eta": str(datetime.timedelta(seconds=int(eta_sec))), "time_diff": self.iter_timer.seconds()}
        return log_json_stats(stats) if du.get_rank() == 0 else None

    def finalize_metrics(self, ks=(1, 5)):
        if not all(self.clip_count == self.num_clips):
            logger.warning("clip count {} != num clips {}".format(
                ", ".join("{}: {}".format(i, k) for i, k in enumerate(self.clip_count.tolist()) if k != self.num_clips), self.num_clips))
        self.stats = {"split": "test_final"}
        num_topks_correct = topks_correct(self.video_preds, self.video_labels, ks)
        for k, x in zip(ks, num_topks_correct):
            self.stats["top{}_acc".format(k)] = "{:.{prec}f}".format(float(x / self.video_preds.size(0)) * 100.0, prec=2)
        if du.get_rank() == 0:
            log_json_stats(self.stats)
        return self.stats


# ----------------------------------------------------------------------------------------------------------------------
# checkpoints
# ----------------------------------------------------------------------------------------------------------------------
def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job, "checkpoints")


def get_path_to_checkpoint(path_to_job, epoch):
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def has_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    return any("checkpoint" in f for f in (os.listdir(d) if os.path.exists(d) else []))


def get_last_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    names = [f for f in (os.listdir(d) if os.path.exists(d) else []) if "checkpoint" in f]
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, sorted(names)[-1])


def is_checkpoint_epoch(cfg, cur_epoch):
    """checkpoint.py:84-104 without the multigrid branch (out of scope)."""
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    return (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def is_eval_epoch(cfg, cur_epoch):
    """misc.py:209-230 without the multigrid branch."""
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    return (cur_epoch + 1) % cfg.TRAIN.EVAL_PERIOD == 0


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _cfg_dump(cfg):
    return cfg.dump() if hasattr(cfg, "dump") else str(cfg)


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg, scaler=None):
    """Master process only; returns the path (checkpoint.py:107-139)."""
    if du.get_rank() != 0:
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    sd = {k: v.detach().cpu() for k, v in _unwrap(model).state_dict().items()}
    checkpoint = {"epoch": epoch, "model_state": sd, "optimizer_state": optimizer.state_dict(), "cfg": _cfg_dump(cfg)}
    if scaler is not None:
        checkpoint["scaler_state"] = scaler.state_dict()
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    with open(path, "wb") as f:
        torch.save(checkpoint, f)
    return path


def load_checkpoint(path_to_checkpoint, model, optimizer=None, scaler=None, epoch_reset=False):
    """Shape-matched non-strict load of ``model_state`` (checkpoint.py:322-347); returns the checkpoint's epoch (-1 if absent
    or reset).  Tensors whose shape differs (e.g. ``pos_embed_spatial`` between 224 and 448 crops) are skipped with a log."""
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    ms = _unwrap(model)
    model_dict = ms.state_dict()
    pre = checkpoint["model_state"]
    match = {k: v for k, v in pre.items() if k in model_dict and v.size() == model_dict[k].size()}
    for k in model_dict.keys():
        if k not in match:
            logger.info("Network weights {} not loaded.".format(k))
    ms.load_state_dict(match, strict=False)
    epoch = -1
    if "epoch" in checkpoint and not epoch_reset:
        epoch = checkpoint["epoch"]
        if optimizer is not None and "optimizer_state" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
        if scaler is not None and "scaler_state" in checkpoint:
            scaler.load_state_dict(checkpoint["scaler_state"])
    return epoch


def load_train_checkpoint(cfg, model, optimizer, scaler=None):
    """Start epoch (checkpoint.py:504-532): auto-resume from OUTPUT_DIR, else TRAIN.CHECKPOINT_FILE_PATH, else 0."""
    if cfg.TRAIN.AUTO_RESUME and has_checkpoint(cfg.OUTPUT_DIR):
        last = get_last_checkpoint(cfg.OUTPUT_DIR)
        logger.info("Load from last checkpoint, {}.".format(last))
        return load_checkpoint(last, model, optimizer, scaler) + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        logger.info("Load from given checkpoint file.")
        return load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, optimizer, scaler,
                               epoch_reset=cfg.TRAIN.CHECKPOINT_EPOCH_RESET) + 1
    return 0


# ----------------------------------------------------------------------------------------------------------------------
# loops
# ----------------------------------------------------------------------------------------------------------------------
def _to_device(x, dev):
    if isinstance(x, (list, tuple)):
        return [_to_device(v, dev) for v in x]
    if isinstance(x, dict):
        return {k: _to_device(v, dev) for k, v in x.items()}
    return x.to(dev, non_blocking=True) if torch.is_tensor(x) else x


def _device_of(model):
    return next(_unwrap(model).parameters()).device


def _loss(cfg, preds, labels):
    """losses.py:282-302 for the two functions the Aicity configs use."""
    name = cfg.MODEL.LOSS_FUNC
    if name == "soft_cross_entropy":
        if labels.dim() == 1:
            labels = torch.nn.functional.one_hot(labels, preds.shape[1]).float()
        return solver.soft_target_cross_entropy(preds, labels.float())
    if name == "cross_entropy":
        return torch.nn.functional.cross_entropy(preds, labels, reduction="mean")
    raise NotImplementedError("Loss {} is not supported".format(name))


def check_nan_losses(loss):
    if math.isnan(loss):
        raise RuntimeError("ERROR: Got NaN losses {}".format(datetime.datetime.now()))


def train_epoch(train_loader, model, optimizer, scaler, train_meter, cur_epoch, cfg):
    """One training epoch in the reference's step order (train_net.py:35-324, single-label branch)."""
    model.train()
    train_meter.iter_tic()
    data_size = len(train_loader)
    dev = _device_of(model)
    world = du.get_world_size()
    for cur_iter, (inputs, labels, _, meta) in enumerate(train_loader):
        inputs, labels = _to_device(inputs, dev), _to_device(labels, dev)
        lr = solver.get_lr_at_epoch(cfg, cur_epoch + float(cur_iter) / data_size)
        optimizer.set_lr(lr)
        train_meter.data_toc()
        preds = model(inputs)
        loss = _loss(cfg, preds, labels)
        optimizer.zero_grad()
        if scaler is not None and scaler.is_enabled():
            scaler.scale(loss).backward()
            scaler.step(optimizer)          # unscale + inf check + clip + AdamW fused in the optimizer step
            scaler.update()
        else:
            loss.backward()
            optimizer.step()                # global-norm clip (SOLVER.CLIP_GRAD_L2NORM) + AdamW, fused
        lab_idx = labels if labels.dim() == 1 else labels.argmax(1)
        n1, n5 = topks_correct(preds.detach(), lab_idx, (1, 5))
        top1_err = (1.0 - n1 / preds.size(0)) * 100.0
        top5_err = (1.0 - n5 / preds.size(0)) * 100.0
        loss_d = loss.detach()
        if world > 1:
            loss_d, top1_err, top5_err = du.all_reduce([loss_d, top1_err, top5_err])      # one collective
        loss_v, top1_v, top5_v = torch.stack([loss_d.float().reshape(()), top1_err.reshape(()), top5_err.reshape(())]).tolist()
        check_nan_losses(loss_v)            # the reference checks before backward; here the one host sync of the iteration
        train_meter.update_stats(top1_v, top5_v, loss_v, lr, inputs[0].size(0) * max(world, 1))
        train_meter.iter_toc()
        train_meter.log_iter_stats(cur_epoch, cur_iter)
        train_meter.iter_tic()
    train_meter.log_epoch_stats(cur_epoch)
    train_meter.reset()


@torch.no_grad()
def eval_epoch(val_loader, model, val_meter, cur_epoch, cfg):
    """Validation epoch (train_net.py:337-470); returns the top-5 error of the epoch."""
    model.eval()
    val_meter.iter_tic()
    dev = _device_of(model)
    world = du.get_world_size()
    for cur_iter, (inputs, labels, _, meta) in enumerate(val_loader):
        inputs, labels = _to_device(inputs, dev), _to_device(labels, dev)
        val_meter.data_toc()
        preds = model(inputs)
        lab_idx = labels if labels.dim() == 1 else labels.argmax(1)
        n1, n5 = topks_correct(preds, lab_idx, (1, 5))
        top1_err = (1.0 - n1 / preds.size(0)) * 100.0
        top5_err = (1.0 - n5 / preds.size(0)) * 100.0
        if world > 1:
            top1_err, top5_err = du.all_reduce([top1_err, top5_err])
        top1_v, top5_v = torch.stack([top1_err.reshape(()), top5_err.reshape(())]).tolist()
        val_meter.iter_toc()
        val_meter.update_stats(top1_v, top5_v, inputs[0].size(0) * max(world, 1))
        val_meter.update_predictions(preds, labels)
        val_meter.log_iter_stats(cur_epoch, cur_iter)
        val_meter.iter_tic()
    result = val_meter.log_epoch_stats(cur_epoch)
    val_meter.reset()
    return result


@torch.no_grad()
def perform_test(test_loader, model, test_meter, cfg):
    """Multi-view testing (test_net.py:27-170): every clip's softmax scores are gathered over the ranks and summed per video."""
    model.eval()
    test_meter.iter_tic()
    dev = _device_of(model)
    for cur_iter, (inputs, labels, video_idx, meta) in enumerate(test_loader):
        inputs, labels, video_idx = _to_device(inputs, dev), _to_device(labels, dev), _to_device(video_idx, dev)
        test_meter.data_toc()
        preds = model(inputs)
        if du.get_world_size() > 1:
            preds, labels, video_idx = du.all_gather_cat(preds), du.all_gather_cat(labels), du.all_gather_cat(video_idx)
        test_meter.iter_toc()
        test_meter.update_stats(preds.detach().float().cpu(), labels.detach().cpu(), video_idx.detach().cpu())
        test_meter.log_iter_stats(cur_iter)
        test_meter.iter_tic()
    return test_meter.finalize_metrics()


def train(cfg, model, train_loader, val_loader=None, optimizer=None, scaler=None):
    """Epoch driver (train_net.py:612-800): resume, per-epoch train / checkpoint / eval in the reference's order."""
    optimizer = optimizer if optimizer is not None else solver.construct_optimizer(model, cfg)
    if scaler is None:
        # train_net.py:634: GradScaler(enabled=cfg.TRAIN.MIXED_PRECISION); only the fp16 arithmetic needs the loss scale
        fp16 = getattr(getattr(cfg, "HIP", None), "PRECISION", "bf16") == "fp16"
        scaler = solver.HipGradScaler(enabled=bool(cfg.TRAIN.MIXED_PRECISION) and fp16)
    start_epoch = load_train_checkpoint(cfg, model, optimizer, scaler if scaler.is_enabled() else None)
    train_meter = TrainMeter(len(train_loader), cfg)
    val_meter = ValMeter(len(val_loader), cfg) if val_loader is not None else None
    if du.get_rank() == 0:
        os.makedirs(get_checkpoint_dir(cfg.OUTPUT_DIR), exist_ok=True)
        logger.info("Start epoch: {}".format(start_epoch + 1))
    results = []
    for cur_epoch in range(start_epoch, cfg.SOLVER.MAX_EPOCH):
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(cur_epoch)            # loader.shuffle_dataset (train_net.py:735)
        t0 = time.perf_counter()
        train_epoch(train_loader, model, optimizer, scaler, train_meter, cur_epoch, cfg)
        if du.get_rank() == 0:
            logger.info("Epoch {} takes {:.2f}s.".format(cur_epoch + 1, time.perf_counter() - t0))
        if is_checkpoint_epoch(cfg, cur_epoch):
            save_checkpoint(cfg.OUTPUT_DIR, model, optimizer, cur_epoch, cfg, scaler if scaler.is_enabled() else None)
        if val_meter is not None and is_eval_epoch(cfg, cur_epoch):
            results.append((cur_epoch, eval_epoch(val_loader, model, val_meter, cur_epoch, cfg)))
    return results
