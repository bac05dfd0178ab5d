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
Pinned by tests/golden/train_loop.json (generated from the reference's metrics / logging / checkpoint / lr_policy modules).
The meter classes (aicity_action_amd/meters.py) keep the reference's public names and log lines only.

Host / device overlap: the reference reads three scalars back with ``.item()`` every iteration (train_net.py:290-294), which
drains the GPU queue once per step.  Here the three scalars go through ``meters.DeviceScalarQueue`` (non-blocking copy to
pinned memory + event, absorbed one iteration late; a blocking read only on ``LOG_PERIOD`` lines and at the end of the epoch),
so the loop keeps the step time ``bench.py --mode train`` measures (``bench.py --mode loop`` times exactly this function).
"""
import datetime
import decimal
import json
import logging
import math
import os
import time
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


# the statistics classes live in meters.py (own design: column ring + weighted sums, non-blocking device-scalar queue)
from .meters import DeviceScalarQueue, ScalarMeter, TestMeter, TrainMeter, ValMeter  # noqa: E402,F401


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
    """Master process only; returns the path (checkpoint.py:107-139).  A sharded optimizer (solver.HipZeroAdamW) first gathers its
    moments on the master: that is a collective, so every rank enters here (the reference's loop calls save_checkpoint on all ranks,
    train_net.py:747-759, and returns early on the others)."""
    if hasattr(optimizer, "consolidate_state_dict"):
        optimizer.consolidate_state_dict(0)
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


_GRAPHED = {}


def _graphed_step(model, optimizer, cfg, scaler):
    """The GraphedTrainStep of (model, optimizer) when cfg.HIP.GRAPH_STEP asks for it (single rank, no loss scaler), else None."""
    hip = getattr(cfg, "HIP", None)
    if hip is None or not bool(getattr(hip, "GRAPH_STEP", False)):
        return None
    if (scaler is not None and scaler.is_enabled()) or du.get_world_size() > 1:
        return None
    key = (id(model), id(optimizer))
    if key not in _GRAPHED:
        from .graph_step import GraphedTrainStep
        _GRAPHED.clear()                     # one live graph per process (its memory pool is private)
        _GRAPHED[key] = GraphedTrainStep(model, optimizer, cfg, _loss, topks_correct)
    return _GRAPHED[key]


def train_epoch(train_loader, model, optimizer, scaler, train_meter, cur_epoch, cfg):
    """One training epoch in the reference's step order (train_net.py:35-324, single-label branch).

    NaN handling: the reference raises on a NaN loss BEFORE ``backward()`` (train_net.py:221-223) at the price of a host sync
    per iteration.  Here the optimizer step itself is guarded on the device -- the fused clip + AdamW kernels skip the update
    when the loss-derived gradient norm is not finite (solver.HipAdamW.step) -- and the ``RuntimeError`` is raised when the
    iteration's scalars reach the host (at most ``HIP.STAT_QUEUE_DEPTH`` iterations later), with the parameters still clean."""
    model.train()
    train_meter.iter_tic()
    data_size = len(train_loader)
    dev = _device_of(model)
    world = du.get_world_size()
    graphed = _graphed_step(model, optimizer, cfg, scaler)
    for cur_iter, (inputs, labels, _, meta) in enumerate(train_loader):
        inputs, labels = _to_device(inputs, dev), _to_device(labels, dev)
        lr = solver.get_lr_at_epoch(cfg, cur_epoch + float(cur_iter) / data_size)
        optimizer.set_lr(lr)
        train_meter.data_toc()
        if graphed is not None:             # HIP.GRAPH_STEP: the whole step is one replayed hipGraph (graph_step.py)
            stats = graphed.run(inputs[0], labels, lr)
            train_meter.update_stats_async(stats, lr, inputs[0].size(0) * max(world, 1))
            train_meter.iter_toc()
            train_meter.log_iter_stats(cur_epoch, cur_iter)
            train_meter.iter_tic()
            continue
        preds = model(inputs)
        loss = _loss(cfg, preds, labels)
        optimizer.zero_grad()               # (train_net.py:228; HipAdamW drops the buffers: no 350 zero-fill + 350 accumulate launches)
        if scaler is not None and scaler.is_enabled():
            scaler.scale(loss).backward()
            scaler.step(optimizer)          # unscale + inf check + clip + AdamW fused in the optimizer step
            scaler.update()
        else:
            loss.backward()
            optimizer.step()                # global-norm clip (SOLVER.CLIP_GRAD_L2NORM) + AdamW, fused; skipped on a non-finite norm
        lab_idx = labels if labels.dim() == 1 else labels.argmax(1)
        n1, n5 = topks_correct(preds.detach(), lab_idx, (1, 5))
        stats = torch.stack([loss.detach().float().reshape(()), (1.0 - n1 / preds.size(0)) * 100.0,
                             (1.0 - n5 / preds.size(0)) * 100.0])                          # [loss, top1_err, top5_err]
        if world > 1:
            stats = du.all_reduce([stats])[0]                                              # one collective (train_net.py:284-287)
        train_meter.update_stats_async(stats, lr, inputs[0].size(0) * max(world, 1))       # no host sync here
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
        stats = torch.stack([(1.0 - n1 / preds.size(0)) * 100.0, (1.0 - n5 / preds.size(0)) * 100.0])
        if world > 1:
            stats = du.all_reduce([stats])[0]
        val_meter.iter_toc()
        val_meter.update_stats_async(stats, inputs[0].size(0) * max(world, 1))
        val_meter.update_predictions(preds, labels)
        val_meter.log_iter_stats(cur_epoch, cur_iter)
        val_meter.iter_tic()
    core = _unwrap(model)
    if hasattr(core, "check_finite"):
        core.check_finite()                              # fp16 guard of HIP.PRECISION auto (the epoch log below synchronises anyway)
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
        test_meter.update_stats(preds, labels, video_idx)
        test_meter.log_iter_stats(cur_iter)
        test_meter.iter_tic()
    test_meter.flush()                                   # every batch's scores are on the host (a non-finite one raises in TestMeter._apply)
    core = _unwrap(model)
    if hasattr(core, "check_finite"):
        core.check_finite()                              # the fp16 guard of HIP.PRECISION auto: raised by THIS call, not by a later forward
    return test_meter.finalize_metrics()


def train(cfg, model, train_loader, val_loader=None, optimizer=None, scaler=None):
    """Epoch driver (train_net.py:612-800): resume, per-epoch train / checkpoint / eval in the reference's order."""
    optimizer = optimizer if optimizer is not None else solver.construct_optimizer(model, cfg)
    if scaler is None:
        # train_net.py:634: GradScaler(enabled=cfg.TRAIN.MIXED_PRECISION); only the fp16 arithmetic needs the loss scale
        # (fp16 gradients underflow without one, so HIP.PRECISION fp16 always trains with the scaler -- the reference's fp16
        # recipes set TRAIN.MIXED_PRECISION True; bf16 / fp32 never need it)
        fp16 = getattr(getattr(cfg, "HIP", None), "PRECISION", "bf16") == "fp16"
        if fp16 and not cfg.TRAIN.MIXED_PRECISION:
            logger.warning("HIP.PRECISION fp16 without TRAIN.MIXED_PRECISION: enabling the loss scaler anyway")
        scaler = solver.HipGradScaler(enabled=fp16)
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
