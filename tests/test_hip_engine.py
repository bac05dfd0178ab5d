"""GPU: the train / eval / test loops of aicity_action_amd/engine.py driving the HIP model end to end (tiny config):
epoch driver with checkpoint + auto-resume reproduces the uninterrupted run, json_stats lines, view-sum test ensemble."""
import json
import logging

import pytest
import torch

from conftest import cfg_for_case, load_golden

from aicity_action_amd import engine
from aicity_action_amd.models import build_model
from aicity_action_amd.solver import construct_optimizer
from aicity_action_amd.utils.synth import load_synth_weights, synth_clip

pytestmark = pytest.mark.gpu


class _Loader(list):
    """Stand-in for the reference's DataLoader: len() + iteration over (inputs, labels, index, meta)."""


def _batches(meta, n, seed0, ncls):
    out = _Loader()
    for i in range(n):
        clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], seed0 + i)
        labels = torch.tensor([(3 * i + j) % ncls for j in range(meta["batch"])])
        out.append(([clip], labels, torch.arange(meta["batch"]) + i * meta["batch"], {}))
    return out


def _make(meta, precision, outdir):
    cfg = cfg_for_case(meta, precision, train=True)
    cfg.NUM_GPUS = 1
    cfg.MVIT.DROPPATH_RATE = 0.0            # deterministic runs: no drop-path / dropout draws
    cfg.MODEL.DROPOUT_RATE = 0.0
    cfg.SOLVER.MAX_EPOCH, cfg.SOLVER.WARMUP_EPOCHS = 2, 1.0
    cfg.TRAIN.CHECKPOINT_PERIOD, cfg.TRAIN.EVAL_PERIOD, cfg.LOG_PERIOD = 1, 1, 1
    cfg.OUTPUT_DIR = outdir
    cfg.TRAIN.AUTO_RESUME = True
    model = build_model(cfg)
    load_synth_weights(model, meta["weight_seed"])
    return cfg, model


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_driver_checkpoints_and_resumes(tmp_path, precision, caplog):
    _, meta = load_golden("tiny_even")
    ncls = meta.get("num_classes", 18)
    train_loader = _batches(meta, 3, 100, ncls)
    val_loader = _batches(meta, 2, 200, ncls)
    # (a) uninterrupted: 2 epochs
    cfg_a, model_a = _make(meta, precision, str(tmp_path / "a"))
    with caplog.at_level(logging.INFO, logger="aicity_action_amd.engine"):
        res = engine.train(cfg_a, model_a, train_loader, val_loader)
    assert [e for e, _ in res] == [0, 1] and all(0.0 <= r <= 100.0 for _, r in res)
    lines = [json.loads(r.message.split("json_stats: ")[1]) for r in caplog.records if r.message.startswith("json_stats: ")]
    kinds = [l["_type"] for l in lines]
    assert kinds == ["train_iter"] * 3 + ["train_epoch"] + ["val_iter"] * 2 + ["val_epoch"] + ["train_iter"] * 3 + ["train_epoch"] + \
        ["val_iter"] * 2 + ["val_epoch"]
    assert lines[0]["epoch"] == "1/2" and lines[0]["iter"] == "1/3" and lines[4]["iter"] == "1/2"
    assert lines[0]["lr"] == pytest.approx(cfg_a.SOLVER.WARMUP_START_LR, abs=1e-5)
    assert engine.has_checkpoint(cfg_a.OUTPUT_DIR) and engine.get_last_checkpoint(cfg_a.OUTPUT_DIR).endswith("checkpoint_epoch_00002.pyth")
    # (b) one epoch, checkpoint, then a fresh model whose state must come entirely from the file
    cfg_b, model_b = _make(meta, precision, str(tmp_path / "b"))
    opt_b = construct_optimizer(model_b, cfg_b)
    engine.train_epoch(train_loader, model_b, opt_b, None, engine.TrainMeter(len(train_loader), cfg_b), 0, cfg_b)
    engine.save_checkpoint(cfg_b.OUTPUT_DIR, model_b, opt_b, 0, cfg_b)
    cfg_c, model_c = _make(meta, precision, str(tmp_path / "b"))
    for p in model_c.parameters():
        p.data.zero_()
    opt_c = construct_optimizer(model_c, cfg_c)
    assert engine.load_train_checkpoint(cfg_c, model_c, opt_c) == 1
    ck = torch.load(engine.get_last_checkpoint(cfg_c.OUTPUT_DIR), map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == ["cfg", "epoch", "model_state", "optimizer_state"] and ck["epoch"] == 0
    assert len(ck["model_state"]) == 350 or len(ck["model_state"]) == len(model_c.state_dict())
    assert sorted(ck["optimizer_state"].keys()) == ["param_groups", "state"] and len(ck["optimizer_state"]["param_groups"]) == 2
    for (k, a), b in zip(model_b.state_dict().items(), model_c.state_dict().values()):
        assert torch.equal(a, b), k


def test_resume_reproduces_uninterrupted_run(tmp_path):
    _, meta = load_golden("tiny_even")
    train_loader = _batches(meta, 3, 100, 18)
    cfg_a, model_a = _make(meta, "fp32", str(tmp_path / "a"))
    engine.train(cfg_a, model_a, train_loader, None)
    # same schedule (MAX_EPOCH 2), stopped after epoch 0 by deleting the later checkpoint and resuming in a fresh model
    cfg_b, model_b = _make(meta, "fp32", str(tmp_path / "b"))
    opt_b = construct_optimizer(model_b, cfg_b)
    tm = engine.TrainMeter(len(train_loader), cfg_b)
    engine.train_epoch(train_loader, model_b, opt_b, None, tm, 0, cfg_b)
    engine.save_checkpoint(cfg_b.OUTPUT_DIR, model_b, opt_b, 0, cfg_b)
    cfg_c, model_c = _make(meta, "fp32", str(tmp_path / "b"))
    engine.train(cfg_c, model_c, train_loader, None)            # auto-resumes at epoch 1
    # Adam turns the rounding noise of (mathematically) zero gradients -- e.g. key biases, which softmax cancels -- into +-lr
    # steps, and the fp32 atomics make that noise run-dependent: compare what the model computes, and the bulk of the weights
    big, tot = 0, 0
    for (k, a), b in zip(model_a.state_dict().items(), model_c.state_dict().values()):
        big += int(((a - b).abs() > 1e-4).sum())
        tot += a.numel()
    assert big <= 0.01 * tot, (big, tot)
    probe = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], 999).cuda()
    with torch.no_grad():
        pa, pc = model_a.eval()([probe]), model_c.eval()([probe])
    assert (pa - pc).abs().max().item() <= 2e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_weight_copies_follow_the_fused_optimizer(tmp_path, precision):
    """The AdamW kernel writes parameters through raw pointers; the cached 16-bit / transposed GEMM weight copies must be
    rebuilt afterwards (regression: they were keyed on a version counter the kernel did not bump)."""
    _, meta = load_golden("tiny_even")
    loader = _batches(meta, 2, 100, 18)
    cfg, model = _make(meta, precision, str(tmp_path))
    opt = construct_optimizer(model, cfg)
    engine.train_epoch(loader, model, opt, None, engine.TrainMeter(len(loader), cfg), 0, cfg)
    cfg2, fresh = _make(meta, precision, str(tmp_path))
    fresh.load_state_dict(model.state_dict())
    clip = loader[0][0][0].cuda()
    labels = loader[0][1].cuda()
    outs = []
    for m in (model, fresh):
        m.train()
        logits = m([clip])
        loss = engine._loss(cfg, logits, labels)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        outs.append((logits.detach(), torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()))
    assert torch.equal(outs[0][0], outs[1][0])                              # forward sees the stepped weights
    assert abs(outs[0][1] - outs[1][1]) <= 1e-5 * outs[1][1]                # and so does the backward (transposed copies)


def test_perform_test_view_sum(tmp_path):
    _, meta = load_golden("tiny_even")
    cfg, model = _make(meta, "bf16", str(tmp_path))
    B = meta["batch"]
    # 2 "videos" x B clips each (num_clips = B): the loader yields one video's clips per iteration
    loader = _Loader()
    for v in range(2):
        clip = synth_clip(B, meta["num_frames"], meta["crop"], 300 + v)
        loader.append(([clip], torch.full((B,), v + 1), torch.arange(B) + v * B, {}))
    tm = engine.TestMeter(num_videos=2, num_clips=B, num_cls=cfg.MODEL.NUM_CLASSES, overall_iters=len(loader))
    stats = engine.perform_test(loader, model, tm, cfg)
    assert stats["split"] == "test_final" and set(stats) == {"split", "top1_acc", "top5_acc"}
    assert tm.clip_count.tolist() == [B, B]
    with torch.no_grad():
        ref = model.eval()([loader[0][0][0].cuda()]).float().cpu().sum(0)
    assert torch.allclose(tm.video_preds[0], ref, atol=1e-5)
    assert torch.allclose(tm.video_preds.sum(1), torch.full((2,), float(B)), atol=1e-3)      # softmax scores summed over views


def _grads_for(meta, precision, tmp, clip, labels, act_ckpt=False, scale=None, probe=None):
    from aicity_action_amd.autograd import _BlockFn
    cfg, model = _make(meta, precision, tmp)
    cfg.MODEL.ACT_CHECKPOINT = act_ckpt
    if act_ckpt:
        cfg2, model = _make(meta, precision, tmp)
        model.use_act_checkpoint = True
    model.train()
    torch.cuda.synchronize()
    n0, m0 = _BlockFn.forward_launches, torch.cuda.memory_allocated()
    logits = model([clip])
    loss = engine._loss(cfg, logits, labels)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - m0            # what the autograd graph keeps alive between forward and backward
    n_fwd = _BlockFn.forward_launches - n0
    (loss * scale if scale else loss).backward()
    if probe is not None:
        probe.update(held=held, fwd_in_forward=n_fwd, fwd_total=_BlockFn.forward_launches - n0, depth=len(model.blocks))
    return model, loss.item(), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}


def test_activation_checkpointing_gives_the_same_gradients(tmp_path):
    """MODEL.ACT_CHECKPOINT (video_model_builder.py:1036-1037): only the block inputs are kept, every block's forward kernels run
    a second time inside backward (counted), the memory held between forward and backward drops, gradients are unchanged."""
    _, meta = load_golden("tiny_even")
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], 5).cuda()
    labels = torch.tensor([1, 7]).cuda()
    p0, p1 = {}, {}
    _, l0, g0 = _grads_for(meta, "fp32", str(tmp_path), clip, labels, probe=p0)
    _, l1, g1 = _grads_for(meta, "fp32", str(tmp_path), clip, labels, act_ckpt=True, probe=p1)
    assert p0["fwd_in_forward"] == p0["depth"] and p0["fwd_total"] == p0["depth"]
    assert p1["fwd_in_forward"] == p1["depth"] and p1["fwd_total"] == 2 * p1["depth"]          # re-run once per block in backward
    print("activations held between forward and backward: %.2f MB plain, %.2f MB checkpointed" % (p0["held"] / 2 ** 20, p1["held"] / 2 ** 20))
    assert p1["held"] < 0.5 * p0["held"]
    assert abs(l0 - l1) <= 1e-6
    for k in g0:
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-5 * max(1.0, g0[k].abs().max().item()), k


def test_fp16_training_with_loss_scaler(tmp_path):
    """TRAIN.MIXED_PRECISION semantics (train_net.py:126,231-246) on the fp16 build: scaled backward, unscale + inf check + clip
    in the fused optimizer step, skipped step and scale back-off on overflow, GradScaler-compatible state."""
    from aicity_action_amd.solver import HipGradScaler
    _, meta = load_golden("tiny_even")
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], 5).cuda()
    labels = torch.tensor([1, 7]).cuda()
    # gradients of the fp16 path (scale 1024, unscaled on the host here) point the same way as the fp32 ones
    _, l32, g32 = _grads_for(meta, "fp32", str(tmp_path), clip, labels)
    _, l16, g16 = _grads_for(meta, "fp16", str(tmp_path), clip, labels, scale=1024.0)
    assert abs(l32 - l16) <= 2e-2
    a = torch.cat([g32[k].flatten() for k in g32]).double()
    b = torch.cat([g16[k].flatten() for k in g32]).double() / 1024.0
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    assert cos >= 0.999 and abs(float(b.norm() / a.norm()) - 1.0) <= 0.02, (cos, float(b.norm() / a.norm()))
    # loop with the scaler: an absurd initial scale overflows -> step skipped, scale halves; then steps go through
    cfg, model = _make(meta, "fp16", str(tmp_path))
    opt = construct_optimizer(model, cfg)
    scaler = HipGradScaler(init_scale=2.0 ** 40, growth_interval=2)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    loader = _Loader([([clip.cpu()], labels.cpu(), torch.arange(2), {})] * 1)
    engine.train_epoch(loader, model, opt, scaler, engine.TrainMeter(1, cfg), 0, cfg)
    assert scaler.get_scale() == 2.0 ** 39 and opt.step_count == 0
    assert all(torch.equal(before[k], p) for k, p in model.named_parameters())        # skipped step left everything alone
    scaler.load_state_dict(dict(scaler.state_dict(), scale=4096.0))
    loader = _Loader([([clip.cpu()], labels.cpu(), torch.arange(2), {})] * 4)
    engine.train_epoch(loader, model, opt, scaler, engine.TrainMeter(4, cfg), 0, cfg)
    assert opt.step_count == 4 and scaler.get_scale() == 4096.0 * 4                      # grew twice (interval 2)
    assert sorted(scaler.state_dict()) == ["_growth_tracker", "backoff_factor", "growth_factor", "growth_interval", "scale"]
    moved = sum(int(not torch.equal(before[k], p)) for k, p in model.named_parameters())
    assert moved >= 0.9 * len(before) and all(bool(torch.isfinite(p).all()) for p in model.parameters())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_multi_stream_train_step_matches_single_stream(tmp_path, precision):
    """HIP.TRAIN_STREAMS sub-batches (forward and backward on side streams, gradients summed by autograd) vs the single-stream step."""
    _, meta = load_golden("tiny_even")
    clip = synth_clip(6, meta["num_frames"], meta["crop"], 21).cuda()
    labels = torch.tensor([1, 7, 3, 0, 17, 9]).cuda()
    res = []
    for ns in (1, 2, 3):
        cfg, model = _make(meta, precision, str(tmp_path))
        cfg.HIP.TRAIN_STREAMS = ns
        model.train()
        logits = model([clip])
        loss = engine._loss(cfg, logits, labels)
        loss.backward()
        torch.cuda.synchronize()
        res.append((logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    for ns, (lg, gr) in zip((2, 3), res[1:]):
        assert torch.equal(lg, res[0][0]), ns
        for k in gr:
            ref = res[0][1][k]
            tol = (2e-5 if precision == "fp32" else 2e-2) * max(1.0, ref.abs().max().item())
            assert (gr[k] - ref).abs().max().item() <= tol, (ns, k)


def test_multi_stream_train_step_full_depth_is_deterministic(tmp_path):
    """The 16-block model under HIP.TRAIN_STREAMS 2: each sub-batch chain issues ~130 deterministic column reductions (LayerNorm and
    pooling-conv parameter gradients) whose scratch slots used to come from ONE process-wide counter modulo 64 -- past the wrap
    two chains on different streams could hold the same slot at the same time.  Slots are per stream now: two runs
    must agree bit for bit, and the two-stream gradients must match the single-stream ones (tiny model: 8 reductions, no wrap)."""
    _, meta = load_golden("full224_train")
    clip = synth_clip(4, meta["num_frames"], meta["crop"], 33).cuda()
    labels = torch.tensor([2, 11, 5, 16]).cuda()
    res = []
    for ns in (1, 2, 2):
        cfg, model = _make(meta, "bf16", str(tmp_path))
        cfg.HIP.TRAIN_STREAMS = ns
        model.train()
        loss = engine._loss(cfg, model([clip]), labels)
        loss.backward()
        torch.cuda.synchronize()
        res.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
        del model
    for k in res[1]:
        assert torch.equal(res[1][k], res[2][k]), "two-stream runs differ in %s" % k
        ref = res[0][k]
        err = (res[1][k] - ref).abs().max().item()
        assert err <= 3e-2 * max(ref.abs().max().item(), 1e-3), (k, err, ref.abs().max().item())


def test_graphed_train_step_reproduces_the_eager_step(tmp_path):
    """HIP.GRAPH_STEP: after two eager iterations engine.train_epoch captures the whole step (forward, loss, backward with its
    side streams, clip + AdamW, top-k) in one hipGraph and replays it with the clip / labels / [lr, bias corrections] refreshed in
    device memory.  Every reduction on the path is ordered, so the graphed run must reproduce the eager run BIT FOR BIT:
    per-iteration losses and every parameter after 6 iterations with a different learning rate each."""
    _, meta = load_golden("tiny_even")
    ncls = 18
    loader = _batches(meta, 6, 500, ncls)
    res = {}
    for graph in (False, True):
        cfg, model = _make(meta, "bf16", str(tmp_path / ("g%d" % graph)))
        cfg.HIP.GRAPH_STEP = graph
        cfg.SOLVER.MAX_EPOCH, cfg.SOLVER.WARMUP_EPOCHS = 4, 2.0          # LR changes every iteration
        opt = construct_optimizer(model, cfg)
        tm = engine.TrainMeter(len(loader), cfg)
        lines = []

        class _Grab(logging.Handler):
            def emit(self, record):
                if record.getMessage().startswith("json_stats: "):
                    lines.append(json.loads(record.getMessage().split("json_stats: ")[1]))
        h = _Grab()
        lg = logging.getLogger("aicity_action_amd.engine")
        lg.addHandler(h)
        old = lg.level
        lg.setLevel(logging.INFO)
        try:
            engine.train_epoch(loader, model, opt, None, tm, 0, cfg)
        finally:
            lg.removeHandler(h)
            lg.setLevel(old)
        torch.cuda.synchronize()
        if graph:
            g = engine._graphed_step(model, opt, cfg, None)
            assert g is not None and g.replays == 4 and opt.step_count == 6          # 2 eager + 4 replayed iterations
        res[graph] = ([l["loss"] for l in lines if l["_type"] == "train_iter"], {k: p.detach().clone() for k, p in model.named_parameters()})
    assert len(res[True][0]) == 6 and res[True][0] == res[False][0], (res[True][0], res[False][0])
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k
