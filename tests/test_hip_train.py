"""GPU: the training step (forward with saved activations, hand-written backward, clipped AdamW) against the
golden vectors produced by the real reference (loss, every parameter gradient, parameters after one step)."""
import numpy as np
import pytest
import torch

from conftest import cfg_for_case, load_golden, sample_like

from aicity_action_amd.models import build_model
from aicity_action_amd.solver import construct_optimizer, get_lr_at_epoch, param_groups, soft_target_cross_entropy
from aicity_action_amd.utils.synth import load_synth_weights, synth_clip

pytestmark = pytest.mark.gpu


def _setup(name, precision):
    z, meta = load_golden(name)
    cfg = cfg_for_case(meta, precision, train=True)
    cfg.NUM_GPUS = 1
    model = build_model(cfg).train()
    load_synth_weights(model, meta["weight_seed"])
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"]).cuda()
    labels = torch.from_numpy(z["train.labels"]).cuda()
    return z, meta, cfg, model, clip, labels


@pytest.mark.parametrize("name", ["tiny_even", "tiny_odd", "tiny_plain"])
def test_fp32_train_step_matches_reference_golden(name):
    z, meta, cfg, model, clip, labels = _setup(name, "fp32")
    decay, no_decay = param_groups(model, cfg)
    assert sorted(n for n, _ in decay) == sorted(meta["wd_group"]) and sorted(n for n, _ in no_decay) == sorted(meta["no_wd_group"])
    opt = construct_optimizer(model, cfg)
    lr = get_lr_at_epoch(cfg, 0.25)
    assert abs(lr - float(z["train.lr"])) < 1e-12
    opt.set_lr(lr)
    logits = model([clip])
    assert np.abs(logits.detach().cpu().numpy() - z["train.logits"]).max() <= 1e-4
    loss = soft_target_cross_entropy(logits, labels)
    assert abs(loss.item() - float(z["train.loss"])) <= 1e-5
    loss.backward()
    out2 = opt.step()
    tot = out2[0].item()
    assert abs(tot - float(z["train.grad_norm"])) <= 1e-4 * max(1.0, tot)
    coef = out2[1].item()
    worst = 0.0
    for k, p in model.named_parameters():
        gref = z["grad." + k]                              # reference grads are post-clip
        got = sample_like(p.grad * coef, z["gmom." + k])
        err = np.abs(got - gref).max() / max(1.0, np.abs(gref).max())
        worst = max(worst, err)
        assert err <= 1e-4, (k, err)
        got = sample_like(p, z["gmom." + k])
        assert np.abs(got - z["step." + k]).max() <= 5e-6, k
    print("[%s fp32] worst relative grad error %.2e" % (name, worst))


@pytest.mark.parametrize("name", ["tiny_even", "tiny_odd", "tiny_plain"])
def test_bf16_train_step_is_close_to_reference(name):
    z, meta, cfg, model, clip, labels = _setup(name, "bf16")
    logits = model([clip])
    loss = soft_target_cross_entropy(logits, labels)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) <= 2e-2
    tot = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())).item()
    ref_tot = float(z["train.grad_norm"])
    print("[%s bf16] loss %.5f (ref %.5f) |g| %.4f (ref %.4f)" % (name, loss.item(), float(z["train.loss"]), tot, ref_tot))
    assert abs(tot - ref_tot) <= 0.05 * ref_tot
    coef = min(1.0, meta["clip"] / (tot + 1e-6))
    # cosine similarity of the sampled gradient vector with the reference's
    a, b = [], []
    for k, p in model.named_parameters():
        a.append(sample_like(p.grad * coef, z["gmom." + k]))
        b.append(z["grad." + k])
    a, b = np.concatenate(a), np.concatenate(b)
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))
    print("[%s bf16] gradient cosine vs reference %.5f" % (name, cos))
    assert cos >= 0.995


def test_drop_path_and_dropout_statistics():
    """Stochastic ops: with DROPPATH_RATE/DROPOUT on, outputs differ run to run in train mode, are deterministic in
    eval mode, and the per-sample drop-path factor takes only the values {0, 1/keep}."""
    z, meta = load_golden("tiny_even")
    cfg = cfg_for_case(meta, "fp32")
    cfg.NUM_GPUS = 1
    model = build_model(cfg)
    load_synth_weights(model, 0)
    clip = synth_clip(2, meta["num_frames"], meta["crop"], 3).cuda()
    model.train()
    with torch.no_grad():
        a, b = model([clip]), model([clip])
    assert not torch.equal(a, b)
    model.eval()
    with torch.no_grad():
        c, d = model([clip]), model([clip])
    assert torch.equal(c, d) and torch.allclose(c.sum(1), torch.ones(2, device=c.device), atol=1e-5)
