"""CPU: the oracle (oracle/mvit_oracle.py) against the committed golden vectors that were produced by
the real reference (oracle/make_golden.py). Runs anywhere, no reference needed."""
import copy

import numpy as np
import pytest
import torch

from conftest import cfg_for_case, load_golden, sample_like

import mvit_oracle as O
from aicity_action_amd.utils.synth import synth_clip, synth_state_dict


def _mv(cfg):
    return copy.deepcopy(cfg.MVIT.to_dict())


def _run(name, train=False):
    z, meta = load_golden(name)
    cfg = cfg_for_case(meta, train=train)
    sd = synth_state_dict(dict(zip(meta["state_keys"], meta["state_shapes"])), meta["weight_seed"])
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"])
    return z, meta, cfg, sd, clip


@pytest.mark.parametrize("name", ["tiny_even", "tiny_odd", "tiny_plain", "full224", "v32x3_224", "plain224"])
def test_oracle_forward_matches_reference_golden(name):
    z, meta, cfg, sd, clip = _run(name)
    taps = {}
    with torch.no_grad():
        probs, logits = O.forward(sd, clip, _mv(cfg), taps=taps)
    assert np.abs(logits.numpy() - z["logits"]).max() <= 1e-5
    assert np.abs(probs.numpy() - z["probs"]).max() <= 1e-5
    for k in [k for k in z.files if k.startswith("tap.")]:
        got = sample_like(taps[k[4:]], z["mom." + k[4:]])
        ref = z[k]
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
    for k in [k for k in z.files if k.startswith("thw")]:
        assert list(taps[k]) == list(z[k])


@pytest.mark.parametrize("name", ["tiny_even", "tiny_odd", "tiny_plain"])
def test_oracle_train_step_matches_reference_golden(name):
    z, meta, cfg, sd, clip = _run(name, train=True)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    labels = torch.from_numpy(z["train.labels"])
    out, _ = O.forward(sd, clip, _mv(cfg), training=True)
    loss = O.soft_target_cross_entropy(out, labels)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) <= 1e-6
    assert np.abs(out.detach().numpy() - z["train.logits"]).max() <= 1e-5
    tot = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values())).item()
    assert abs(tot - float(z["train.grad_norm"])) <= 1e-5 * max(1.0, tot)
    coef = min(1.0, meta["clip"] / (tot + 1e-6))          # torch clip_grad_norm_ semantics
    lr = O.lr_at_epoch(meta["solver"], 0.25)
    assert abs(lr - float(z["train.lr"])) < 1e-12
    wd_names = set(meta["wd_group"])
    assert len(meta["wd_group"]) + len(meta["no_wd_group"]) == len(sd)
    for k, v in sd.items():
        gref = z["grad." + k]
        g = v.grad * coef
        got = sample_like(g, z["gmom." + k])
        assert np.abs(got - gref).max() <= 1e-5 * max(1.0, np.abs(gref).max()) + 1e-7, k
        # one AdamW step (decoupled weight decay, bias-corrected, eps 1e-8), step 1
        wd = meta["weight_decay"] if k in wd_names else 0.0
        p = v.detach() * (1 - lr * wd)
        m = 0.1 * g
        s = 0.001 * g * g
        p = p - lr * (m / 0.1) / ((s / 0.001).sqrt() + 1e-8)
        got = sample_like(p, z["gmom." + k])
        assert np.abs(got - z["step." + k]).max() <= 2e-6, k


@pytest.mark.parametrize("name", ["tiny_even_stoch", "tiny_odd_stoch", "tiny_plain_stoch", "full224_stoch"])
def test_oracle_stochastic_train_step_matches_reference_golden(name):
    """Row a11: a train step of the REAL reference with DROPPATH_RATE 0.4 / DROPOUT_RATE 0.5 on (oracle/make_golden_stoch.py).  The
    fixture carries the reference's own draws (DropPath keeps per call, common.py:46-59; the head Dropout's kept elements,
    head_helper.py:410-411); the restatement with those draws injected reproduces loss, logits and every clipped gradient."""
    z, meta, cfg, sd, clip = _run(name, train=True)
    assert cfg.MVIT.DROPPATH_RATE == 0.4 and cfg.MODEL.DROPOUT_RATE == 0.5
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    dp_keep, head_keep = torch.from_numpy(z["train.dp_keep"]), torch.from_numpy(z["train.head_keep"])
    assert set(np.unique(z["train.dp_keep"])) == {0, 1} and dp_keep.shape == (len(z["train.dp_rates"]), 2, meta["batch"])
    assert bool((dp_keep[0] == 1).all())                     # block 0 has nn.Identity (attention.py:349-351)
    out, _ = O.forward(sd, clip, _mv(cfg), training=True, head_dropout=cfg.MODEL.DROPOUT_RATE, dp_keep=dp_keep, head_keep=head_keep)
    loss = O.soft_target_cross_entropy(out, torch.from_numpy(z["train.labels"]))
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) <= 1e-6
    assert np.abs(out.detach().numpy() - z["train.logits"]).max() <= 1e-5
    tot = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values())).item()
    assert abs(tot - float(z["train.grad_norm_fp64"])) <= 1e-5 * tot
    coef = min(1.0, meta["clip"] / (float(z["train.grad_norm"]) + 1e-6))
    for k, l2 in zip(meta["grad_keys"], z["train.grad_l2"]):
        g = sd[k].grad * coef
        gref = z["grad." + k]
        assert np.abs(sample_like(g, z["gmom." + k]) - gref).max() <= 1e-5 * max(1.0, np.abs(gref).max()) + 1e-7, k
        assert abs(float(g.double().norm()) - l2) <= 1e-5 * max(l2, 1e-3), k
    # a sample dropped by a DropPath call contributes nothing to that branch's parameters: with EVERY sample of the batch dropped in
    # block i's attention branch the gradient of that block's attention parameters is exactly zero
    for i in range(dp_keep.shape[0]):
        if int(dp_keep[i, 0].sum()) == 0:
            assert float(sd["blocks.%d.attn.proj.weight" % i].grad.abs().max()) == 0.0
        if int(dp_keep[i, 1].sum()) == 0:
            assert float(sd["blocks.%d.mlp.fc2.weight" % i].grad.abs().max()) == 0.0


def test_oracle_full224_train_step_matches_reference_golden():
    """BASELINE configs[0] geometry (all 16 blocks, real widths), one clip: loss, train-mode logits, gradient norm and every
    parameter's post-clip gradient samples of the REAL reference's train step (tests/golden/mvit_full224_train.npz) against
    autograd over the restatement.  (@448 the same check needs ~20 GB / 1 min of host autograd: oracle/make_golden.py runs it
    when the fixture is generated.)"""
    z, meta = load_golden("full224_train")
    cfg = cfg_for_case(meta, train=True)
    sd = synth_state_dict(dict(zip(meta["state_keys"], meta["state_shapes"])), meta["weight_seed"])
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"])
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    out, _ = O.forward(sd, clip, _mv(cfg), training=True)
    loss = O.soft_target_cross_entropy(out, torch.from_numpy(z["train.labels"]))
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) <= 1e-6
    assert np.abs(out.detach().numpy() - z["train.logits"]).max() <= 1e-5
    tot = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values())).item()
    assert abs(tot - float(z["train.grad_norm_fp64"])) <= 1e-5 * tot
    coef = min(1.0, meta["clip"] / (float(z["train.grad_norm"]) + 1e-6))     # the reference's own (fp32-accumulated) norm
    for k, l2 in zip(meta["grad_keys"], z["train.grad_l2"]):
        g = sd[k].grad * coef
        got = sample_like(g, z["gmom." + k])
        gref = z["grad." + k]
        assert np.abs(got - gref).max() <= 1e-5 * max(1.0, np.abs(gref).max()) + 1e-7, k
        assert abs(float(g.double().norm()) - l2) <= 1e-5 * max(l2, 1e-3), k


def test_oracle_matches_the_reference_logits_of_the_statistical_gate_cases():
    """tests/golden/gate_logits.json (oracle/make_golden_gate.py: the real reference's logits for further (weight seed, clip seed) pairs
    and the trained-like stressed model): the CPU restatement reproduces them anywhere.  The 224 cases only (a 448 forward takes ~10 s
    of CPU each; those two are covered on the GPU side and at generation time)."""
    import json
    import os
    from conftest import GOLD, ROOT
    from aicity_action_amd.config import load_config
    from aicity_action_amd.models.mvit import MViT
    from aicity_action_amd.utils.synth import stress_state_dict
    cases = json.load(open(os.path.join(GOLD, "gate_logits.json")))["cases"]
    done = 0
    for c in cases:
        if c["crop"] != 224 or (not c["stressed"] and done >= 2):
            continue
        cfg = load_config(os.path.join(ROOT, "configs", "Aicity", c["yaml"]), ["NUM_GPUS", 0])
        mv = copy.deepcopy(cfg.MVIT.to_dict())
        sd = synth_state_dict({k: v.shape for k, v in MViT(cfg).state_dict().items()}, c["weight_seed"])
        if c["stressed"]:
            sd = stress_state_dict(sd)
        with torch.no_grad():
            _, lg = O.forward(sd, synth_clip(1, c["num_frames"], 224, c["clip_seed"]), mv)
        ref = np.array(c["logits"], np.float32)
        assert np.abs(lg.numpy().reshape(-1) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        done += 0 if c["stressed"] else 1
