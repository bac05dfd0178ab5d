"""Generates tests/golden/mvit_*_stoch.npz: ONE TRAIN STEP OF THE REAL REFERENCE WITH DROP-PATH AND HEAD DROPOUT ON
(build container only).

    python oracle/make_golden_stoch.py            # all cases
    python oracle/make_golden_stoch.py tiny       # subset by name prefix

BASELINE configs[2] trains with MVIT.DROPPATH_RATE 0.4 / MODEL.DROPOUT_RATE 0.5.  The reference draws its noise with
``torch.rand`` inside DropPath (slowfast/models/common.py:46-59; the block calls its one DropPath module twice,
attention.py:434,445) and ``nn.Dropout`` in the head (head_helper.py:410-411).  A HIP kernel cannot reproduce torch's CPU
generator, so parity is pinned on the DRAWS instead: the reference runs here under a fixed ``torch.manual_seed``, forward hooks
on every DropPath and on ``head.dropout`` recover which samples / elements were kept, the restatement (oracle/mvit_oracle.py)
is run with those draws injected and must agree with the reference (loss 1e-6, every clipped gradient 1e-5), and the fixture
stores the draws next to the reference's loss / logits / clipped gradients / post-step parameters.  The GPU test feeds the same
draws through ``aicity_action_amd.autograd.forward_train(noise=...)``.

The torch seed of each case is picked (by the deterministic search below, recorded in the fixture) so that the draws contain
the compositions that matter: a sample dropped in a block's attention branch but kept in its MLP branch, the converse, and the
other sample of the batch kept where the first is dropped -- i.e. factors of exactly 0 next to factors 1/keep in one launch.
Fixtures hold plain arrays only.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import mvit_oracle as O  # noqa: E402
from _reference_loader import build_reference_model, load_reference, reference_cfg  # noqa: E402
from make_golden import FULL_TRAIN_SAMPLE, GOLD, MAX_SAMPLE, TINY, mvit_dict, sample  # noqa: E402
from aicity_action_amd.utils.synth import load_synth_weights, synth_clip  # noqa: E402

CASES = {
    # name: (yaml, overrides, batch, clip seed, per-tensor sample cap)
    "tiny_even_stoch": ("MVITV2_FULL_B_16x4_CONV.yaml", dict(TINY, **{"DATA.TRAIN_CROP_SIZE": 64, "DATA.TEST_CROP_SIZE": 64}), 2, 31, MAX_SAMPLE),
    "tiny_odd_stoch": ("MVITV2_FULL_B_16x4_CONV.yaml", dict(TINY, **{"DATA.TRAIN_CROP_SIZE": 56, "DATA.TEST_CROP_SIZE": 56}), 3, 32, MAX_SAMPLE),
    "tiny_plain_stoch": ("MVITV2_B_16x4_CONV.yaml", dict(TINY, **{"DATA.TRAIN_CROP_SIZE": 64, "DATA.TEST_CROP_SIZE": 64}), 2, 33, MAX_SAMPLE),
    # BASELINE configs[0] geometry at B = 2 with the recipe's own rates (0.4 / 0.5): all 16 blocks, 30 DropPath calls
    "full224_stoch": ("MVITV2_FULL_B_16x4_CONV.yaml", {}, 2, 34, FULL_TRAIN_SAMPLE),
}


def emulate_draws(seed, rates, batch):
    """What the reference's DropPath calls will draw after torch.manual_seed(seed): the forward consumes the CPU generator only
    through torch.rand((B,1,1)) twice per block with rate > 0, in block order (checked against the hooks afterwards)."""
    torch.manual_seed(seed)
    keep = np.ones((len(rates), 2, batch), np.uint8)
    for i, p in enumerate(rates):
        if p > 0.0:
            for j in range(2):
                keep[i, j] = torch.floor((1.0 - p) + torch.rand((batch, 1, 1))).reshape(-1).numpy().astype(np.uint8)
    return keep


def interesting(keep):
    """The compositions the fixture must contain (see the module docstring)."""
    a_drop_m_keep = m_drop_a_keep = mixed_batch = False
    for i in range(keep.shape[0]):
        for b in range(keep.shape[2]):
            a_drop_m_keep |= keep[i, 0, b] == 0 and keep[i, 1, b] == 1
            m_drop_a_keep |= keep[i, 0, b] == 1 and keep[i, 1, b] == 0
        for j in range(2):
            mixed_batch |= 0 < int(keep[i, j].sum()) < keep.shape[2]
    not_all = all(int(keep[:, :, b].sum()) > 0 for b in range(keep.shape[2]))
    return a_drop_m_keep and m_drop_a_keep and mixed_batch and not_all


def run_case(name):
    yaml_name, ov, batch, clip_seed, cap = CASES[name]
    cfg = reference_cfg(yaml_name, ov)
    assert cfg.MVIT.DROPPATH_RATE == 0.4 and cfg.MODEL.DROPOUT_RATE == 0.5      # the recipe's own rates, untouched
    mv = mvit_dict(cfg)
    model = build_reference_model(cfg).train()
    load_synth_weights(model, 0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    mv_built = mvit_dict(cfg)       # construction mutates POOL_KV_STRIDE
    clip = synth_clip(batch, cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, clip_seed)
    depth = len(model.blocks)
    rates = [float(getattr(b.drop_path, "drop_prob", 0.0)) for b in model.blocks]
    spec_rates = [s.drop_path for s in O.derive_specs(mv)]
    assert np.allclose(rates, spec_rates, atol=1e-7), (rates, spec_rates)
    seed = next(s for s in range(1000) if interesting(emulate_draws(s, rates, batch)))
    expect = emulate_draws(seed, rates, batch)

    labels = torch.zeros(batch, cfg.MODEL.NUM_CLASSES)
    for b in range(batch):
        labels[b, (3 * b + 1) % cfg.MODEL.NUM_CLASSES] = 0.9
        labels[b, (5 * b + 2) % cfg.MODEL.NUM_CLASSES] = 0.1

    # --- the reference's step, hooks recording the draws ---------------------------------------------------
    rec = {"dp": [[] for _ in range(depth)], "head": None}
    hooks = []
    for i, blk in enumerate(model.blocks):
        if rates[i] > 0.0:
            def _dp_hook(m, inp, out, i=i):
                kept = (out.detach().reshape(out.shape[0], -1).abs().amax(1) > 0)
                assert bool((inp[0].detach().reshape(out.shape[0], -1).abs().amax(1) > 0).all())     # a zero input would hide the draw
                rec["dp"][i].append(kept.numpy().astype(np.uint8))
            hooks.append(blk.drop_path.register_forward_hook(_dp_hook))

    def _do_hook(m, inp, out):
        assert bool((inp[0] != 0).all())
        rec["head"] = (out.detach() != 0).numpy().astype(np.uint8)
    hooks.append(model.head.dropout.register_forward_hook(_do_hook))

    from slowfast.models import optimizer as ref_optim
    from slowfast.models.losses import get_loss_func
    from slowfast.utils.lr_policy import get_lr_at_epoch
    opt = ref_optim.construct_optimizer(model, cfg)
    cur_epoch = 0.25
    lr = get_lr_at_epoch(cfg, cur_epoch)
    ref_optim.set_lr(opt, lr)
    torch.manual_seed(seed)
    preds = model([clip])                                    # tools/train_net.py:201-246 order from here on
    loss = get_loss_func(cfg.MODEL.LOSS_FUNC)(reduction="mean")(preds, labels)
    opt.zero_grad()
    loss.backward()
    for h in hooks:
        h.remove()
    ref_tot64 = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())).item()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_L2NORM)
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    preds = preds.detach()
    loss_v = loss.item()
    del loss

    dp_keep = np.ones((depth, 2, batch), np.uint8)
    for i in range(depth):
        assert len(rec["dp"][i]) == (2 if rates[i] > 0.0 else 0), (i, len(rec["dp"][i]))
        for j, kept in enumerate(rec["dp"][i]):
            dp_keep[i, j] = kept
    assert np.array_equal(dp_keep, expect), "the hooks saw other draws than the emulated torch.rand sequence"
    head_keep = rec["head"]
    assert head_keep.shape == (batch, sd["head.projection.weight"].shape[1])
    print("[%s] torch seed %d; kept per DropPath call (block: attention | MLP):" % (name, seed),
          "  ".join("%d: %s|%s" % (i, "".join(map(str, dp_keep[i, 0])), "".join(map(str, dp_keep[i, 1])))
                    for i in range(depth) if rates[i] > 0.0), "; head keeps %.3f" % head_keep.mean())

    # --- the restatement with the same draws ---------------------------------------------------------------
    sd2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    o_out, _ = O.forward(sd2, clip, mv, training=True, head_dropout=cfg.MODEL.DROPOUT_RATE,
                         dp_keep=torch.from_numpy(dp_keep), head_keep=torch.from_numpy(head_keep))
    o_loss = O.soft_target_cross_entropy(o_out, labels)
    o_loss.backward()
    assert abs(o_loss.item() - loss_v) <= 1e-6, (o_loss.item(), loss_v)
    assert (o_out.detach() - preds).abs().max().item() <= 1e-5
    tot = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd2.values())).item()
    assert abs(tot - ref_tot64) <= 1e-5 * max(1.0, tot), (tot, ref_tot64)
    coef = min(1.0, cfg.SOLVER.CLIP_GRAD_L2NORM / (gnorm.item() + 1e-6))
    worst = 0.0
    for k in grads:
        d = (sd2[k].grad * coef - grads[k]).abs().max().item()
        worst = max(worst, d / max(1.0, grads[k].abs().max().item()))
        assert d <= 1e-5 * max(1.0, grads[k].abs().max().item()) + 1e-7, (k, d)
    # and the draws matter: the same step without them is a different step
    with torch.no_grad():
        plain, _ = O.forward(sd, clip, mv, training=True)
    assert (plain - preds).abs().max().item() > 1e-3
    print("[%s] oracle(draws) == reference: loss %.6f, |g| %.4f, worst clipped-gradient error %.2e; without the draws the logits move by %.3f"
          % (name, loss_v, tot, worst, (plain - preds).abs().max().item()))

    out = {"train.labels": labels.numpy(), "train.loss": np.array(loss_v, np.float64), "train.logits": preds.numpy(),
           "train.grad_norm": np.array(gnorm.item(), np.float64), "train.grad_norm_fp64": np.array(ref_tot64, np.float64),
           "train.lr": np.array(lr, np.float64), "train.dp_keep": dp_keep, "train.head_keep": head_keep,
           "train.dp_rates": np.array(rates, np.float64)}
    groups = [[], []]
    name_of = {id(p): k for k, p in model.named_parameters()}
    for g in opt.param_groups:
        for p in g["params"]:
            groups[0 if g["weight_decay"] > 0 else 1].append(name_of[id(p)])
    new_sd = model.state_dict()
    l2 = []
    for k in grads:
        s, mom = sample(grads[k], cap)
        out["grad." + k] = s
        out["gmom." + k] = mom
        l2.append(float(grads[k].double().norm().item()))
        out["step." + k] = sample(new_sd[k], cap)[0]
    out["train.grad_l2"] = np.array(l2, np.float64)
    meta = {"yaml": yaml_name, "overrides": ov, "train_overrides": ov, "batch": batch, "clip_seed": clip_seed, "weight_seed": 0,
            "num_frames": cfg.DATA.NUM_FRAMES, "crop": cfg.DATA.TRAIN_CROP_SIZE, "torch_seed": seed,
            "state_keys": list(sd.keys()), "state_shapes": [list(v.shape) for v in sd.values()],
            "droppath_rate": cfg.MVIT.DROPPATH_RATE, "dropout_rate": cfg.MODEL.DROPOUT_RATE,
            "wd_group": groups[0], "no_wd_group": groups[1], "weight_decay": cfg.SOLVER.WEIGHT_DECAY,
            "clip": cfg.SOLVER.CLIP_GRAD_L2NORM, "solver": {k: cfg.SOLVER[k] for k in cfg.SOLVER}, "grad_keys": list(grads.keys()),
            "pool_kv_stride_built": [list(map(int, e)) for e in mv_built["POOL_KV_STRIDE"]]}
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLD, "mvit_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    load_reference()
    os.makedirs(GOLD, exist_ok=True)
    want = sys.argv[1:]
    for n in CASES:
        if not want or any(n.startswith(w) for w in want):
            run_case(n)
