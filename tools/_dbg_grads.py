import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from conftest import cfg_for_case, load_golden, sample_like
from aicity_action_amd.models import build_model
from aicity_action_amd.solver import soft_target_cross_entropy
from aicity_action_amd.utils.synth import load_synth_weights, synth_clip
z, meta = load_golden("tiny_even")
cfg = cfg_for_case(meta, "fp32", train=True); cfg.NUM_GPUS = 1
model = build_model(cfg).train(); load_synth_weights(model, 0)
clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"]).cuda()
labels = torch.from_numpy(z["train.labels"]).cuda()
loss = soft_target_cross_entropy(model([clip]), labels); loss.backward()
coef = min(1.0, 1.0 / (float(z["train.grad_norm"]) + 1e-6))
for k, p in model.named_parameters():
    g = sample_like(p.grad * coef, z["gmom." + k]); r = z["grad." + k]
    e = np.abs(g - r).max() / max(1e-12, np.abs(r).max())
    if e > 1e-3: print("%-40s rel err %.3e  |ref| %.3e" % (k, e, np.abs(r).max()))
