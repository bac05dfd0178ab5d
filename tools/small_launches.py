#!/usr/bin/env python3
"""Which Python lines launch the small fill / copy kernels of a train step?  torch.profiler with stacks over 3 steps of
bench.py's train step (B=8 @448); prints aten::fill_/zero_/copy_/clone/... grouped by the innermost repo frame."""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd.config import load_config
from aicity_action_amd.models import build_model
from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy
from aicity_action_amd.utils.synth import load_synth_weights
cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1, "HIP.PRECISION", "bf16"])
dev = "cuda:0"
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0)
load_synth_weights(model, 0)
model.train()
opt = construct_optimizer(model, cfg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
clip = torch.randn(B, 3, 16, 448, 448, device=dev)
labels = torch.zeros(B, cfg.MODEL.NUM_CLASSES, device=dev); labels[torch.arange(B), torch.arange(B) % cfg.MODEL.NUM_CLASSES] = 1.0
def step():
    logits = model([clip]); loss = soft_target_cross_entropy(logits, labels)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::to", "aten::_to_copy", "aten::cat", "aten::add_", "aten::mul", "aten::add", "aten::div", "aten::empty")
cnt = collections.Counter()
allops = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU": continue
    if not ev.name.startswith("aten::"): continue
    allops[ev.name] += 1
    if ev.name not in want or ev.name == "aten::empty": continue
    fr = next((s for s in ev.stack if ROOT in s or "bench" in s), ev.stack[0] if ev.stack else "?")
    cnt[(ev.name, fr.replace(ROOT + "/", ""))] += 1
print("aten ops per step:", {k: v // 3 for k, v in allops.most_common(25)})
for (name, fr), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print("%5.1f /step  %-18s %s" % (n / 3, name, fr))
