"""tools/find_copies.py [fwd|train]: which Python lines issue device copies / fills in one step (torch.profiler, with_stack)."""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aicity_action_amd.config import load_config
from aicity_action_amd.models import build_model
from aicity_action_amd.utils.synth import load_synth_weights

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1])
model = build_model(cfg, gpu_id=0)
load_synth_weights(model, 0)
clip = torch.randn(8, 3, 16, 448, 448, device="cuda")
if mode == "fwd":
    model.eval()
    fn = lambda: model([clip])
else:
    model.train()
    def fn():
        out = model([clip])
        out.float().sum().backward()
with torch.no_grad() if mode == "fwd" else torch.enable_grad():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        fn()
        torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::to", "aten::_to_copy", "aten::zeros", "aten::cat", "aten::contiguous", "aten::clone"):
        st = [s for s in (e.stack or []) if "aicity_action_amd" in s or "bench" in s or "tools/" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(40):
    print("%4d  %-16s %s" % (c, n, s))
