import sys, os, torch, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
import test_hip_engine as T
from aicity_action_amd import engine
from aicity_action_amd.solver import construct_optimizer
_, meta = load_golden("tiny_even")
tl = T._batches(meta, 3, 100, 18)
def run_full(d):
    cfg, m = T._make(meta, "fp32", d); engine.train(cfg, m, tl, None); return m
d = tempfile.mkdtemp()
ma = run_full(d + "/a"); ma2 = run_full(d + "/a2")
def diff(x, y):
    big = tot = 0; worst = 0
    for (k, a), b in zip(x.state_dict().items(), y.state_dict().values()):
        big += int(((a - b).abs() > 1e-4).sum()); tot += a.numel(); worst = max(worst, (a - b).abs().max().item())
    return big, tot, worst
print("a vs a2", diff(ma, ma2))
cfg_b, mb = T._make(meta, "fp32", d + "/b"); ob = construct_optimizer(mb, cfg_b)
engine.train_epoch(tl, mb, ob, None, engine.TrainMeter(3, cfg_b), 0, cfg_b)
engine.save_checkpoint(cfg_b.OUTPUT_DIR, mb, ob, 0, cfg_b)
cfg_c, mc = T._make(meta, "fp32", d + "/b"); oc = construct_optimizer(mc, cfg_c)
print("start epoch", engine.load_train_checkpoint(cfg_c, mc, oc))
print("model b vs c after load", diff(mb, mc), "steps", ob.step_count, oc.step_count, "lr", ob.lr, oc.lr)
ws = 0
for gb, gc in zip(ob.groups, oc.groups):
    for pb, pc in zip(gb["params"], gc["params"]):
        ws = max(ws, (ob.state[pb][0] - oc.state[pc][0]).abs().max().item(), (ob.state[pb][1] - oc.state[pc][1]).abs().max().item())
print("opt state diff", ws)
tm = engine.TrainMeter(3, cfg_b)
engine.train_epoch(tl, mb, ob, None, tm, 1, cfg_b)
engine.train_epoch(tl, mc, oc, None, engine.TrainMeter(3, cfg_c), 1, cfg_c)
print("b vs c after epoch 1", diff(mb, mc)); print("a vs b", diff(ma, mb))
print("---- manual iteration")
cfg_b, mb = T._make(meta, "fp32", d + "/b2"); ob = construct_optimizer(mb, cfg_b)
engine.train_epoch(tl, mb, ob, None, engine.TrainMeter(3, cfg_b), 0, cfg_b)
engine.save_checkpoint(cfg_b.OUTPUT_DIR, mb, ob, 0, cfg_b)
cfg_c, mc = T._make(meta, "fp32", d + "/b2"); oc = construct_optimizer(mc, cfg_c)
engine.load_train_checkpoint(cfg_c, mc, oc)
from aicity_action_amd import solver
for name, m, o in (("b", mb, ob), ("c", mc, oc)):
    m.train()
    inputs, labels, _, _ = tl[0]
    o.set_lr(1e-3)
    preds = m([inputs[0].cuda()])
    loss = engine._loss(cfg_b, preds, labels.cuda())
    o.zero_grad(); loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    out2 = o.step()
    print(name, "loss", loss.item(), "gnorm", gn, "out2", out2.tolist(), "step", o.step_count, "table n", o._n)
print("after manual iter", diff(mb, mc))
