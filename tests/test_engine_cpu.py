"""CPU: host logic of the train / eval / test loops (aicity_action_amd/engine.py) against tests/golden/train_loop.json, which
oracle/make_golden_loop.py generated from the reference's own metrics / logging / checkpoint / lr_policy modules."""
import json
import os

import pytest
import torch

from aicity_action_amd import engine, solver
from aicity_action_amd.config import get_cfg, load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "train_loop.json")))


def test_topks_correct_matches_reference():
    for c in G["topks_correct"]:
        got = engine.topks_correct(torch.tensor(c["preds"]), torch.tensor(c["labels"]), tuple(c["ks"]))
        assert [float(x) for x in got] == c["correct"]


def test_json_stats_lines_are_byte_identical():
    for s, line in zip(G["json_stats"]["samples"], G["json_stats"]["lines"]):
        assert engine.json_stats_line(s) == line


def test_checkpoint_naming_and_selection(tmp_path):
    for e, p in G["ckpt_paths"].items():
        assert engine.get_path_to_checkpoint("/job", int(e)) == p
    assert engine.get_checkpoint_dir("/job") == G["ckpt_dir"]
    d = str(tmp_path)
    assert not engine.has_checkpoint(d)
    os.makedirs(engine.get_checkpoint_dir(d))
    for n in G["last_checkpoint"]["names"]:
        open(os.path.join(engine.get_checkpoint_dir(d), n), "wb").close()
    assert engine.has_checkpoint(d) == G["last_checkpoint"]["has"]
    assert os.path.basename(engine.get_last_checkpoint(d)) == G["last_checkpoint"]["last"]


def test_checkpoint_and_eval_epoch_schedule():
    for s in G["epoch_schedule"]:
        cfg = get_cfg()
        cfg.SOLVER.MAX_EPOCH, cfg.TRAIN.CHECKPOINT_PERIOD, cfg.TRAIN.EVAL_PERIOD = s["max_epoch"], s["ckpt_period"], s["eval_period"]
        assert [engine.is_checkpoint_epoch(cfg, e) for e in range(s["max_epoch"])] == s["is_ckpt"]
        ev = [engine.is_eval_epoch(cfg, e) for e in range(s["max_epoch"])]
        assert ev[-1] and all(ev[e] == ((e + 1) % s["eval_period"] == 0) for e in range(s["max_epoch"] - 1))


def test_per_iteration_learning_rate():
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", G["iter_lr"]["yaml"]))
    ds = G["iter_lr"]["data_size"]
    for p in G["iter_lr"]["points"]:
        assert solver.get_lr_at_epoch(cfg, p["epoch"] + float(p["iter"]) / ds) == pytest.approx(p["lr"], rel=1e-12, abs=0)


class _TinyOpt(object):
    def __init__(self, m):
        self.o = torch.optim.AdamW(m.parameters(), lr=1e-3)

    def state_dict(self):
        return self.o.state_dict()

    def load_state_dict(self, sd):
        self.o.load_state_dict(sd)


def test_checkpoint_file_layout_and_shape_matched_load(tmp_path):
    cfg = get_cfg()
    cfg.OUTPUT_DIR = str(tmp_path)
    m = torch.nn.Linear(3, 2)
    opt = _TinyOpt(m)
    p = engine.save_checkpoint(cfg.OUTPUT_DIR, m, opt, 4, cfg)
    lay = G["ckpt_layout"]
    assert os.path.basename(p) == lay["file"]
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == lay["keys"] and ck["epoch"] == lay["epoch"]
    assert list(ck["model_state"].keys()) == lay["model_state_keys"]
    assert sorted(ck["optimizer_state"].keys()) == lay["optimizer_state_keys"] and type(ck["cfg"]).__name__ == lay["cfg_type"]
    # resume: auto-resume picks the file up and returns epoch + 1; a differently shaped tensor is skipped, not fatal
    m2 = torch.nn.Linear(3, 2)
    assert engine.load_train_checkpoint(cfg, m2, _TinyOpt(m2)) == 5
    assert torch.equal(m2.weight, m.weight)
    m3 = torch.nn.Linear(4, 2)
    w3 = m3.weight.detach().clone()
    assert engine.load_checkpoint(p, m3) == 4
    assert torch.equal(m3.weight, w3) and torch.equal(m3.bias, m.bias)
    cfg.TRAIN.AUTO_RESUME = False
    assert engine.load_train_checkpoint(cfg, m2, _TinyOpt(m2)) == 0


def test_meters_window_median_and_epoch_means():
    cfg = get_cfg()
    cfg.LOG_PERIOD, cfg.SOLVER.MAX_EPOCH = 2, 5
    tm = engine.TrainMeter(4, cfg)
    lines = []
    for it, (e1, e5, loss) in enumerate([(100.0, 50.0, 3.0), (50.0, 0.0, 1.0), (0.0, 0.0, 2.0), (100.0, 100.0, 4.0)]):
        tm.update_stats(e1, e5, loss, 1e-4, 8)
        lines.append(tm.log_iter_stats(0, it))
    assert lines[0] is None and lines[2] is None
    d1 = json.loads(lines[1].split("json_stats: ")[1])
    assert d1["_type"] == "train_iter" and d1["iter"] == "2/4" and d1["epoch"] == "1/5" and d1["loss"] == 2.0 and d1["top1_err"] == 75.0
    d3 = json.loads(lines[3].split("json_stats: ")[1])
    assert d3["loss"] == 3.0 and d3["top5_err"] == 50.0          # median of the last LOG_PERIOD values
    ep = json.loads(tm.log_epoch_stats(0).split("json_stats: ")[1])
    assert ep["_type"] == "train_epoch" and ep["loss"] == 2.5 and ep["top1_err"] == 62.5 and ep["top5_err"] == 37.5
    vm = engine.ValMeter(2, cfg)
    vm.update_stats(40.0, 10.0, 8)
    vm.update_stats(20.0, 0.0, 8)
    assert vm.log_epoch_stats(0) == 5.0 and vm.min_top1_err == 30.0


def test_test_meter_view_sum_ensemble():
    tm = engine.TestMeter(num_videos=2, num_clips=3, num_cls=4, overall_iters=2)
    preds = torch.tensor([[0.3, 0.5, 0.1, 0.1], [0.6, 0.2, 0.1, 0.1], [0.5, 0.3, 0.1, 0.1],      # video 0: sum favours class 0
                          [0.0, 0.0, 0.9, 0.1], [0.0, 0.1, 0.8, 0.1], [0.0, 0.0, 0.2, 0.8]])     # video 1: class 2
    tm.update_stats(preds[:3], torch.tensor([1, 1, 1]), torch.tensor([0, 1, 2]))
    tm.update_stats(preds[3:], torch.tensor([2, 2, 2]), torch.tensor([3, 4, 5]))
    st = tm.finalize_metrics(ks=(1, 2))
    assert st == {"split": "test_final", "top1_acc": "50.00", "top2_acc": "100.00"}
    assert torch.allclose(tm.video_preds[0], preds[:3].sum(0)) and tm.clip_count.tolist() == [3, 3]


def test_device_scalar_queue_and_async_meter_on_host_tensors():
    """The non-blocking statistics path with CPU tensors (no GPU here): rows pass straight through, in order; the async entry
    point of TrainMeter gives the same log lines as the float one; a NaN loss raises when its row is absorbed."""
    from aicity_action_amd.meters import DeviceScalarQueue
    q = DeviceScalarQueue(3, depth=2)
    for i in range(5):
        q.put(torch.tensor([float(i), 2.0 * i, 3.0 * i]), ("tag", i))
    rows = q.ready()
    assert [t for _, t in rows] == [("tag", i) for i in range(5)] and [float(r[1]) for r, _ in rows] == [0.0, 2.0, 4.0, 6.0, 8.0]
    assert q.ready(wait=True) == []
    cfg = get_cfg()
    cfg.LOG_PERIOD, cfg.SOLVER.MAX_EPOCH = 2, 5
    a, b = engine.TrainMeter(4, cfg), engine.TrainMeter(4, cfg)
    data = [(100.0, 50.0, 3.0), (50.0, 0.0, 1.0), (0.0, 0.0, 2.0), (100.0, 100.0, 4.0)]
    for it, (e1, e5, loss) in enumerate(data):
        a.update_stats(e1, e5, loss, 1e-4, 8)
        b.update_stats_async(torch.tensor([loss, e1, e5]), 1e-4, 8)
        la, lb = a.log_iter_stats(0, it), b.log_iter_stats(0, it)
        if la is not None:
            la, lb = json.loads(la.split("json_stats: ")[1]), json.loads(lb.split("json_stats: ")[1])
            la.pop("gpu_mem"), lb.pop("gpu_mem")
            assert la == lb
    ea, eb = [json.loads(m.log_epoch_stats(0).split("json_stats: ")[1]) for m in (a, b)]
    assert (ea["loss"], ea["top1_err"], ea["top5_err"]) == (eb["loss"], eb["top1_err"], eb["top5_err"]) == (2.5, 62.5, 37.5)
    c = engine.TrainMeter(4, cfg)
    with pytest.raises(RuntimeError, match="NaN losses"):
        c.update_stats_async(torch.tensor([float("nan"), 0.0, 0.0]), 1e-4, 8)
    s = engine.ScalarMeter(2)
    for v in (1.0, 5.0, 3.0):
        s.add_value(v)
    assert (s.get_win_median(), s.get_win_avg(), s.get_global_avg(), s.count) == (4.0, 4.0, 3.0, 3)


def test_test_meter_max_ensemble_and_label_consistency():
    tm = engine.TestMeter(num_videos=2, num_clips=2, num_cls=3, overall_iters=1, ensemble_method="max")
    preds = torch.tensor([[0.2, 0.7, 0.1], [0.6, 0.3, 0.1], [0.1, 0.1, 0.8], [0.3, 0.3, 0.4]])
    tm.update_stats(preds, torch.tensor([1, 1, 2, 2]), torch.tensor([0, 1, 2, 3]))
    assert torch.allclose(tm.video_preds, torch.tensor([[0.6, 0.7, 0.1], [0.3, 0.3, 0.8]]))
    with pytest.raises(AssertionError):
        tm.update_stats(preds[:1], torch.tensor([2]), torch.tensor([0]))          # video 0 was labelled 1
    with pytest.raises(NotImplementedError):
        engine.TestMeter(1, 1, 3, 1, ensemble_method="mean")
