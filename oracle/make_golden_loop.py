#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- build container only (imports the real reference from /root/reference).

Golden vectors for the host logic of the train / eval / test loops, taken from the
reference modules that import here with the third-party stubs of ``_reference_loader``:

  slowfast/utils/metrics.py     topks_correct                         (train_net.py:262, test meters)
  slowfast/utils/logging.py     log_json_stats line format            (meters.py: every *_iter / *_epoch line)
  slowfast/utils/checkpoint.py  path naming, last-checkpoint choice, is_checkpoint_epoch, save_checkpoint dict layout
  slowfast/utils/lr_policy.py   per-iteration learning rate of train_epoch (train_net.py:118)

``slowfast/utils/meters.py`` and ``slowfast/utils/misc.py`` do not import here (they pull the dataset package: cv2, av,
matplotlib ...), so ``is_eval_epoch`` (same arithmetic as the pinned ``is_checkpoint_epoch``) and the meter classes
in aicity_action_amd/engine.py are restated from its text and pinned only through these pieces ("parity unpinned" for the
windowed-median bookkeeping itself).  Writes tests/golden/train_loop.json.
"""
import io
import json
import logging
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_loader as R  # noqa: E402


def main():
    R.load_reference()
    import slowfast.utils.checkpoint as cu
    import slowfast.utils.logging as rlog
    import slowfast.utils.lr_policy as lrp
    import slowfast.utils.metrics as metrics

    out = {}
    # ---- topks_correct ------------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    cases = []
    for n, c in ((8, 18), (5, 18), (16, 7)):
        preds = torch.rand(n, c, generator=g)
        labels = torch.randint(0, c, (n,), generator=g)
        ks = (1, 5)
        res = [float(x) for x in metrics.topks_correct(preds, labels, ks)]
        cases.append({"preds": preds.tolist(), "labels": labels.tolist(), "ks": list(ks), "correct": res})
    out["topks_correct"] = cases

    # ---- log_json_stats line format --------------------------------------------------------------------------
    buf = io.StringIO()
    lg = logging.getLogger("slowfast.utils.logging")
    h = logging.StreamHandler(buf)
    h.setFormatter(logging.Formatter("%(message)s"))
    lg.addHandler(h)
    lg.setLevel(logging.INFO)
    lg.propagate = False
    samples = [
        {"_type": "train_iter", "epoch": "3/200", "iter": "10/57", "loss": 2.123456789, "lr": 0.0000123456, "gpu_mem": "12.34G",
         "top1_err": 87.5, "top5_err": 50.0},
        {"_type": "train_epoch", "epoch": "3/200", "lr": 1e-4, "gpu_mem": "1.00G", "RAM": "10.00/64.00G", "loss": 1.0 / 3.0,
         "top1_err": 66.66666666, "top5_err": 12.5},
        {"_type": "val_epoch", "epoch": "10/200", "gpu_mem": "1.00G", "RAM": "10.00/64.00G", "top1_err": 45.0, "top5_err": 5.0,
         "min_top1_err": 44.123456, "min_top5_err": 5.0},
        {"split": "test_final", "top1_acc": "71.43", "top5_acc": "100.00"},
    ]
    lines = []
    for s in samples:
        buf.seek(0)
        buf.truncate()
        rlog.log_json_stats(s)
        lines.append(buf.getvalue().strip())
    out["json_stats"] = {"samples": samples, "lines": lines}

    # ---- checkpoint naming / scheduling ----------------------------------------------------------------------
    out["ckpt_paths"] = {str(e): cu.get_path_to_checkpoint("/job", e) for e in (1, 10, 123, 20000)}
    out["ckpt_dir"] = cu.get_checkpoint_dir("/job")
    sched = []
    for max_epoch, period, eval_period in ((200, 10, 10), (7, 3, 2), (1, 1, 1), (30, 1, 5)):
        cfg = R.reference_cfg(None, {"SOLVER.MAX_EPOCH": max_epoch, "TRAIN.CHECKPOINT_PERIOD": period, "TRAIN.EVAL_PERIOD": eval_period})
        sched.append({"max_epoch": max_epoch, "ckpt_period": period, "eval_period": eval_period,
                      "is_ckpt": [bool(cu.is_checkpoint_epoch(cfg, e, None)) for e in range(max_epoch)]})
    out["epoch_schedule"] = sched
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(cu.get_checkpoint_dir(d))
        names = ["checkpoint_epoch_00002.pyth", "checkpoint_epoch_00010.pyth", "checkpoint_epoch_00009.pyth", "notes.txt"]
        for n in names:
            open(os.path.join(cu.get_checkpoint_dir(d), n), "wb").close()
        out["last_checkpoint"] = {"names": names, "last": os.path.basename(cu.get_last_checkpoint(d)),
                                  "has": bool(cu.has_checkpoint(d))}
        # dict layout written by save_checkpoint (tiny module, reference AdamW groups are irrelevant here)
        m = torch.nn.Linear(3, 2)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
        cfg = R.reference_cfg(None, {})
        p = cu.save_checkpoint(d, m, opt, 4, cfg, scaler=None)
        ck = torch.load(p, map_location="cpu", weights_only=False)
        out["ckpt_layout"] = {"file": os.path.basename(p), "keys": sorted(ck.keys()), "epoch": ck["epoch"],
                              "model_state_keys": list(ck["model_state"].keys()),
                              "optimizer_state_keys": sorted(ck["optimizer_state"].keys()), "cfg_type": type(ck["cfg"]).__name__}

    # ---- per-iteration learning rate as train_epoch sets it ------------------------------------------------------
    cfg = R.reference_cfg("MVITV2_FULL_B_16x4_CONV_448.yaml", {})
    data_size = 57
    its = [(0, 0), (0, 28), (0, 56), (1, 0), (4, 30), (29, 56), (30, 0), (100, 5), (199, 56)]
    out["iter_lr"] = {"yaml": "MVITV2_FULL_B_16x4_CONV_448.yaml", "data_size": data_size,
                      "points": [{"epoch": e, "iter": i, "lr": float(lrp.get_lr_at_epoch(cfg, e + float(i) / data_size))} for e, i in its]}

    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "train_loop.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
