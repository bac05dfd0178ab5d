"""Generates tests/golden/gate_logits.json from the REAL reference (build container only): the pre-softmax logits of the BASELINE
model for several (weight seed, clip seed) pairs at 224 and 448, plus one "trained-like" stressed case (synth.stress_state_dict),
so the north-star gate (16-bit logits within 1e-3 of the reference's fp32 CPU logits) is checked over a distribution of inputs
instead of one clip per size.  Every case: reference (slowfast/models/video_model_builder.py:1161-1335 through
oracle/_reference_loader.py) and the CPU restatement must agree to 1e-5 (relative to the logit scale for the stressed case)
before anything is written.  Fixtures are data: seeds + 18 logits per clip.

    python oracle/make_golden_gate.py
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
from aicity_action_amd.utils.synth import load_synth_weights, stress_state_dict, synth_clip  # noqa: E402
from make_golden import mvit_dict  # noqa: E402

CASES = [
    # (yaml, weight seed, clip seed, stressed)
    ("MVITV2_FULL_B_16x4_CONV.yaml", 1, 101, False),
    ("MVITV2_FULL_B_16x4_CONV.yaml", 2, 102, False),
    ("MVITV2_FULL_B_16x4_CONV.yaml", 3, 103, False),
    ("MVITV2_FULL_B_16x4_CONV.yaml", 4, 104, False),
    ("MVITV2_FULL_B_16x4_CONV_448.yaml", 5, 105, False),
    ("MVITV2_FULL_B_16x4_CONV_448.yaml", 6, 106, False),
    ("MVITV2_FULL_B_16x4_CONV.yaml", 7, 107, True),
]


def main():
    load_reference()
    out = []
    for yaml_name, ws, cs, stressed in CASES:
        cfg = reference_cfg(yaml_name, {})
        mv = mvit_dict(cfg)
        model = build_reference_model(cfg).eval()
        load_synth_weights(model, ws)
        if stressed:
            model.load_state_dict(stress_state_dict(model.state_dict()))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        clip = synth_clip(1, cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, cs)
        ref = {}
        hk = model.head.projection.register_forward_hook(lambda m, i, o: ref.__setitem__("logits", o.detach()))
        with torch.no_grad():
            probs = model([clip])
        hk.remove()
        with torch.no_grad():
            o_probs, o_logits = O.forward(sd, clip, mv)
        scale = max(1.0, ref["logits"].abs().max().item())
        d = (o_logits - ref["logits"]).abs().max().item()
        assert d <= 1e-5 * scale, (yaml_name, ws, cs, d)
        print("[%s ws %d cs %d%s] oracle==reference %.2e  |logit| max %.3f  top prob %.3f" % (
            yaml_name, ws, cs, " stressed" if stressed else "", d, ref["logits"].abs().max().item(), probs.max().item()))
        out.append({"yaml": yaml_name, "crop": int(cfg.DATA.TRAIN_CROP_SIZE), "num_frames": int(cfg.DATA.NUM_FRAMES), "weight_seed": ws,
                    "clip_seed": cs, "stressed": stressed, "logits": [float(x) for x in ref["logits"].reshape(-1).tolist()],
                    "probs": [float(x) for x in probs.reshape(-1).tolist()]})
    path = os.path.join(ROOT, "tests", "golden", "gate_logits.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/make_golden_gate.py", "stress": {"linear_gain": 4.0, "qk_gain": 2.5}, "cases": out}, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
