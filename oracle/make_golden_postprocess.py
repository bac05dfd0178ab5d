"""Generates tests/golden/postprocess.json from the REAL reference functions (build container only):
scripts/aicity_inf_graph.py: aggregate_predictions, get_chunks, compute_f1 (importable: numpy/matplotlib/tqdm only).
Inputs are a seeded synthetic score list over the 57-window layout of a 900-frame stream."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join("/root/reference", "scripts"))
sys.dont_write_bytecode = True
_argv, sys.argv = sys.argv, [sys.argv[0]]
import aicity_inf_graph as ref  # noqa: E402
sys.argv = _argv


def synth_preds(seed, num_frames=900, ncls=18):
    g = np.random.Generator(np.random.PCG64([77, seed]))
    out = []
    for t0 in range(0, num_frames, 16):
        p = g.random(ncls).astype(np.float32) ** 3
        if 300 <= t0 < 520:
            p[3] += 0.8
        if 640 <= t0 < 900:
            p[7] += 0.6
        if t0 >= 880:
            p[11] += 0.9
        out.append((t0, t0 + 64, (p / p.sum()).astype(np.float32)))
    return out


def main():
    res = {"cases": []}
    for seed in (0, 1, 2):
        preds = synth_preds(seed)
        case = {"seed": seed}
        for name, fn in (("mean", np.mean), ("max", np.max)):
            agg = ref.aggregate_predictions(preds, fn, 18)
            case["agg_%s_shape" % name] = list(agg.shape)
            case["agg_%s_rows" % name] = {str(i): agg[i].tolist() for i in (0, 15, 16, 63, 64, 447, 899, 900, 959)}
            case["agg_%s_sum" % name] = float(agg.astype(np.float64).sum())
            chunks = {}
            for a, thr in ((3, 0.2), (7, 0.15), (11, 0.1), (5, 0.9)):
                cs = ref.get_chunks(agg[:, a], thr)
                chunks[str(a)] = [[int(c[0]), int(c[1]), int(c[2]), float(c[3])] for c in cs]
            case["chunks_%s" % name] = chunks
        res["cases"].append(case)
    # compute_f1 on a hand-built annotation
    anno = {"1": [("fA", "u", 10, 18, 3), ("fB", "u", 10, 18, 3), ("fC", "u", 10, 18, 3),
                  ("fA", "u", 21, 30, 7), ("fB", "u", 21, 30, 7), ("fC", "u", 21, 30, 7),
                  ("fA", "u", 40, 44, 11), ("fB", "u", 40, 44, 11), ("fC", "u", 40, 44, 11)]}
    chunks = {"fA": {3: [(10.2, 17.6, 222, 0.5)], 7: [(25.0, 30.0, 150, 0.4)], 11: []},
              "fB": {3: [(9.0, 12.0, 90, 0.9)], 7: [(20.6, 30.4, 294, 0.3)], 11: []},
              "fC": {3: [], 7: [], 11: []}}
    for kw in ({}, {"use_num_chunk": 2}, {"chunk_sort_base": "score"}, {"use_tight_times": True}, {"use_ori_times": True, "sec_thres": 0.5}):
        f1, p, r = ref.compute_f1(anno, [3, 7, 11], chunks, return_pr=True, **kw)
        res.setdefault("f1", []).append({"kw": kw, "f1": f1, "p": p, "r": r})
    with open(os.path.join(ROOT, "tests", "golden", "postprocess.json"), "w") as f:
        json.dump(res, f)
    print("wrote tests/golden/postprocess.json")


if __name__ == "__main__":
    main()
