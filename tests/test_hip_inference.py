"""GPU: sliding-window front end (gather + cv2-style 8-bit bilinear resize + normalise) bit-exact against the oracle
restatement, and the end-to-end windowed inference against per-window model calls."""
import numpy as np
import pytest
import torch

from conftest import cfg_for_case, load_golden

import window_oracle as WO
from aicity_action_amd.inference import SlidingWindowClassifier, get_proposals
from aicity_action_amd.models import build_model
from aicity_action_amd.utils.synth import load_synth_weights

pytestmark = pytest.mark.gpu


def _stream(n, h, w, seed):
    g = np.random.Generator(np.random.PCG64([9, seed]))
    base = g.integers(0, 256, (n, h // 6 + 1, w // 6 + 1, 3), dtype=np.uint8)
    return np.ascontiguousarray(np.repeat(np.repeat(base, 6, 1), 6, 2)[:, :h, :w] // 2 + g.integers(0, 128, (n, h, w, 3), dtype=np.uint8))


@pytest.mark.parametrize("H,W,S", [(54, 96, 64), (108, 192, 56), (540, 960, 448)])
def test_window_preprocess_bit_exact_vs_oracle(H, W, S):
    n = 40 if S < 448 else 20
    frames = _stream(n, H, W, 1)
    z, meta = load_golden("tiny_even")
    cfg = cfg_for_case(meta, "fp32")
    cfg.NUM_GPUS = 1
    swc = SlidingWindowClassifier.__new__(SlidingWindowClassifier)
    swc.frame_length, swc.frame_size, swc.mean, swc.std = 16, S, 0.45, 0.225
    windows = [(0, 64), (16, 80)] if S < 448 else [(0, 64)]
    out = swc.preprocess(torch.from_numpy(frames).cuda(), windows).cpu().numpy()
    for k, (t0, t1) in enumerate(windows):
        ref = WO.preprocess_window(frames, WO.frame_idxs_uniform(t0, t1, 16, n), S)
        assert out[k].shape == ref.shape == (3, 16, S, S)
        # the integer (uint8) stage must be bit-exact; the final float normalisation may differ by an ulp of the division
        assert np.array_equal(np.rint((out[k] * 0.225 + 0.45) * 255).astype(np.int32), np.rint((ref * 0.225 + 0.45) * 255).astype(np.int32))
        assert np.abs(out[k] - ref).max() <= 2.4e-7


def test_sliding_window_end_to_end_matches_per_window_forward():
    z, meta = load_golden("tiny_even")        # crop 64, 4 frames
    cfg = cfg_for_case(meta, "fp32")
    cfg.NUM_GPUS = 1
    model = build_model(cfg).eval()
    load_synth_weights(model, 0)
    frames = torch.from_numpy(_stream(50, 54, 96, 2)).cuda()
    swc = SlidingWindowClassifier(model, frame_length=4, frame_stride=4, proposal_length=16, proposal_stride=8, frame_size=64,
                                  batch_size=3)
    res = swc.run(frames)
    wins = get_proposals(50, 16, 8)
    assert [(r[0], r[1]) for r in res] == wins and len(res) == 7
    for t0, t1, p in res:
        assert p.dtype == np.float32 and p.shape == (18,) and abs(p.sum() - 1.0) < 1e-5
        clip = swc.preprocess(frames, [(t0, t1)])
        with torch.no_grad():
            ref = model([clip])[0].cpu().numpy()
        assert np.abs(ref - p).max() <= 1e-6
