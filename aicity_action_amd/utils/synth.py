"""Deterministic synthetic weights and clips (numpy PCG64, keyed by tensor name).

A full state_dict is 138 MB, so parity fixtures cannot carry weights: both sides of a parity
test (oracle / reference in the build container, HIP path on the GPU box) regenerate them from
this generator instead.  Distributions follow the reference's init
(video_model_builder.py:1046-1055,1126-1133: Linear/pos-embed ~ truncN(0, .02); convs keep torch's
kaiming-uniform bound 1/sqrt(fan_in)) except that biases and LayerNorm affine parameters are
*also* randomised (the reference zero/one-initialises them) so that every term of the
arithmetic is exercised.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def synth_tensor(name, shape, seed=0):
    g = _rng(name, seed)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if name.startswith("pos_embed"):
        a = np.clip(g.standard_normal(shape) * 0.02, -2, 2)
    elif parent.startswith("norm"):           # LayerNorm affine
        a = 1.0 + 0.1 * g.standard_normal(shape) if leaf == "weight" else 0.05 * g.standard_normal(shape)
    elif len(shape) == 5:                      # Conv3d weight
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        a = g.uniform(-b, b, shape)
    elif name == "patch_embed.proj.bias":
        b = 1.0 / np.sqrt(3 * 3 * 7 * 7)
        a = g.uniform(-b, b, shape)
    elif leaf == "weight":                     # Linear weight
        a = np.clip(g.standard_normal(shape) * 0.02, -2, 2)
    else:                                      # Linear bias
        a = g.standard_normal(shape) * 0.02
    return torch.from_numpy(a.astype(np.float32))


def synth_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. from model.state_dict()). Returns name -> fp32 tensor."""
    return {k: synth_tensor(k, tuple(v), seed) for k, v in shapes.items()}


def load_synth_weights(model, seed=0):
    sd = model.state_dict()
    new = synth_state_dict({k: v.shape for k, v in sd.items()}, seed)
    model.load_state_dict(new)
    return model


def synth_clip(batch, frames, size, seed=1, channels=3):
    """[B,3,T,S,S] fp32 ~ N(0,1): the normalised-pixel domain the model sees ((x/255-.45)/.225)."""
    g = np.random.Generator(np.random.PCG64([0xC11B, seed]))
    return torch.from_numpy(g.standard_normal((batch, channels, frames, size, size), dtype=np.float32))


def stress_state_dict(sd, linear_gain=4.0, qk_gain=2.5):
    """A "trained-like" variant of a synthetic state dict: every Linear weight of the blocks and of the head x linear_gain (activations and logits of
    O(1-10) instead of the O(0.1) of a random initialisation) and the LayerNorm gains of the pooled q and k x qk_gain (attention
    scores x qk_gain^2: rows with a dominant key).  Used by the statistical logit gate (oracle/make_golden_gate.py and its test)."""
    out = {}
    for k, v in sd.items():
        if (k.startswith("blocks.") or k == "head.projection.weight") and k.endswith(".weight") and v.dim() == 2:
            out[k] = v * linear_gain
        elif k.endswith("attn.norm_q.weight") or k.endswith("attn.norm_k.weight"):
            out[k] = v * qk_gain
        else:
            out[k] = v.clone()
    return out
