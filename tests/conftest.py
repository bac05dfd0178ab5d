import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, "mvit_%s.npz" % name))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def cfg_for_case(meta, precision="fp32", train=False):
    """Build this repo's cfg for a golden case from its recorded yaml + overrides."""
    from aicity_action_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", meta["yaml"]))
    ov = meta["train_overrides"] if train else meta["overrides"]
    opts = []
    for k, v in ov.items():
        opts += [k, v]
    cfg.merge_from_list(opts)
    cfg.NUM_GPUS = 0
    cfg.HIP.PRECISION = precision
    return cfg


def sample_like(t, mom):
    """Same strided sample as oracle/make_golden.py:sample()."""
    stride = int(mom[2])
    return t.detach().reshape(-1).float().cpu()[::stride].numpy()


@pytest.fixture(scope="session")
def hip_lib():
    from aicity_action_amd import _hip
    _hip.build()
    return _hip.lib()
