"""CPU: host-side logic -- cfg loading, block geometry vs the reference-derived table, state_dict layout,
optimizer grouping, registry/build_model semantics, C-ABI exports."""
import ctypes
import json
import os
import re

import pytest
import torch

from conftest import GOLD, ROOT

from aicity_action_amd import _hip
from aicity_action_amd.config import load_config
from aicity_action_amd.models import MODEL_REGISTRY, MViT, build_model
from aicity_action_amd.models.spec import derive_block_geoms

ARITH = json.load(open(os.path.join(GOLD, "mvit_arith.json")))
OURS = [y for y in ARITH if os.path.exists(os.path.join(ROOT, "configs", "Aicity", y))]


def _cfg(y):
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", y))
    cfg.NUM_GPUS = 0
    return cfg


@pytest.mark.parametrize("y", OURS)
def test_block_geometry_matches_reference(y):
    cfg = _cfg(y)
    geoms, kv = derive_block_geoms(cfg)
    ref = ARITH[y]
    assert len(geoms) == len(ref["blocks"])
    assert [list(map(int, e)) for e in kv] == ref["pool_kv_stride"]
    for g, r in zip(geoms, ref["blocks"]):
        assert (g.dim_in, g.dim_out, g.heads) == (r["dim_in"], r["dim_out"], r["heads"])
        assert list(g.stride_q) == r["stride_q"] and list(g.stride_kv) == r["stride_kv"]
        assert g.expand == r["has_pmp"]
        assert abs(g.drop_path - r["drop_path"]) < 1e-7
        if r["skip"] is None:
            assert g.skip_kernel is None
        else:
            assert [list(g.skip_kernel), list(g.skip_stride), list(g.skip_pad)] == r["skip"]
    assert list(geoms[0].thw_in) == ref["patch_dims"]


@pytest.mark.parametrize("y", [y for y in OURS if "16x4" in y])
def test_state_dict_layout_and_param_groups(y):
    cfg = _cfg(y)
    model = build_model(cfg)
    ref = ARITH[y]
    sd = model.state_dict()
    assert list(sd.keys()) == ref["keys"]
    assert [list(v.shape) for v in sd.values()] == ref["shapes"]
    assert all(v.dtype == torch.float32 for v in sd.values())
    assert sum(p.numel() for p in model.parameters()) == ref["n_params"]
    assert model.no_weight_decay() == {}
    # the reference's grouping rule (slowfast/models/optimizer.py:56-75, ZERO_WD_1D_PARAM): 1-D -> no decay
    wd = [p for m in model.modules() for p in m.parameters(recurse=False) if p.ndim > 1]
    nwd = [p for m in model.modules() for p in m.parameters(recurse=False) if p.ndim == 1]
    sizes = dict(zip(ref["group_wd"], ref["group_sizes"]))
    assert len(wd) == sizes[cfg.SOLVER.WEIGHT_DECAY] and len(nwd) == sizes[0.0]
    # construction mutates cfg.MVIT.POOL_KV_STRIDE like the reference (video_model_builder.py:960-967)
    assert [list(map(int, e)) for e in cfg.MVIT.POOL_KV_STRIDE] == ref["pool_kv_stride"]


def test_registry_and_build_model_contract():
    assert MODEL_REGISTRY.get("MViT") is MViT
    with pytest.raises(KeyError):
        MODEL_REGISTRY.get("SlowFast")
    cfg = _cfg("MVITV2_FULL_B_16x4_CONV.yaml")
    if not torch.cuda.is_available():
        cfg.NUM_GPUS = 1
        with pytest.raises(AssertionError):
            build_model(cfg)      # slowfast/models/build.py:29-32
        cfg.NUM_GPUS = 0
    m = build_model(cfg)
    # product path has no CPU fallback
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m([torch.zeros(1, 3, 16, 224, 224)])


def test_cfg_cli_overrides_and_unsupported_branches():
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV.yaml"),
                      ["NUM_GPUS", "0", "SOLVER.BASE_LR", "1e-4", "MVIT.POOL_Q_STRIDE", "[[1, 1, 2, 2]]",
                       "TRAIN.MIXED_PRECISION", "False", "MODEL.LOSS_FUNC", "cross_entropy"])
    assert cfg.NUM_GPUS == 0 and cfg.SOLVER.BASE_LR == 1e-4 and cfg.MVIT.POOL_Q_STRIDE == [[1, 1, 2, 2]]
    assert cfg.TRAIN.MIXED_PRECISION is False and cfg.MODEL.LOSS_FUNC == "cross_entropy"
    assert cfg.SOLVER.WEIGHT_DECAY == 1e-4 and isinstance(cfg.dump(), str)
    cfg2 = _cfg("MVITV2_FULL_B_16x4_CONV.yaml")
    cfg2.MVIT.CLS_EMBED_ON = True
    with pytest.raises(NotImplementedError):
        MViT(cfg2)
    cfg4 = _cfg("MVITV2_FULL_B_16x4_CONV.yaml")
    assert cfg4.HIP.REL_POS_BIAS is False            # the hook north_star asks for: present, off, and refusing to be turned on
    cfg4.HIP.REL_POS_BIAS = True
    with pytest.raises(NotImplementedError):
        MViT(cfg4)
    cfg3 = _cfg("MVITV2_FULL_B_16x4_CONV.yaml")
    cfg3.DATA.TEST_CROP_SIZE = 256
    with pytest.raises(AssertionError):
        MViT(cfg3)


def test_checkpoint_roundtrip_pyth_layout(tmp_path):
    """.pyth dict layout of slowfast/utils/checkpoint.py:127-134 round-trips through load_state_dict."""
    cfg = _cfg("MVITV2_FULL_B_16x4_CONV.yaml")
    m = build_model(cfg)
    ck = {"epoch": 3, "model_state": m.state_dict(), "optimizer_state": {}, "cfg": cfg.dump()}
    p = tmp_path / "checkpoint_epoch_00004.pyth"
    torch.save(ck, str(p))
    m2 = build_model(_cfg("MVITV2_FULL_B_16x4_CONV.yaml"))
    missing, unexpected = m2.load_state_dict(torch.load(str(p), map_location="cpu")["model_state"], strict=False)
    assert not missing and not unexpected
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)


def test_capi_library_builds_and_exports_every_declared_symbol():
    path = _hip.build()
    L = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "mvit_hip.h")).read()
    declared = set(re.findall(r"\b(mvit_[a-z0-9_]+)\s*\(", header))
    assert declared, "no entry points parsed from include/mvit_hip.h"
    for name in declared:
        assert hasattr(L, name), "libmvit_hip.so does not export %s" % name
    assert declared == set(_hip.EXPORTS), (declared ^ set(_hip.EXPORTS))
    assert _hip.lib().mvit_version().decode().startswith("mvit-hip")
    assert _hip.lib().mvit_strerror(-4).decode()


def test_init_weights_distributions():
    """a15: MViT._init_weights / trunc_normal_ (video_model_builder.py:1046-1055,1126-1133).  Every nn.Linear weight and the
    two position embeddings ~ N(0, 0.02) truncated at the ABSOLUTE bounds +-2 (trunc_normal_'s a/b defaults: 100 sigma, so
    effectively untruncated) with zero bias; LayerNorm weight 1 / bias 0; Conv3d (stem, depthwise pools) keep torch's default
    kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)), the stem bias likewise."""
    torch.manual_seed(1234)
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV.yaml"), ["NUM_GPUS", 0])
    m = MViT(cfg)
    lin_w = []
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.Linear):
            w = mod.weight.detach()
            assert w.abs().max().item() <= 2.0 and float(mod.bias.detach().abs().max()) == 0.0, name
            if w.numel() >= 96 * 96:
                assert abs(w.std().item() - 0.02) <= 0.0015 and abs(w.mean().item()) <= 0.002, (name, w.std().item())
            lin_w.append(w.flatten())
        elif isinstance(mod, torch.nn.LayerNorm):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), name
        elif isinstance(mod, torch.nn.Conv3d):
            fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3] * mod.weight.shape[4]
            bound = 1.0 / fan_in ** 0.5
            w = mod.weight.detach()
            assert w.abs().max().item() <= bound + 1e-7, name
            # uniform on [-bound, bound]: std = bound / sqrt(3); checked on the tensors large enough for the estimate
            if w.numel() >= 2000:
                assert abs(w.std().item() / (bound / 3 ** 0.5) - 1.0) <= 0.05, (name, w.std().item(), bound)
            if mod.bias is not None:
                assert mod.bias.detach().abs().max().item() <= bound + 1e-7, name
    allw = torch.cat(lin_w)
    assert abs(allw.std().item() - 0.02) <= 2e-4 and abs(allw.mean().item()) <= 2e-5
    # N(0, 0.02): 4.55 % of the mass beyond 2 sigma, 0.27 % beyond 3 sigma -- a +-2 sigma truncation would show here
    frac2 = (allw.abs() > 0.04).float().mean().item()
    frac3 = (allw.abs() > 0.06).float().mean().item()
    assert abs(frac2 - 0.0455) <= 0.002 and abs(frac3 - 0.0027) <= 0.0005, (frac2, frac3)
    for pe in (m.pos_embed_spatial, m.pos_embed_temporal):
        assert abs(pe.detach().std().item() - 0.02) <= (0.001 if pe.numel() > 10000 else 0.004)
    # the 224 config: 34,415,538 parameters
    assert sum(p.numel() for p in m.parameters()) == 34415538


def test_no_weight_decay_grouping_follows_the_module_name_rule():
    """optimizer.py:56-75 tests the MODULE name against model.no_weight_decay(): with MVIT.ZERO_DECAY_POS_CLS True the
    root-level pos_embed_* parameters (module name "") therefore stay in the decay group, exactly as in the reference, and the
    [decay..., no-decay...] index order of AdamW.state_dict() is unchanged."""
    from aicity_action_amd.solver import param_groups
    for flag in (False, True):
        cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV.yaml"), ["NUM_GPUS", 0, "MVIT.ZERO_DECAY_POS_CLS", flag])
        m = MViT(cfg)
        decay, no_decay = param_groups(m, cfg)
        assert (len(decay), len(no_decay)) == (119, 231)
        assert {"pos_embed_spatial", "pos_embed_temporal"} <= {n for n, _ in decay}


def test_bench_self_launch_builds_the_launcher_command_without_touching_the_gpu(monkeypatch):
    """bench.py --gpus N without WORLD_SIZE: the parent must only start `python -m torch.distributed.run ... bench.py <same args>` as a
    child (--standalone on 127.0.0.1, N ranks) and return its code -- it must not import torch (an exec / fork of a process that has
    initialised HIP takes the machine down on this pool)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    import pytest as _pt
    with _pt.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7                                  # the launcher's return code is bench.py's
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    # the launcher picks and holds its own rendezvous port (no bind-close-reuse race), on 127.0.0.1
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
