"""TEST INFRASTRUCTURE ONLY -- runs in the build container, never on the GPU box.

Imports the real reference (``/root/reference``, read-only, Python) so that
(a) the CPU restatement in ``oracle/mvit_oracle.py`` can be validated against it and
(b) golden vectors can be generated (``oracle/make_golden.py``).

The reference's MViT path needs only torch + numpy once six absent third-party
imports are stubbed in ``sys.modules``.  Nothing from
the reference is copied: the stubs below are tiny stand-ins for *third-party* packages
(fvcore / iopath / simplejson / detectron2), not for reference code.
"""
import ast
import copy
import json
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("AICITY_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "slowfast"))


class _Registry(dict):
    """Stand-in for fvcore.common.registry.Registry (name -> object)."""

    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


class _CfgNode(dict):
    """Stand-in for fvcore.common.config.CfgNode (attribute dict + clone)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def dump(self, **kw):
        return json.dumps(self, default=str)


def _simplejson_dumps(obj, sort_keys=False, use_decimal=True, **kw):
    """Stand-in for simplejson.dumps (third party, absent here): as published, ``use_decimal`` serialises decimal.Decimal as a
    bare number literal with the Decimal's own digits (str(d)); everything else as the stdlib encoder does."""
    import decimal

    def enc(o):
        if isinstance(o, decimal.Decimal):
            return str(o)
        if isinstance(o, dict):
            items = sorted(o.items()) if sort_keys else o.items()
            return "{" + ", ".join(json.dumps(str(k)) + ": " + enc(v) for k, v in items) + "}"
        if isinstance(o, (list, tuple)):
            return "[" + ", ".join(enc(v) for v in o) + "]"
        return json.dumps(o)
    return enc(obj)


def _install_stubs():
    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    mod("fvcore")
    mod("fvcore.common")
    mod("fvcore.nn")
    mod("fvcore.common.registry", Registry=_Registry)
    mod("fvcore.common.config", CfgNode=_CfgNode)
    mod("fvcore.nn.weight_init", c2_msra_fill=lambda *a, **k: None)
    mod("iopath")
    mod("iopath.common")

    class _PM:
        def open(self, p, mode="r", **k):
            return open(p, mode)

        def exists(self, p):
            return os.path.exists(p)

        def mkdirs(self, p):
            os.makedirs(p, exist_ok=True)

        def ls(self, p):
            return os.listdir(p)

    class _PMF:
        @staticmethod
        def get(key=None, **k):
            return _PM()

    mod("iopath.common.file_io", PathManagerFactory=_PMF, g_pathmgr=_PM())
    mod("simplejson", dumps=_simplejson_dumps, loads=json.loads)
    mod("detectron2")
    mod("detectron2.layers", ROIAlign=object)


def _merge_yaml(cfg, node):
    for k, v in node.items():
        if isinstance(v, dict):
            _merge_yaml(cfg[k], v)
        else:
            if isinstance(v, str) and v.strip().startswith(("(", "[")):
                try:
                    v = list(ast.literal_eval(v))
                except Exception:
                    pass
            elif isinstance(v, str) and isinstance(cfg.get(k), float):
                v = float(v)  # PyYAML reads "1e-4" as a string; yacs would coerce it
            cfg[k] = v


def load_reference():
    """Returns (get_cfg, build_model, modules-dict). Raises if the reference is absent."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from slowfast.config.defaults import get_cfg  # noqa
    import slowfast.models.video_model_builder  # noqa  (registers MViT)
    from slowfast.models.build import build_model  # noqa
    return get_cfg, build_model


def reference_cfg(yaml_name=None, overrides=None):
    """Fresh reference cfg from configs/Aicity/<yaml_name> with NUM_GPUS=0 + overrides (dict of dotted keys)."""
    import yaml
    get_cfg, _ = load_reference()
    cfg = get_cfg()
    if yaml_name is not None:
        with open(os.path.join(REFERENCE_ROOT, "configs", "Aicity", yaml_name), encoding="utf-8", errors="replace") as f:
            _merge_yaml(cfg, yaml.safe_load(f))
    cfg.NUM_GPUS = 0
    for k, v in (overrides or {}).items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def build_reference_model(cfg):
    _, build_model = load_reference()
    return build_model(cfg)
