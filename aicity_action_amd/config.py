"""Attribute-style config node for the MViT hot path.

Mirrors the subset of the reference's fvcore ``CfgNode`` tree that the MViT path and its
train/test step read (slowfast/config/defaults.py:12-1136; MVIT keys :404-498, fork additions
:485-498), so that ``configs/Aicity/*.yaml`` and ``KEY VALUE`` CLI overrides load unchanged
(slowfast/utils/parser.py:70-98).  Any attribute-style cfg (incl. a real fvcore CfgNode) is
accepted by ``build_model``; this node exists so the GPU box needs neither fvcore nor yacs.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        """YAML text, as stored under the ``cfg`` key of a .pyth checkpoint (utils/checkpoint.py:131)."""
        return yaml.safe_dump(self.to_dict(), **kwargs)

    # -- merging -------------------------------------------------------------------------
    @staticmethod
    def _coerce(value, ref):
        if isinstance(value, str):
            s = value.strip()
            if isinstance(ref, str):
                return value
            try:
                value = ast.literal_eval(s)
            except (ValueError, SyntaxError):
                return value
        if isinstance(value, tuple):
            value = list(value)
        if isinstance(ref, float) and isinstance(value, int) and not isinstance(value, bool):
            value = float(value)
        return value

    def _merge(self, other, path=""):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = self._coerce(v, self.get(k))

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_list(self, opts):
        """['A.B', value, 'C', value, ...] as given on the reference CLI (parser.py:81-84)."""
        if opts is None:
            return
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            node[parts[-1]] = self._coerce(value, node.get(parts[-1]))


def _defaults():
    # values = slowfast/config/defaults.py defaults (line numbers in the module docstring)
    return {
        "TRAIN": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 64, "EVAL_PERIOD": 10,
                  "CHECKPOINT_PERIOD": 10, "AUTO_RESUME": True, "CHECKPOINT_FILE_PATH": "",
                  "CHECKPOINT_EPOCH_RESET": False, "MIXED_PRECISION": False},
        "TEST": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 8, "CHECKPOINT_FILE_PATH": "",
                 "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3},
        "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400,
                  "LOSS_FUNC": "cross_entropy", "DROPOUT_RATE": 0.5, "HEAD_ACT": "softmax",
                  "USE_HEAD_ACT_IN_TRAIN": False, "ACT_CHECKPOINT": False, "USE_MULTI_HEAD": False,
                  "MULTI_USE_MOCO": False, "USE_VICREG_LOSS": False},
        "MVIT": {"MODE": "conv", "POOL_FIRST": False, "CLS_EMBED_ON": True, "PATCH_KERNEL": [3, 7, 7],
                 "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [2, 4, 4], "PATCH_2D": False,
                 "EMBED_DIM": 96, "NUM_HEADS": 1, "MLP_RATIO": 4.0, "QKV_BIAS": True,
                 "DROPPATH_RATE": 0.1, "DEPTH": 16, "NORM": "layernorm", "DIM_MUL": [], "HEAD_MUL": [],
                 "POOL_KV_STRIDE": None, "POOL_KV_STRIDE_ADAPTIVE": None, "POOL_Q_STRIDE": [],
                 "POOL_KVQ_KERNEL": None, "ZERO_DECAY_POS_CLS": True, "NORM_STEM": False,
                 "SEP_POS_EMBED": False, "DROPOUT_RATE": 0.0, "DIRECT_INPUT": False,
                 "Q_POOL_RESIDUAL": False, "Q_POOL_ALL": False, "CHANNEL_EXPAND_FRONT": False,
                 "POOL_SKIP_USE_CONV": False, "NO_NORM_BEFORE_AVG": False},
        "DATA": {"NUM_FRAMES": 8, "SAMPLING_RATE": 8, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256,
                 "INPUT_CHANNEL_NUM": [3, 3], "MEAN": [0.45, 0.45, 0.45], "STD": [0.225, 0.225, 0.225],
                 "MULTI_LABEL": False, "ENSEMBLE_METHOD": "sum"},
        "CONTRA": {"ENABLE": False},
        "DETECTION": {"ENABLE": False, "USE_CUBE_PROP": False, "USE_SPATIAL_MAXPOOL_BEFORE_PROJ": False},
        "SOLVER": {"BASE_LR": 0.1, "LR_POLICY": "cosine", "COSINE_END_LR": 0.0, "MAX_EPOCH": 300,
                   "MOMENTUM": 0.9, "DAMPENING": 0.0, "NESTEROV": True, "WEIGHT_DECAY": 1e-4,
                   "WARMUP_FACTOR": 0.1, "WARMUP_EPOCHS": 0.0, "WARMUP_START_LR": 0.01,
                   "OPTIMIZING_METHOD": "sgd", "BASE_LR_SCALE_NUM_SHARDS": False,
                   "COSINE_AFTER_WARMUP": False, "ZERO_WD_1D_PARAM": False, "CLIP_GRAD_VAL": None,
                   "CLIP_GRAD_L2NORM": None},
        "NUM_GPUS": 1, "NUM_SHARDS": 1, "SHARD_ID": 0, "OUTPUT_DIR": "./tmp", "RNG_SEED": 1, "LOG_PERIOD": 100,
        "DIST_BACKEND": "nccl",
        # build-specific knob (not in the reference): arithmetic of the HIP path, "bf16" or "fp32"
        # DDP_*: how build_model wraps DistributedDataParallel (gradients live in the all-reduce buckets, static graph; optional
        # bf16 gradient payload: 70.6 MB instead of 141 MB per step over xGMI).  STAT_QUEUE_DEPTH: iterations the training loop may
        # run ahead of the GPU before it waits for a step's scalars (meters.DeviceScalarQueue).  GRAPH_STEP: engine.train_epoch
        # captures the whole train step in a hipGraph after two eager iterations and replays it (graph_step.py)
        "HIP": {"PRECISION": "auto", "STREAMS": 3, "TRAIN_STREAMS": 1, "WGRAD_STREAM": True, "DDP_BUCKET_VIEW": True,
                "DDP_STATIC_GRAPH": True, "DDP_BF16_GRADS": False, "STAT_QUEUE_DEPTH": 2, "GRAPH_STEP": False,
                "REL_POS_BIAS": False},
    }


def get_cfg():
    """Fresh default tree (slowfast/config/defaults.py:1167-1171)."""
    return CfgNode(_defaults())


def load_config(cfg_file=None, opts=None):
    """get_cfg -> merge_from_file -> merge_from_list (slowfast/utils/parser.py:70-98)."""
    cfg = get_cfg()
    if cfg_file:
        cfg.merge_from_file(cfg_file)
    cfg.merge_from_list(opts)
    return cfg
