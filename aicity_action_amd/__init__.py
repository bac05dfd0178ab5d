"""MI355X-native MViTv2 clip classifier (hot path of JunweiLiang/aicity_action)."""
from .config import CfgNode, get_cfg, load_config  # noqa: F401
