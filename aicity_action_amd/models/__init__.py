from .build import MODEL_REGISTRY, build_model  # noqa: F401
from .mvit import MViT  # noqa: F401  (registers "MViT")
