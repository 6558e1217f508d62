from .ffmlp import FFMLP, fused_mlp  # noqa: F401
