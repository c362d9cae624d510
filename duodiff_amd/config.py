"""YAML config loading for the sampling path.

Mirrors reference ``utils/config_utils.py:5-13`` (``load_config``: existence check,
``yaml.safe_load``) and the schema of ``configs/uvit_*.yaml``.  Unlike the
reference's ``UViT(**config["model_params"])`` (sampler.py:297), unknown keys such
as ``classifier_type`` are ignored instead of raising TypeError.
"""
from dataclasses import dataclass
from pathlib import Path

import yaml

_MODEL_KEYS = (
    "img_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads",
    "mlp_ratio", "qkv_bias", "mlp_time_embed", "num_classes", "normalize_timesteps",
)


def load_config(path):
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Config file {path} does not exist")
    with path.open("r") as f:
        return yaml.safe_load(f)


@dataclass(frozen=True)
class ModelParams:
    img_size: int
    patch_size: int
    in_chans: int
    embed_dim: int
    depth: int
    num_heads: int
    mlp_ratio: int = 4
    qkv_bias: bool = False
    mlp_time_embed: bool = False
    num_classes: int = -1
    normalize_timesteps: bool = True

    @classmethod
    def from_dict(cls, d):
        if "model_params" in d:
            d = d["model_params"]
        missing = [k for k in ("img_size", "patch_size", "in_chans", "embed_dim", "depth",
                               "num_heads", "num_classes", "normalize_timesteps") if k not in d]
        if missing:
            raise KeyError(f"model_params is missing {missing}")
        kw = {k: d[k] for k in _MODEL_KEYS if k in d}
        mp = cls(img_size=int(kw["img_size"]), patch_size=int(kw["patch_size"]),
                 in_chans=int(kw["in_chans"]), embed_dim=int(kw["embed_dim"]),
                 depth=int(kw["depth"]), num_heads=int(kw["num_heads"]),
                 mlp_ratio=int(kw.get("mlp_ratio", 4)), qkv_bias=bool(kw.get("qkv_bias", False)),
                 mlp_time_embed=bool(kw.get("mlp_time_embed", False)),
                 num_classes=int(kw["num_classes"]),
                 normalize_timesteps=bool(kw["normalize_timesteps"]))
        mp.validate()
        return mp

    def validate(self):
        if self.img_size % self.patch_size:
            raise ValueError("img_size must be a multiple of patch_size")
        if self.embed_dim % self.num_heads:
            raise ValueError("embed_dim must be a multiple of num_heads")
        if self.depth % 2 != 1:
            raise ValueError("depth must be odd (depth//2 in-blocks, mid, depth//2 out-blocks)")

    # derived sizes
    @property
    def num_patches(self):
        return (self.img_size // self.patch_size) ** 2

    @property
    def extras(self):
        return 2 if self.num_classes > 0 else 1

    @property
    def seq_len(self):
        return self.extras + self.num_patches

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self):
        return self.patch_size ** 2 * self.in_chans

    def as_dict(self):
        return {k: getattr(self, k) for k in _MODEL_KEYS}

    def flops_per_image(self):
        """Algorithmic GEMM FLOPs of one forward for one image."""
        L, D, N = self.seq_len, self.embed_dim, self.num_patches
        pd, C, S = self.patch_dim, self.in_chans, self.img_size
        return (self.depth * (24 * L * D * D + 4 * L * L * D) + (self.depth // 2) * 4 * L * D * D
                + 2 * N * pd * D + 2 * L * D * pd + 2 * C * C * 9 * S * S)
