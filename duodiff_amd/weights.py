"""Parameter inventory (= checkpoint schema) and seeded synthetic weights.

The names and shapes follow the state_dict of reference ``UViT.__init__``
(models/uvit.py:228-336), so that reference checkpoints
load by name.  No trained checkpoints exist offline, so benchmarks and parity
fixtures use ``synthetic_state_dict``: every tensor -- including biases and
LayerNorm gamma/beta, which the reference's init zeroes -- is drawn from a seeded
torch-CPU generator (recipe in SURVEY section 8d).
"""
from collections import OrderedDict

import torch

from .config import ModelParams


def _block(prefix, D, hidden, skip, qkv_bias=False):
    s = OrderedDict()
    s[prefix + "norm1.weight"] = (D,)
    s[prefix + "norm1.bias"] = (D,)
    s[prefix + "attn.qkv.weight"] = (3 * D, D)
    if qkv_bias:                                     # nn.Linear(dim, 3 dim, bias=qkv_bias), models/uvit.py:150
        s[prefix + "attn.qkv.bias"] = (3 * D,)
    s[prefix + "attn.proj.weight"] = (D, D)
    s[prefix + "attn.proj.bias"] = (D,)
    s[prefix + "norm2.weight"] = (D,)
    s[prefix + "norm2.bias"] = (D,)
    s[prefix + "mlp.fc1.weight"] = (hidden, D)
    s[prefix + "mlp.fc1.bias"] = (hidden,)
    s[prefix + "mlp.fc2.weight"] = (D, hidden)
    s[prefix + "mlp.fc2.bias"] = (D,)
    if skip:
        s[prefix + "skip_linear.weight"] = (D, 2 * D)
        s[prefix + "skip_linear.bias"] = (D,)
    return s


def param_shapes(mp: ModelParams) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor in the reference state_dict."""
    D, P, C = mp.embed_dim, mp.patch_size, mp.in_chans
    hidden = int(D * mp.mlp_ratio)
    s = OrderedDict()
    s["pos_embed"] = (1, mp.seq_len, D)
    s["patch_embed.proj.weight"] = (D, C, P, P)
    s["patch_embed.proj.bias"] = (D,)
    if mp.mlp_time_embed:                            # Linear(D, 4D) -> SiLU -> Linear(4D, D), models/uvit.py:264-272
        s["time_embed.0.weight"] = (4 * D, D)
        s["time_embed.0.bias"] = (4 * D,)
        s["time_embed.2.weight"] = (D, 4 * D)
        s["time_embed.2.bias"] = (D,)
    if mp.num_classes > 0:
        s["label_emb.weight"] = (mp.num_classes, D)
    for i in range(mp.depth // 2):
        s.update(_block(f"in_blocks.{i}.", D, hidden, False, mp.qkv_bias))
    s.update(_block("mid_block.", D, hidden, False, mp.qkv_bias))
    for i in range(mp.depth // 2):
        s.update(_block(f"out_blocks.{i}.", D, hidden, True, mp.qkv_bias))
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    s["decoder_pred.weight"] = (mp.patch_dim, D)
    s["decoder_pred.bias"] = (mp.patch_dim,)
    s["final_layer.weight"] = (C, C, 3, 3)
    s["final_layer.bias"] = (C,)
    return s


def num_params(mp: ModelParams) -> int:
    n = 0
    for shp in param_shapes(mp).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n


def synthetic_state_dict(mp: ModelParams, seed: int = 1234, weight_std: float = 0.02) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 state_dict: W ~ N(0, std^2), b ~ N(0, 0.02^2), LN gamma ~ 1+N(0, 0.1^2).

    ``final_layer`` (3x3 conv) uses std 0.2 so the conv is not a near-null map.
    Draw order is the ``param_shapes`` order; the stream is torch-CPU mt19937.
    """
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd = OrderedDict()
    for name, shp in param_shapes(mp).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif name.startswith("final_layer"):
            t = 0.2 * torch.randn(shp, generator=g)
        else:
            t = weight_std * torch.randn(shp, generator=g)
        sd[name] = t.to(torch.float32).contiguous()
    return sd


# ---- early-exit baseline (reference models/early_exit.py:193-268) ------------------------------
EE_CLASSIFIER_TYPES = ("attention_probe", "mlp_probe_per_layer", "mlp_probe_per_timestep", "mlp_probe_per_layer_per_timestep")


def ee_probe_keys(mp: ModelParams, classifier_type: str):
    """ModuleDict keys of ``EarlyExitUViT.matrix`` (early_exit.py:219-240)."""
    if classifier_type in ("mlp_probe_per_layer", "attention_probe"):
        return [f"{i}" for i in range(mp.depth)]
    if classifier_type == "mlp_probe_per_timestep":
        return [f"{t}" for t in range(1000)]
    if classifier_type == "mlp_probe_per_layer_per_timestep":
        return [f"{i}, {t}" for t in range(1000) for i in range(mp.depth)]
    raise ValueError(f"Unknown classifier type: {classifier_type}")        # early_exit.py:204


def ee_head_prefixes(mp: ModelParams):
    """Output heads in layer order: the head applied BEFORE block i (early_exit.py:290-313)."""
    half = mp.depth // 2
    return ([f"in_blocks_heads.{i}." for i in range(half)] + ["mid_block_head."] +
            [f"out_blocks_heads.{i}." for i in range(half)])


def ee_param_shapes(mp: ModelParams, classifier_type: str = "mlp_probe_per_layer") -> "OrderedDict[str, tuple]":
    """state_dict schema of the reference ``EarlyExitUViT``: ``uvit.*`` + probes + per-layer output heads."""
    D, C = mp.embed_dim, mp.in_chans
    s = OrderedDict(("uvit." + k, v) for k, v in param_shapes(mp).items())
    for key in ee_probe_keys(mp, classifier_type):
        if classifier_type == "attention_probe":                           # AttentionProbe, early_exit.py:46-58 (num_heads = 1)
            s[f"matrix.{key}.q"] = (1, 1, 1, D)
            s[f"matrix.{key}.weight_kv.weight"] = (2 * D, D)
            s[f"matrix.{key}.weight_kv.bias"] = (2 * D,)
            s[f"matrix.{key}.classification.0.weight"] = (D, D)
            s[f"matrix.{key}.classification.0.bias"] = (D,)
            s[f"matrix.{key}.classification.2.weight"] = (1, D)
            s[f"matrix.{key}.classification.2.bias"] = (1,)
        else:
            s[f"matrix.{key}.classifier.0.weight"] = (1, D)
            s[f"matrix.{key}.classifier.0.bias"] = (1,)
    for p in ee_head_prefixes(mp):
        s[p + "norm.weight"] = (D,)
        s[p + "norm.bias"] = (D,)
        s[p + "decoder_pred.weight"] = (mp.patch_dim, D)
        s[p + "decoder_pred.bias"] = (mp.patch_dim,)
        s[p + "final_layer.weight"] = (C, C, 3, 3)
        s[p + "final_layer.bias"] = (C,)
    return s


def synthetic_ee_state_dict(mp: ModelParams, seed: int = 1234, classifier_type: str = "mlp_probe_per_layer"):
    """Seeded EarlyExitUViT weights: the U-ViT part is ``synthetic_state_dict(mp, seed)``; probes get
    w ~ N(0, (4/sqrt(D))^2), b ~ N(0, 1) so that the uncertainty estimates spread over (0, 1); heads as the U-ViT head."""
    sd = OrderedDict(("uvit." + k, v) for k, v in synthetic_state_dict(mp, seed).items())
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 7919)
    D = mp.embed_dim
    for name, shp in ee_param_shapes(mp, classifier_type).items():
        if name.startswith("uvit."):
            continue
        if name.startswith("matrix.") and classifier_type == "attention_probe":
            # q ~ N(0, 4), weights ~ N(0, (2/sqrt(D))^2): scores spread enough for a non-uniform softmax, outputs O(1)
            t = 2.0 * torch.randn(shp, generator=g) if name.endswith(".q") else \
                (2.0 / D ** 0.5) * torch.randn(shp, generator=g) if name.endswith("weight") else 0.3 * torch.randn(shp, generator=g)
        elif name.startswith("matrix."):
            t = (4.0 / D ** 0.5) * torch.randn(shp, generator=g) if name.endswith("weight") else torch.randn(shp, generator=g)
        elif name.endswith("norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif "final_layer" in name:
            t = 0.2 * torch.randn(shp, generator=g)
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        sd[name] = t.to(torch.float32).contiguous()
    return sd
