"""Oracle (test infrastructure): U-ViT noise-predictor forward in numpy fp32.

A functional restatement of reference ``models/uvit.py`` (inference path only);
parameters are a plain dict keyed by the reference's state_dict names
(``UViT.__init__`` models/uvit.py:228-336), so a reference checkpoint drops in.

Every function cites the reference lines it follows.  Pinned against golden
vectors generated from the reference itself (tests/golden/uvit_tiny_*.npz,
uvit_full_*.npz); see oracle/gen_golden.py.
"""
import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


def linear(x, w, b=None):
    """nn.Linear: y = x @ w.T + b, fp32 (weights stored [out, in])."""
    y = np.matmul(x, w.T, dtype=F32)
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def timestep_embedding(timesteps, dim, max_period=10000):
    """models/uvit.py:95-115: [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) i / half)."""
    half = dim // 2
    freqs = np.exp((F32(-math.log(max_period)) * np.arange(half, dtype=F32)) / F32(half)).astype(F32)
    args = (np.asarray(timesteps, F32)[:, None] * freqs[None]).astype(F32)
    emb = np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(F32)
    if dim % 2:
        emb = np.concatenate([emb, np.zeros_like(emb[:, :1])], axis=-1)
    return emb


def layer_norm(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm(dim), biased variance, eps 1e-5 (models/uvit.py:185,189,326)."""
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    y = (x64 - mu) / np.sqrt(var + eps)
    return (y.astype(F32) * gamma + beta).astype(F32)


def gelu_erf(x):
    """nn.GELU() default = exact erf form (models/uvit.py:75,82)."""
    x = x.astype(F32, copy=False)
    return (F32(0.5) * x * (F32(1) + _erf(x * F32(1.0 / math.sqrt(2.0))).astype(F32))).astype(F32)


def attention(x, p, prefix, num_heads):
    """models/uvit.py:155-168: qkv (no bias unless present) -> fp32 SDPA, scale 1/sqrt(hd) -> proj."""
    B, L, C = x.shape
    hd = C // num_heads
    qkv = linear(x, p[prefix + "qkv.weight"], p.get(prefix + "qkv.bias"))
    # "B L (K H D) -> K B H L D"
    qkv = qkv.reshape(B, L, 3, num_heads, hd).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = np.matmul(q, k.transpose(0, 1, 3, 2), dtype=F32) * F32(1.0 / math.sqrt(hd))
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s, dtype=F32)
    pr = (e / e.sum(-1, keepdims=True, dtype=F32)).astype(F32)
    o = np.matmul(pr, v, dtype=F32)                          # B H L D
    o = o.transpose(0, 2, 1, 3).reshape(B, L, C)             # "B H L D -> B L (H D)"
    return linear(o, p[prefix + "proj.weight"], p[prefix + "proj.bias"])


def mlp(x, p, prefix):
    """models/uvit.py:86-92: fc2(GELU(fc1(x)))."""
    h = gelu_erf(linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"]))
    return linear(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])


def block_forward(x, p, prefix, num_heads, skip=None):
    """models/uvit.py:203-208."""
    if skip is not None:
        x = linear(np.concatenate([x, skip], axis=-1),
                   p[prefix + "skip_linear.weight"], p[prefix + "skip_linear.bias"])
    x = x + attention(layer_norm(x, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"]),
                      p, prefix + "attn.", num_heads)
    x = x + mlp(layer_norm(x, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]),
                p, prefix + "mlp.")
    return x.astype(F32, copy=False)


def patch_embed(x, w, b):
    """models/uvit.py:221-225: conv k=s=P == per-patch GEMM; tokens row-major over (h, w)."""
    B, C, H, W = x.shape
    D, _, P, _ = w.shape
    gh, gw = H // P, W // P
    patches = x.reshape(B, C, gh, P, gw, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
    return linear(patches.astype(F32), w.reshape(D, C * P * P), b)


def unpatchify(x, channels):
    """models/uvit.py:125-132: "B (h w) (p1 p2 C) -> B C (h p1) (w p2)" (channel fastest in a patch)."""
    B, N, pd = x.shape
    P = int(round((pd // channels) ** 0.5))
    g = int(round(N ** 0.5))
    assert g * g == N and P * P * channels == pd
    return x.reshape(B, g, g, P, P, channels).transpose(0, 5, 1, 3, 2, 4).reshape(B, channels, g * P, g * P)


def conv3x3(x, w, b):
    """nn.Conv2d(C, C, 3, padding=1) (models/uvit.py:329-333, 382), cross-correlation."""
    B, C, H, W = x.shape
    xp = np.zeros((B, C, H + 2, W + 2), F32)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, w.shape[0], H, W), F32)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("bchw,oc->bohw", xp[:, :, dy:dy + H, dx:dx + W], w[:, :, dy, dx]).astype(F32)
    return (out + b[None, :, None, None]).astype(F32)


class UViTOracle:
    """Inference-only U-ViT (models/uvit.py:228-383) over a name-keyed fp32 param dict.

    ``cfg`` is the YAML ``model_params`` dict (unknown keys ignored, quirk Q3).
    ``taps`` (optional dict) receives intermediate tensors for per-op parity tests.
    """

    def __init__(self, cfg, params):
        self.img_size = int(cfg["img_size"])
        self.patch_size = int(cfg["patch_size"])
        self.in_chans = int(cfg["in_chans"])
        self.embed_dim = int(cfg["embed_dim"])
        self.depth = int(cfg["depth"])
        self.num_heads = int(cfg["num_heads"])
        self.num_classes = int(cfg["num_classes"])
        self.normalize_timesteps = bool(cfg["normalize_timesteps"])
        self.mlp_time_embed = bool(cfg.get("mlp_time_embed", False))
        self.extras = 2 if self.num_classes > 0 else 1
        self.p = {k: np.ascontiguousarray(np.asarray(v, F32)) for k, v in params.items()}
        self.calls = 0

    def __call__(self, x, timesteps, y=None, taps=None):
        self.calls += 1
        p = self.p
        x = np.asarray(x, F32)
        t = np.asarray(timesteps, F32)
        if self.normalize_timesteps:                         # uvit.py:352-353
            t = (t / F32(1000)).astype(F32)
        tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"])  # :355
        time_token = timestep_embedding(t, self.embed_dim)                               # :358
        if self.mlp_time_embed:                                                          # :264-272 Linear -> SiLU -> Linear
            hmid = linear(time_token, p["time_embed.0.weight"], p["time_embed.0.bias"])
            hmid = (hmid / (F32(1) + np.exp(-hmid, dtype=F32))).astype(F32)
            time_token = linear(hmid, p["time_embed.2.weight"], p["time_embed.2.bias"])
        time_token = time_token[:, None, :]
        tok = np.concatenate([time_token, tok], axis=1)                                  # :360
        if y is not None:                                                                # :361-364
            tok = np.concatenate([p["label_emb.weight"][np.asarray(y)][:, None, :], tok], axis=1)
        h = (tok + p["pos_embed"]).astype(F32)                                           # :365
        if taps is not None:
            taps["time_token"] = time_token[:, 0]
            taps["tokens"] = h
        skips = []
        for i in range(self.depth // 2):                                                 # :367-370
            h = block_forward(h, p, f"in_blocks.{i}.", self.num_heads)
            skips.append(h)
            if taps is not None:
                taps[f"in_blocks.{i}"] = h
        h = block_forward(h, p, "mid_block.", self.num_heads)                            # :372
        if taps is not None:
            taps["mid_block"] = h
        for i in range(self.depth // 2):                                                 # :374-375
            h = block_forward(h, p, f"out_blocks.{i}.", self.num_heads, skip=skips.pop())
            if taps is not None:
                taps[f"out_blocks.{i}"] = h
        h = layer_norm(h, p["norm.weight"], p["norm.bias"])                              # :377
        h = linear(h, p["decoder_pred.weight"], p["decoder_pred.bias"])                  # :378
        h = h[:, self.extras:, :]                                                        # :379-380
        if taps is not None:
            taps["decoder_pred"] = h
        img = unpatchify(h, self.in_chans)                                               # :381
        return conv3x3(img, p["final_layer.weight"], p["final_layer.bias"])              # :382
