"""Oracle (test infrastructure): the same U-ViT forward restated over torch-CPU functional ops.

Why a second statement: the numpy oracle (uvit_oracle.py) is the independent checker but its
elementwise ops are single-threaded; the reference itself runs ATen kernels (addmm, gelu,
layer_norm, scaled_dot_product_attention, conv2d -- SURVEY section 3.2).  This variant calls
those same ATen ops functionally (no nn.Module, name-keyed params), so timing it on the host
cores is a fair stand-in for "the reference's CPU sampler" on a box where the reference
itself cannot travel.  Used by bench.py's cpu_baseline leg and pinned by the same golden
vectors (tests/test_oracle_golden.py).  Follows reference models/uvit.py:95-115, 125-132,
155-168, 86-92, 203-208, 221-225, 351-383.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def block(x, p, pre, heads, skip=None):
    if skip is not None:
        x = F.linear(torch.cat([x, skip], dim=-1), p[pre + "skip_linear.weight"], p[pre + "skip_linear.bias"])
    B, L, D = x.shape
    h = F.layer_norm(x, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
    qkv = F.linear(h, p[pre + "attn.qkv.weight"], p.get(pre + "attn.qkv.bias")).reshape(B, L, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).permute(0, 2, 1, 3).reshape(B, L, D)
    x = x + F.linear(a, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
    return x + F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


class UViTTorchOracle:
    def __init__(self, cfg, params):
        self.cfg = dict(cfg)
        self.p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in params.items()}
        self.depth, self.heads = int(cfg["depth"]), int(cfg["num_heads"])
        self.C, self.P = int(cfg["in_chans"]), int(cfg["patch_size"])
        self.extras = 2 if int(cfg["num_classes"]) > 0 else 1
        self.normalize = bool(cfg["normalize_timesteps"])
        self.calls = 0

    @torch.no_grad()
    def __call__(self, x, timesteps, y=None):
        self.calls += 1
        p = self.p
        x = torch.as_tensor(x, dtype=torch.float32)
        t = torch.as_tensor(timesteps).float()
        if self.normalize:
            t = t / 1000
        tok = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=self.P).flatten(2).transpose(1, 2)
        D = tok.shape[-1]
        tt = timestep_embedding(t, D)
        if "time_embed.0.weight" in p:                           # mlp_time_embed=True (models/uvit.py:264-272)
            tt = F.linear(F.silu(F.linear(tt, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                          p["time_embed.2.weight"], p["time_embed.2.bias"])
        tok = torch.cat([tt[:, None], tok], dim=1)
        if y is not None:
            tok = torch.cat([p["label_emb.weight"][torch.as_tensor(y).long()][:, None], tok], dim=1)
        h = tok + p["pos_embed"]
        skips = []
        for i in range(self.depth // 2):
            h = block(h, p, f"in_blocks.{i}.", self.heads)
            skips.append(h)
        h = block(h, p, "mid_block.", self.heads)
        for i in range(self.depth // 2):
            h = block(h, p, f"out_blocks.{i}.", self.heads, skips.pop())
        h = F.layer_norm(h, (D,), p["norm.weight"], p["norm.bias"], 1e-5)
        h = F.linear(h, p["decoder_pred.weight"], p["decoder_pred.bias"])[:, self.extras:]
        B, N, _ = h.shape
        g = int(round(N ** 0.5))
        img = h.reshape(B, g, g, self.P, self.P, self.C).permute(0, 5, 1, 3, 2, 4).reshape(B, self.C, g * self.P, g * self.P)
        return F.conv2d(img, p["final_layer.weight"], p["final_layer.bias"], padding=1).numpy()
