#!/usr/bin/env python3
"""Fused MLP kernel vs the two-GEMM path on the GPU: parity of eps (same weights, same inputs) and in-context timing.

    python tools/mlp_check.py [--config uvit_celeba_3] [--batches 2 5 128] [--num_cus 0]
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.uvit import UViT  # noqa: E402
from duodiff_amd.weights import synthetic_state_dict  # noqa: E402


def build(mp, sd, fused, max_batch):
    from duodiff_amd import _lib
    from duodiff_amd.engine import Context
    ctx = Context.get()
    ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0 if fused else _lib.DD_DEV_NO_FUSED_MLP))
    m = UViT(**mp.as_dict(), precision="bf16", max_batch=max_batch).load_state_dict(sd).to("cuda")
    m.engine_model(max_batch)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="uvit_celeba_3")
    ap.add_argument("--batches", type=int, nargs="+", default=[2, 5, 128])
    ap.add_argument("--num_cus", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--time_only", action="store_true", help="skip the unfused model (ablation libraries compute garbage)")
    a = ap.parse_args()
    mp = ModelParams.from_dict(load_config(REPO / "configs" / f"{a.config}.yaml"))
    sd = synthetic_state_dict(mp, 1237)
    from duodiff_amd.engine import Context
    ctx = Context.get()
    if a.num_cus:
        ctx.check(ctx.lib.dd_set_num_cus(ctx.handle, a.num_cus))
    bmax = max(a.batches)
    mf = build(mp, sd, True, bmax)
    mu = None if a.time_only else build(mp, sd, False, bmax)
    for B in a.batches:
        x = torch.randn(B, mp.in_chans, mp.img_size, mp.img_size, generator=torch.Generator().manual_seed(B)).cuda()
        t = torch.full((B,), 417.0)
        y = torch.randint(0, mp.num_classes, (B,)) if mp.num_classes > 0 else None
        if mu is not None:
            ef, eu = mf(x, t, y), mu(x, t, y)
            torch.cuda.synchronize()
            d = (ef - eu).abs()
            print(f"B={B}: fused vs unfused max|d|={d.max().item():.3e} rms={d.pow(2).mean().sqrt().item():.3e} "
                  f"eps std={eu.std().item():.3f} finite={bool(torch.isfinite(ef).all())}", flush=True)
        else:
            print(f"B={B}:", flush=True)
        xs = x.clone()
        for name, m in (("fused", mf), ("unfused", mu)):
            if m is None:
                continue
            em = m.engine_model(B)
            ms, n = em.profile_steps(xs.clone(), t_start=699, steps=a.steps)
            print(f"   {name}: dominant-kernel launch {ms * 1e3:.1f} us over {n} launches", flush=True)


if __name__ == "__main__":
    main()
