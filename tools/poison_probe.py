#!/usr/bin/env python3
"""Which configurations read workspace bytes that no launch of the call wrote?  One forward on a fresh model vs one forward after
dd_dev_poison_workspaces (every activation buffer filled with NaN bytes on the launch stream), per configuration and dev-flag set.

    python tools/poison_probe.py
"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from duodiff_amd import _lib as L  # noqa: E402
from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.uvit import UViT  # noqa: E402
from duodiff_amd.weights import synthetic_state_dict  # noqa: E402

TINY = dict(img_size=8, patch_size=2, in_chans=3, embed_dim=64, depth=3, num_heads=1, mlp_ratio=4, qkv_bias=False,
            mlp_time_embed=False, num_classes=-1, normalize_timesteps=True)


def one(name, cfg, B, prec, flags=0):
    mp = ModelParams.from_dict(cfg)
    outs = []
    for poison in (False, True):
        from duodiff_amd.engine import Context
        ctx = Context.get()
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
        m = UViT(**mp.as_dict(), precision=prec, max_batch=B).load_state_dict(synthetic_state_dict(mp, 5)).to("cuda")
        em = m.engine_model(B)
        if poison:
            ctx.check(ctx.lib.dd_dev_poison_workspaces(ctx.handle, em.handle, torch.cuda.current_stream().cuda_stream))
        x = torch.randn(B, mp.in_chans, mp.img_size, mp.img_size, generator=torch.Generator().manual_seed(1))
        y = torch.randint(0, mp.num_classes, (B,), generator=torch.Generator().manual_seed(2)) if mp.num_classes > 0 else None
        outs.append(m(x, torch.full((B,), 417.0), y).cpu())
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
        del m, em
    same = torch.equal(outs[0], outs[1])
    nanf = float(torch.isnan(outs[1]).float().mean())
    print(f"{name:44s} B={B:3d} {prec} flags={flags:5d}: {'ok' if same else 'DIFFERS'}  (NaN fraction after poison {nanf:.3f}, clean finite {bool(torch.isfinite(outs[0]).all())})", flush=True)


def loop(name, cfg_s, cfg_f, B, prec, flags, steps=8, tsw=3, use_graph=True):
    """the sampling loop (dd_sample) on a fresh model pair vs after poisoning both models' workspaces"""
    from duodiff_amd.engine import Context, sample_loop
    outs = []
    for poison in (False, True):
        ctx = Context.get()
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
        ms = []
        for cfg, seed in ((cfg_s, 131), (cfg_f, 132)):
            mp = ModelParams.from_dict(cfg)
            ms.append(UViT(**mp.as_dict(), precision=prec, max_batch=B).load_state_dict(synthetic_state_dict(mp, seed)).to("cuda"))
        es, ef = ms[0].engine_model(B), ms[1].engine_model(B)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        x = torch.randn(B, mp.in_chans, mp.img_size, mp.img_size, generator=torch.Generator().manual_seed(14)).cuda()
        y = torch.randint(0, mp.num_classes, (B,), generator=torch.Generator().manual_seed(15)).cuda() if mp.num_classes > 0 else None
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            if poison:
                for e in (es, ef):
                    ctx.check(ctx.lib.dd_dev_poison_workspaces(ctx.handle, e.handle, stream.cuda_stream))
            sample_loop(ctx, es, ef, x, t_switch=tsw, t_start=999, t_end=1000 - steps, y=y, seed=19, noise="philox", use_graph=use_graph, stream=stream)
            stream.synchronize()
        outs.append(x.cpu())
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
        del es, ef, ms
    bad = (outs[0] != outs[1]).flatten(1).any(1).nonzero().flatten().tolist()
    print(f"loop {name:40s} B={B:3d} {prec} flags={flags:5d} steps={steps} tsw={tsw} graph={int(use_graph)}: {'ok' if not bad else 'DIFFERS: images ' + str(bad[:8])}"
          f"  (NaN fraction {float(torch.isnan(outs[1]).float().mean()):.3f})", flush=True)


def main():
    for prec in ("bf16", "fp32"):
        one("tiny depth 1", dict(TINY, depth=1), 3, prec)
        one("tiny depth 1, max_batch 6 run at 3", dict(TINY, depth=1), 3, prec)
    t1, t3 = dict(TINY, depth=1), dict(TINY, depth=3)
    loop("tiny pair single chain", t1, t3, 6, "bf16", L.DD_DEV_NO_CHAINS)
    loop("tiny pair single chain, eager", t1, t3, 6, "bf16", L.DD_DEV_NO_CHAINS, use_graph=False)
    loop("tiny pair single chain, no switch", t3, t3, 6, "bf16", L.DD_DEV_NO_CHAINS, tsw=0)
    loop("tiny pair single chain, 1 step", t3, t3, 6, "bf16", L.DD_DEV_NO_CHAINS, steps=1, tsw=0)
    loop("tiny pair single chain, 2 steps", t3, t3, 6, "bf16", L.DD_DEV_NO_CHAINS, steps=2, tsw=0)
    loop("tiny pair forced chains", t1, t3, 6, "bf16", L.DD_DEV_FORCE_CHAINS)
    loop("tiny pair forced chains, no switch", t3, t3, 6, "bf16", L.DD_DEV_FORCE_CHAINS, tsw=0)
    loop("tiny pair fp32 single", t1, t3, 6, "fp32", L.DD_DEV_NO_CHAINS)
    loop("celeba pair default", load_config(REPO / "configs" / "uvit_celeba_3.yaml"), load_config(REPO / "configs" / "uvit_celeba.yaml"), 64, "bf16", 0, steps=4, tsw=2)
    return

    w = lambda D, nc, depth=3: dict(img_size=32, patch_size=2, in_chans=3, embed_dim=D, depth=depth, num_heads=D // 64, mlp_ratio=4, qkv_bias=False,
                                    mlp_time_embed=False, num_classes=nc, normalize_timesteps=True)
    for prec in ("bf16", "fp32"):
        one("tiny", TINY, 3, prec)
        one("tiny cond", dict(TINY, num_classes=10), 3, prec)
        one("tiny D=128 (QKV in tail)", dict(TINY, embed_dim=128, num_heads=2), 3, prec)
        one("tiny D=256", dict(TINY, embed_dim=256, num_heads=4), 3, prec)
    one("width 512 cond", w(512, 10), 4, "bf16")
    one("width 512 uncond", w(512, -1), 4, "bf16")
    one("width 768 cond", w(768, 10), 4, "bf16")
    one("width 1024 uncond", w(1024, -1), 4, "bf16")
    one("width 512 cond, depth 5", w(512, 10, 5), 3, "bf16")
    for f in (L.DD_DEV_NO_FUSED_QA, L.DD_DEV_NO_FUSED_MLP, L.DD_DEV_NO_FUSED_SKIP, L.DD_DEV_NO_FUSED_PROJ, L.DD_DEV_NO_EMBED_LN, L.DD_DEV_NO_FUSED_HEAD):
        one("width 512 cond", w(512, 10), 4, "bf16", f)
    for f in (L.DD_DEV_NO_ROWLIN, L.DD_DEV_NO_ROWLIN_PROJ, L.DD_DEV_NO_ROWLIN_SKIP, L.DD_DEV_NO_FUSED_QA):
        one("width 768 cond", w(768, 10), 4, "bf16", f)
    for f in (L.DD_DEV_NO_SPLITK, L.DD_DEV_NO_FUSED_QA):
        one("width 1024 uncond", w(1024, -1), 4, "bf16", f)
    one("celeba full", load_config(REPO / "configs" / "uvit_celeba.yaml"), 8, "bf16")
    one("imagenet64 shallow", load_config(REPO / "configs" / "uvit_imagenet64_3.yaml"), 8, "bf16")
    one("imagenet256 shallow", load_config(REPO / "configs" / "uvit_imagenet256_3.yaml"), 8, "bf16")


if __name__ == "__main__":
    main()
