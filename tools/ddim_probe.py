#!/usr/bin/env python3
"""Where DDIM-50's time goes: dd_sample_affine (device-resident loop, hipGraph replays) with two chains / one chain, wall and GPU time of the
loop, against the step-by-step Python loop.   python tools/ddim_probe.py"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd import _lib as L, sampler  # noqa: E402
from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.uvit import UViT  # noqa: E402
from duodiff_amd.weights import synthetic_state_dict  # noqa: E402

B = 128
mp_s = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba_3.yaml"))
mp_f = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba.yaml"))
ms = UViT(**mp_s.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mp_s, 1237)).to("cuda")
mf = UViT(**mp_f.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mp_f, 1236)).to("cuda")
ctx = ms.engine_model(B).ctx
kw = dict(use_ddim=True, ddim_steps=50, ddim_eta=0.0, late_model=mf, t_switch=300, return_device_tensor=True)
run = lambda noise: sampler.get_samples(ms, B, sampler.predict_noise_postprocessing, 0, 3, 64, 64, noise=noise, **kw)
for name, flags, noise in (("device loop, two chains", 0, "device"), ("device loop, one chain", L.DD_DEV_NO_CHAINS, "device"), ("python loop", 0, "torch_device")):
    ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
    run(noise)
    best, gpu = 1e9, None
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(noise)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt < best:
            best, gpu = dt, (ctx.last_sample_timing() if noise == "device" else None)
    print(f"{name:28s}: wall {best * 1e3:7.2f} ms = {B / best:6.1f} images/s" + (f"; GPU time of the loop {gpu[0]:.2f} ms (first backbone {gpu[1]:.2f}, late {gpu[2]:.2f}), chains {ctx.lib.dd_dev_last_sample_chains(ctx.handle)}" if gpu else ""), flush=True)
ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))

# ---- the early-exit loop the same way (deediff_celeba: 13 blocks + 13 heads + probes), 40 steps
from duodiff_amd import eesampler  # noqa: E402
from duodiff_amd.early_exit import EarlyExitUViT  # noqa: E402
from duodiff_amd.weights import synthetic_ee_state_dict  # noqa: E402

del ms
cfg = dict(load_config(REPO / "configs" / "deediff_celeba.yaml")["model_params"])
ctype = cfg.pop("classifier_type")
mp = ModelParams.from_dict(cfg)
ee = EarlyExitUViT(UViT(**mp.as_dict(), max_batch=B), ctype).load_state_dict(synthetic_ee_state_dict(mp, 77, ctype)).eval().to("cuda")
K = 40
for name, flags in (("early exit, two chains", 0), ("early exit, one chain", L.DD_DEV_NO_CHAINS)):
    ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
    f = lambda: eesampler.get_samples(ee, B, 0, 3, 64, 64, 0.1, mp.depth, noise="device", num_steps=K)
    f()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gpu = ctx.last_sample_timing()
    print(f"{name:28s}: wall {best / K * 1e3:6.3f} ms per step; GPU time of the loop {gpu[0] / K:.3f} ms per step, chains {ctx.lib.dd_dev_last_sample_chains(ctx.handle)}", flush=True)
ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
