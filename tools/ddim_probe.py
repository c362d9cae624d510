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
