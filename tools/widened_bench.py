#!/usr/bin/env python3
"""Throughput of the widened loops as device-resident loops, on the GPU box:

  * DDIM-50 (reference sampler.py:103-126), CelebA pair, B = 128, eta = 0: dd_sample_affine (one hipGraph replay per step)
    against the step-by-step Python loop (dd_forward + dd_affine_step, torch device noise) -> images / s;
  * the early-exit baseline (eesampler.py:40-89), deediff_celeba (13 blocks + 13 heads + probes), B = 128: dd_sample_early_exit
    per step against a plain dd_sample step of the same backbone -> the early-exit overhead per step.

    python tools/widened_bench.py [--out gpurun_out/widened_bench.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd import eesampler, sampler  # noqa: E402
from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.early_exit import EarlyExitUViT  # noqa: E402
from duodiff_amd.engine import sample_loop  # noqa: E402
from duodiff_amd.uvit import UViT  # noqa: E402
from duodiff_amd.weights import synthetic_ee_state_dict, synthetic_state_dict  # noqa: E402


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "widened_bench.json"))
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    B = a.batch
    out = {}
    mp_s = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba_3.yaml"))
    mp_f = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba.yaml"))
    ms = UViT(**mp_s.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mp_s, 1237)).to("cuda")
    mf = UViT(**mp_f.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mp_f, 1236)).to("cuda")
    kw = dict(use_ddim=True, ddim_steps=50, ddim_eta=0.0, late_model=mf, t_switch=300, return_device_tensor=True)
    run = lambda noise: sampler.get_samples(ms, B, sampler.predict_noise_postprocessing, 0, 3, 64, 64, noise=noise, **kw)
    run("device"); run("torch_device")                       # warm-up: graph capture, code objects
    t_graph, t_py = timed(lambda: run("device")), timed(lambda: run("torch_device"))
    out["ddim50_celeba_b%d" % B] = dict(images_per_sec_device_loop=B / t_graph, images_per_sec_python_loop=B / t_py,
                                        seconds_device_loop=t_graph, seconds_python_loop=t_py,
                                        note="49 updates: 34 full-model + 15 shallow-model forwards at t_switch = 300; wall time incl. x_T draw and the final image conversion")
    print(json.dumps(out), flush=True)
    del ms

    cfg = dict(load_config(REPO / "configs" / "deediff_celeba.yaml")["model_params"])
    ctype = cfg.pop("classifier_type")
    mp = ModelParams.from_dict(cfg)
    ee = EarlyExitUViT(UViT(**mp.as_dict(), max_batch=B), ctype).load_state_dict(synthetic_ee_state_dict(mp, 77, ctype)).eval().to("cuda")
    K = 40
    run_ee = lambda noise: eesampler.get_samples(ee, B, 0, 3, 64, 64, 0.1, mp.depth, noise=noise, num_steps=K)
    run_ee("device"); run_ee("torch_device")
    t_ee, t_ee_py = timed(lambda: run_ee("device")), timed(lambda: run_ee("torch_device"))
    em = mf.engine_model(B)
    x = torch.randn(B, 3, 64, 64, device="cuda")
    st = torch.cuda.Stream()

    def plain():
        with torch.cuda.stream(st):
            sample_loop(em.ctx, em, None, x, t_start=999, t_end=1000 - K, seed=0, noise="philox", stream=st)
        st.synchronize()
    plain()
    t_plain = timed(plain)
    out["early_exit_celeba_b%d" % B] = dict(ms_per_step_early_exit_device_loop=t_ee / K * 1e3, ms_per_step_early_exit_python_loop=t_ee_py / K * 1e3,
                                            ms_per_step_plain_backbone=t_plain / K * 1e3,
                                            overhead_ms_per_step=(t_ee - t_plain) / K * 1e3,
                                            note=f"{K} steps; early exit = 13 output heads (LayerNorm + decoder product (split-bf16 in the bf16 engine) + unpatchify/conv) + 13 probes (inside the heads' launches) + selection on top of the 13-block backbone")
    print(json.dumps(out), flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
