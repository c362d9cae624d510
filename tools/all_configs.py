"""Measure the three DuoDiff workloads of BASELINE.json at their stated per-GPU batch sizes (short runs, scaled)."""
import json, sys, time, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict
from duodiff_amd.uvit import UViT
from duodiff_amd.engine import sample_loop
from duodiff_amd import sampler
R = "/root/repo/configs/"
work = [("CelebA-64", "uvit_celeba_3", "uvit_celeba", 128), ("ImageNet-64", "uvit_imagenet64_3", "uvit_imagenet64", 256),
        ("ImageNet-256-latent", "uvit_imagenet256_3", "uvit_imagenet256", 32)]
STEPS = 100
out = {}
for name, cs, cf, B in work:
    mps, mpf = ModelParams.from_dict(load_config(R + cs + ".yaml")), ModelParams.from_dict(load_config(R + cf + ".yaml"))
    s = UViT(**mps.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mps, 1)).to("cuda")
    f = UViT(**mpf.as_dict(), max_batch=B).load_state_dict(synthetic_state_dict(mpf, 2)).to("cuda")
    es, ef = s.engine_model(B), f.engine_model(B)
    sampler.seed_everything(0)
    x = torch.randn(B, mpf.in_chans, mpf.img_size, mpf.img_size).cuda()
    y = None
    if mpf.num_classes > 0:
        y = torch.randint(1, 1001, (B,)).clamp(max=mpf.num_classes - 1).cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for rep in range(2):
            xx = x.clone() if rep == 0 else x
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sample_loop(es.ctx, es, ef, xx, t_switch=30, t_start=999, t_end=1000 - STEPS, y=y, seed=0, noise="philox", stream=st)
            st.synchronize(); dt = time.perf_counter() - t0
    tim = es.ctx.last_sample_timing()
    ms_s, ms_f = tim[1] / 30, tim[2] / 70
    per1000 = (300 * ms_s + 700 * ms_f) / 1e3
    flop_img = (0.3 * mps.flops_per_image() + 0.7 * mpf.flops_per_image()) * 1000
    out[name] = dict(batch=B, ms_shallow_step=ms_s, ms_full_step=ms_f, images_per_sec=B / per1000,
                     tflops=B / per1000 * flop_img / 1e12, finite=bool(torch.isfinite(x).all()))
    print(name, json.dumps(out[name]), flush=True)
    del s, f, es, ef
json.dump(out, open("/root/repo/gpurun_out/all_configs.json", "w"), indent=1)
