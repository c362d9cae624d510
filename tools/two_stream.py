"""Experiment: one B=128 chain vs two concurrent B=64 chains on two streams (full CelebA model, 50 steps)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict
from duodiff_amd.engine import Context, Model, sample_loop
mp = ModelParams.from_dict(load_config("/root/repo/configs/uvit_celeba.yaml"))
sd = synthetic_state_dict(mp, 1)
def build(ctx, B):
    m = Model(ctx, mp, B)
    for k, v in sd.items(): m.set_param(k, v)
    m.finalize("bf16"); return m
STEPS = 50
def run(nchains, B):
    ctxs = [Context(0) for _ in range(nchains)]
    ms = [build(c, B) for c in ctxs]
    xs = [torch.randn(B, 3, 64, 64).cuda() for _ in range(nchains)]
    sts = [torch.cuda.Stream() for _ in range(nchains)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c, m, x, s in zip(ctxs, ms, xs, sts):
            with torch.cuda.stream(s):
                sample_loop(c, m, None, x, t_start=999, t_end=1000 - STEPS, seed=1, noise="philox", use_graph=True, stream=s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nchains} chain(s) x B={B}: {dt/STEPS*1e3:.3f} ms per step-of-{nchains*B}-images -> {nchains*B/(dt/STEPS)/1000:.2f} img-steps/ms", flush=True)
run(1, 128)
run(2, 64)
run(4, 32)
run(2, 128)
