"""Dev experiment: would MORE than two chains pay on the CelebA headline?  Two contexts (each with its own pair of models, state and side stream) run
dd_sample on 64 images each from two host threads = four chains of 32, against one context on 128 images = two chains of 64.
    python tools/four_chains.py [steps]"""
import sys, time, threading, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.engine import Context, Model, sample_loop
from duodiff_amd.weights import synthetic_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
def build(ctx, name, seed, B):
    mp = ModelParams.from_dict(load_config(f"/root/repo/configs/{name}.yaml"))
    m = Model(ctx, mp, B)
    for k, v in synthetic_state_dict(mp, seed).items():
        m.set_param(k, v)
    m.finalize("bf16")
    return m
def run(parts):
    """parts: list of (ctx, first, late, x, stream); every part runs its own loop from its own host thread."""
    def work(p):
        ctx, first, late, x, st = p
        with torch.cuda.stream(st):
            sample_loop(ctx, first, late, x, t_switch=int(0.3 * steps), t_start=999, t_end=1000 - steps, seed=3, noise="philox", use_graph=True, stream=st)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(p,)) for p in parts]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt
c1 = Context.get("cuda:0")
f1, l1 = build(c1, "uvit_celeba_3", 1, 128), build(c1, "uvit_celeba", 2, 128)
x = torch.randn(128, 3, 64, 64).cuda()
s1 = torch.cuda.Stream()
dt2 = run([(c1, f1, l1, x.clone(), s1)])
print(f"one context, B = 128 (two chains of 64): {dt2 * 1e3 / steps:.3f} ms per step")
c2 = Context(0)
f2, l2 = build(c2, "uvit_celeba_3", 1, 64), build(c2, "uvit_celeba", 2, 64)
s2 = torch.cuda.Stream()
dt4 = run([(c1, f1, l1, x[:64].clone(), s1), (c2, f2, l2, x[64:].clone(), s2)])
print(f"two contexts, B = 64 each (four chains of 32): {dt4 * 1e3 / steps:.3f} ms per step")
