"""Experiment: two half-batch chains on disjoint CU halves (hipExtStreamCreateWithCUMask) vs one full-batch chain."""
import ctypes as C, sys, time, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict
from duodiff_amd.engine import Context, Model, sample_loop
hip = C.CDLL("libamdhip64.so")
mp = ModelParams.from_dict(load_config("/root/repo/configs/uvit_celeba.yaml"))
sd = synthetic_state_dict(mp, 1)
def build(ctx, B):
    m = Model(ctx, mp, B)
    for k, v in sd.items(): m.set_param(k, v)
    m.finalize("bf16"); return m
def masked_stream(words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
STEPS = 40
def run(label, chains, ncus, use_graph):
    ctxs = [Context(0) for _ in chains]
    ctxs[0].check(ctxs[0].lib.dd_set_num_cus(ctxs[0].handle, ncus))
    ms = [build(c, B) for c, (B, _) in zip(ctxs, chains)]
    xs = [torch.randn(B, 3, 64, 64).cuda() for B, _ in chains]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c, m, x, (B, s) in zip(ctxs, ms, xs, chains):
            with torch.cuda.stream(s):
                sample_loop(c, m, None, x, t_start=999, t_end=1000 - STEPS, seed=1, noise="philox", use_graph=use_graph, stream=s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot = sum(B for B, _ in chains)
    print(f"{label}: {dt/STEPS*1e3:.3f} ms per step of {tot} images -> {tot/(dt/STEPS)/1000:.2f} img-steps/ms", flush=True)
    ctxs[0].check(ctxs[0].lib.dd_set_num_cus(ctxs[0].handle, 256))
full = [0xffffffff] * 8
lo = [0xffffffff] * 4 + [0] * 4
hi = [0] * 4 + [0xffffffff] * 4
even = [0x55555555] * 8
odd = [0xaaaaaaaa] * 8
for g in (False, True):
    run(f"graph={g} 1 chain B=128 full mask", [(128, masked_stream(full))], 256, g)
    run(f"graph={g} 2 chains B=64 lo/hi halves", [(64, masked_stream(lo)), (64, masked_stream(hi))], 128, g)
    run(f"graph={g} 2 chains B=64 even/odd CUs", [(64, masked_stream(even)), (64, masked_stream(odd))], 128, g)
