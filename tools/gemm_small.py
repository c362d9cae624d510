"""Dev tool: GEMM shapes of the ImageNet-256-latent workload (B=32, M=8256, D=1024): gemm256 fills only 128-384 tiles."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
stream = torch.cuda.Stream()
M = 32 * 258
with torch.cuda.stream(stream):
    for name, N, K, epi in (("qkv", 3072, 1024, 0), ("proj", 1024, 1024, 2), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 2), ("skip", 1024, 2048, 3)):
        for v in (8, 0, 1, 3, 5, 6):
            ms, tf, mm = ctx.dev_gemm(M, N, K, variant=v, epilogue=epi, iters=20, check=(v != 0), stream=stream)
            print(f"{name:5s} N={N:5d} K={K:5d} variant {v}: {ms*1e3:7.1f} us {tf:7.1f} TF mism {mm}", flush=True)
