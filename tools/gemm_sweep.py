"""Dev tool: sweep GEMM variants over the CelebA shapes on the GPU (not part of the product)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
shapes = [("qkv", 1536, 512, 0), ("proj", 512, 512, 2), ("fc1", 2048, 512, 1), ("fc2", 512, 2048, 2)]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(8))
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in shapes:
        for v in variants:
            try:
                ms, tf, mm = ctx.dev_gemm(M, N, K, variant=v, epilogue=epi, iters=30, check=True, stream=stream)
                print(f"{name:5s} N={N:5d} K={K:5d} epi={epi} variant={v}: {ms*1e3:8.1f} us {tf:7.1f} TF mismatches={mm}", flush=True)
            except Exception as e:
                print(f"{name} variant {v}: FAILED {e}", flush=True)
