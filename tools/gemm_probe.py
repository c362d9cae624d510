"""Dev tool: fixed-cost probe of the GEMM kernels."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for v in (0, 8):
        for M in (128, 1024, 4096, 16384, 32896):
            for ab in (7, 0):
                ms, tf, _ = ctx.dev_gemm(M, 1536, 512, variant=v | (ab << 8), epilogue=0, iters=50, check=False, stream=stream)
                print(f"variant={v} M={M:6d} ablate={ab}: {ms*1e3:8.2f} us", flush=True)
