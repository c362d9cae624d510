import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ms, tf, _ = ctx.dev_gemm(128 * 257, 2048, 512, variant=14, epilogue=1, iters=3, check=False, stream=stream)
    print(f"v14 (stamped fc1): {ms*1e3:.1f} us")
