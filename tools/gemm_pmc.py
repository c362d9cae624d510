"""Dev tool for rocprofv3 --pmc: a few launches of the fc1 GEMM, variant argv[1], ablation argv[2]."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
v = int(sys.argv[1]); ab = int(sys.argv[2]) if len(sys.argv) > 2 else 0
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ms, tf, _ = ctx.dev_gemm(128 * 257, 2048, 512, variant=v | (ab << 8), epilogue=1, iters=4, check=False, stream=stream)
    print(f"v{v} ablate {ab}: {ms*1e3:.1f} us")
