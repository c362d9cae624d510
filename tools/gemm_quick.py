"""Dev tool: the four block GEMMs, chosen variant (default 8), full + no-epilogue, with a bitwise check against variant 0."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
v = int(sys.argv[1]) if len(sys.argv) > 1 else 8
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in (("fc1", 2048, 512, 1), ("fc2", 512, 2048, 2), ("qkv", 1536, 512, 0), ("proj", 512, 512, 2)):
        ms, tf, mm = ctx.dev_gemm(M, N, K, variant=v, epilogue=epi, iters=30, check=True, stream=stream)
        ms1, _, _ = ctx.dev_gemm(M, N, K, variant=v | (1 << 8), epilogue=epi, iters=30, check=False, stream=stream)
        ms3, _, _ = ctx.dev_gemm(M, N, K, variant=v | (3 << 8), epilogue=epi, iters=30, check=False, stream=stream)
        print(f"{name:5s} v{v} full {ms*1e3:7.1f} us {tf:7.1f} TF mismatches {mm} | no-epi {ms1*1e3:6.1f} | mfma+ds {ms3*1e3:6.1f}", flush=True)
