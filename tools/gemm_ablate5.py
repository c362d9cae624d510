import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in (("fc1", 2048, 512, 1), ("qkv", 1536, 512, 0)):
        for ab, what in ((0, "full"), (8, "no gelu math"), (16, "no out stores"), (24, "no gelu, no out stores"), (1, "no epilogue")):
            ms, tf, _ = ctx.dev_gemm(M, N, K, variant=8 | (ab << 8), epilogue=epi, iters=30, check=False, stream=stream)
            print(f"{name} {what:26s} {ms*1e3:7.1f} us", flush=True)
