import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
shapes = [("qkv", 1536, 512, 0), ("proj", 512, 512, 2), ("fc1", 2048, 512, 1), ("fc2", 512, 2048, 2), ("skip", 512, 1024, 3)]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in shapes:
        r = {}
        for ab in (0, 1, 3, 5, 7):
            ms, tf, _ = ctx.dev_gemm(M, N, K, variant=8 | (ab << 8), epilogue=epi, iters=20, check=False, stream=stream)
            r[ab] = ms * 1e3
        print(f"{name:5s} full {r[0]:7.1f} | no-epilogue {r[1]:7.1f} | mfma+ds only {r[3]:7.1f} | dma only {r[5]:7.1f} | skeleton {r[7]:6.1f}  -> epilogue costs {r[0]-r[1]:6.1f} us", flush=True)
