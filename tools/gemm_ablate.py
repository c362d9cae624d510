"""Dev tool: ablate parts of the GEMM kernel to find the bound (not part of the product)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
shapes = [("qkv", 1536, 512, 0), ("fc2", 512, 2048, 2)]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3]
names = {0: "full", 1: "no-epilogue-traffic", 2: "no-lds-dma", 4: "no-mfma", 3: "no-epi+no-dma (mfma+ds_read only)",
         5: "no-epi+no-mfma (dma only)", 6: "no-dma+no-mfma (epilogue only)", 7: "nothing"}
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in shapes:
        for v in variants:
            for ab in (0, 1, 2, 4, 3, 5, 6, 7):
                ms, tf, _ = ctx.dev_gemm(M, N, K, variant=v | (ab << 8), epilogue=epi, iters=20, check=False, stream=stream)
                print(f"{name:5s} variant={v} ablate={ab} {names[ab]:36s}: {ms*1e3:8.1f} us", flush=True)
