"""Dev tool: where does the time of the four block GEMMs go (gemm256, variant 8)? Ablation bits: 1 skip epilogue,
2 skip DMA after the first k-tile, 4 skip MFMA, 8 skip GELU math, 16 skip out stores."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
M = 128 * 257
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for name, N, K, epi in (("fc2", 512, 2048, 2), ("fc1", 2048, 512, 1), ("qkv", 1536, 512, 0), ("proj", 512, 512, 2)):
        for ab, what in ((0, "full"), (1, "no epilogue"), (3, "no epi, no dma (mfma+ds)"), (5, "no epi, no mfma (dma only)"),
                         (6, "epilogue only"), (7, "nothing (skeleton)")):
            ms, tf, _ = ctx.dev_gemm(M, N, K, variant=8 | (ab << 8), epilogue=epi, iters=20, check=False, stream=stream)
            print(f"{name:5s} {what:28s} {ms*1e3:7.1f} us", flush=True)
