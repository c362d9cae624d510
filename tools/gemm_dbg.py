import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.engine import Context
ctx = Context.get()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for (M, N, K, epi) in ((1024, 512, 512, 0), (32896, 512, 512, 0), (32896, 1536, 512, 0), (4096, 128, 64, 0), (4096, 128, 128, 0)):
        ms, tf, mm = ctx.dev_gemm(M, N, K, variant=8, epilogue=epi, iters=2, check=True, stream=st)
        print(M, N, K, epi, "mismatches", mm, flush=True)
