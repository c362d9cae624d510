#!/usr/bin/env python3
"""Yardstick, not product: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the engine's plain-GEMM shapes,
bf16, random operands, on the same GPU -- the number the hand-written gemm256 kernel is compared with in DESIGN.md."""
import torch, time
shapes = [("celeba qkv", 32896, 1536, 512), ("celeba skip", 32896, 512, 1024), ("celeba fc1", 32896, 2048, 512),
          ("imagenet64 qkv", 66048, 2304, 768), ("imagenet64 fc1", 66048, 3072, 768), ("imagenet64 fc2", 66048, 768, 3072),
          ("imagenet256 fc1", 8256, 4096, 1024), ("square 4096", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:18s} M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
