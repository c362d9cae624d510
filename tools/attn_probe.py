#!/usr/bin/env python3
"""Phase timeline and in-kernel clock of qkv_attention_kernel<512> (stamped variant build, tools/experiments/attn_stamps_variant.py):

    DUODIFF_LIB=duodiff_amd/libduodiff_astamps.so python tools/attn_probe.py [--B 128] [--iters 2000]

stand-alone launches (dd_dev_qkv_attention, random operands); per wave: cycles of prologue, phase A (12 weight blocks), the K / V image epilogue,
phase B (32 queries x 9 key tiles), the extras' split chunk; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd.engine import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[128, 32])
    ap.add_argument("--iters", type=int, default=2000)
    a = ap.parse_args()
    ctx = Context.get()
    D, H, E = 512, 8, 1
    L = 256 + E
    g = np.random.default_rng(0)
    w = (g.standard_normal((3 * D, D), dtype=np.float32) * 0.05).astype(np.float32)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    for B in a.B:
        h = g.standard_normal((B * L, D), dtype=np.float32)
        out = np.zeros((B * L, D), np.uint16)
        ms = C.c_float(0)
        ctx.check(ctx.lib.dd_dev_qkv_attention(ctx.handle, B, L, H, E, P(h), P(w), None, P(out), a.iters, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms)))
        line = f"B={B:4d} ({B * H} workgroups): {ms.value * 1e3:7.1f} us/launch"
        rd = getattr(ctx.lib, "dd_dev_read_attn_stamps", None)
        if rd is not None:
            n = min(B * H, 1024) * 8
            buf = np.zeros((8192, 8), np.uint64)
            rd.argtypes = [C.c_void_p, C.c_int]
            rd(P(buf), buf.size)
            s = buf[:n].astype(np.int64)
            d = lambda i, j: np.median(s[:, j] - s[:, i])
            clk = np.median((s[:, 5] - s[:, 0]) / np.maximum(s[:, 7] - s[:, 6], 1)) * 100.0
            line += (f" | cycles per wave: prologue {d(0, 1):6.0f}  phase A {d(1, 2):6.0f} ({d(1, 2) / 12:5.0f} per block; the pipe needs 1 088)  K/V images {d(2, 3):5.0f}"
                     f"  phase B {d(3, 4):6.0f}  extras' chunk {d(4, 5):5.0f}  total {d(0, 5):6.0f} | clock {clk:5.0f} MHz")
        print(line, flush=True)


if __name__ == "__main__":
    main()
