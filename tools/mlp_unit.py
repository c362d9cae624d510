#!/usr/bin/env python3
"""Unit check of the fused MLP kernel against a float64 reference built from the same bf16-rounded operands.

    python tools/mlp_unit.py [--M 300] [--D 512] [--hidden 2048] [--extras 0] [--iters 0]
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


def bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()


def layernorm(x, gb):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + 1e-5) * gb[0] + gb[1]).astype(np.float32)


def reference(h, w1, b1, w2, b2, x):
    from scipy.special import erf
    hb, w1b, w2b = bf16_round(h).astype(np.float64), bf16_round(w1).astype(np.float64), bf16_round(w2).astype(np.float64)
    s = hb @ w1b.T + b1.astype(np.float64)
    g = 0.5 * s * (1.0 + erf(s / np.sqrt(2.0)))
    p = bf16_round(g.astype(np.float32)).astype(np.float64)
    return x.astype(np.float64) + p @ w2b.T + b2.astype(np.float64), s, p


def run(ctx, M, D, hidden, seed=0, iters=0, extras=0, ln=False, proj=False):
    g = np.random.default_rng(seed)
    h = g.standard_normal((M, D), dtype=np.float32)
    w1 = (g.standard_normal((hidden, D), dtype=np.float32) * 0.05).astype(np.float32)
    b1 = (g.standard_normal(hidden, dtype=np.float32) * 0.2).astype(np.float32)
    w2 = (g.standard_normal((D, hidden), dtype=np.float32) * 0.05).astype(np.float32)
    b2 = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    x = (g.standard_normal((M, D), dtype=np.float32) * 1.5 + 0.3).astype(np.float32)
    ln_in = np.stack([1 + 0.1 * g.standard_normal(D), 0.05 * g.standard_normal(D)]).astype(np.float32)
    ln_out = np.stack([1 + 0.1 * g.standard_normal(D), 0.05 * g.standard_normal(D)]).astype(np.float32)
    ao = g.standard_normal((M, D), dtype=np.float32)
    wp = (g.standard_normal((D, D), dtype=np.float32) * 0.05).astype(np.float32)
    bp = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    x0 = x
    if proj:
        x = (x.astype(np.float64) + bf16_round(ao).astype(np.float64) @ bf16_round(wp).astype(np.float64).T + bp).astype(np.float32)
    if ln:
        h = layernorm(x, ln_in)                      # the kernel's prologue computes this itself
    want, s, p = reference(h, w1, b1, w2, b2, x)
    x = x0
    got = x.copy()
    out = np.zeros((M, D), np.uint16)
    hout = np.zeros((M, D), np.uint16)
    ms = C.c_float(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, extras, P(h), P(w1), P(b1), P(w2), P(b2), P(got), P(out),
                                 P(ln_in) if ln else None, P(ln_out) if ln else None, P(hout) if ln else None, iters,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                 P(ao) if proj else None, P(wp) if proj else None, P(bp) if proj else None, None, None, None, None, None))
    if ln:
        hb = torch.from_numpy(hout.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
        print(f"   ln_out: max|h' - LayerNorm(x')| = {np.abs(hb - layernorm(want, ln_out)).max():.3e}")
    err = np.abs(got - want)
    scale = np.abs(want - x).std()
    ob = torch.from_numpy(out.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
    cp = np.abs(ob - bf16_round(got)).max()
    rows = err.max(axis=1)
    print(f"M={M} D={D} hidden={hidden} extras={extras} ln={int(ln)} proj={int(proj)}: max|err|={err.max():.3e} rms={np.sqrt((err ** 2).mean()):.3e} (mlp std {scale:.3f}) "
          f"bf16-copy mismatch={cp:.1e} worst row {int(rows.argmax())} worst col {int(err.max(axis=0).argmax())}"
          + (f"  {ms.value * 1e3:.1f} us/launch" if iters else ""), flush=True)
    return err.max(), scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, nargs="+", default=[300])
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--hidden", type=int, default=0)
    ap.add_argument("--extras", type=int, default=0)
    ap.add_argument("--ln", action="store_true", help="fused LayerNorm prologue + epilogue")
    ap.add_argument("--proj", action="store_true", help="attention projection fused in front (implies --ln)")
    ap.add_argument("--iters", type=int, default=0)
    a = ap.parse_args()
    ctx = Context.get()
    for M in a.M:
        run(ctx, M, a.D, a.hidden or 4 * a.D, iters=a.iters, extras=a.extras, ln=a.ln or a.proj, proj=a.proj)


if __name__ == "__main__":
    main()
