#!/usr/bin/env python3
"""Is the fused block tail bound by cycles or by the clock the chip holds under it?

    DUODIFF_LIB=duodiff_amd/libduodiff_stamps.so python tools/power_probe.py [--iters 2000]

Runs mlp_fused_kernel<512, LN, PROJ> stand-alone (dd_dev_mlp) on 256 / 128 / 64 main tiles, on random and on all-zero
operands, `iters` back-to-back launches each (long enough for the clock to settle), and prints the time per launch; with
the stamped variant build (build/var_stamps: s_memtime / s_memrealtime at the phase boundaries, wave 0 of every
workgroup) also the cycles of each phase and the in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
(MI355X_MICROARCH.md, DVFS give-back (6)).  Stamps go to a buffer of their own; no output depends on them.
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


def run(ctx, M, D, kind, iters, seed=0):
    hidden = 4 * D
    g = np.random.default_rng(seed)
    z = kind == "zero"
    rnd = lambda shape, s=1.0: (np.zeros(shape, np.float32) if z else (g.standard_normal(shape, dtype=np.float32) * s).astype(np.float32))
    h = rnd((M, D))
    w1, b1, w2, b2 = rnd((hidden, D), 0.05), rnd(hidden, 0.2), rnd((D, hidden), 0.05), rnd(D, 0.2)
    x = rnd((M, D), 1.5)
    ln_in = np.stack([np.ones(D), np.zeros(D)]).astype(np.float32)
    ln_out = ln_in.copy()
    if z:
        ln_in[:] = 0
        ln_out[:] = 0
    ao, wp, bp = rnd((M, D)), rnd((D, D), 0.05), rnd(D, 0.2)
    out = np.zeros((M, D), np.uint16)
    hout = np.zeros((M, D), np.uint16)
    ms = C.c_float(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, 0, P(h), P(w1), P(b1), P(w2), P(b2), P(x), P(out), P(ln_in), P(ln_out), P(hout), iters,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms), P(ao), P(wp), P(bp), None, None, None, None, None))
    line = f"tiles={M // 128:4d} {kind:6s}: {ms.value * 1e3:7.1f} us/launch"
    rd = getattr(ctx.lib, "dd_dev_read_stamps", None)
    if rd is not None:
        n = M // 128
        buf = np.zeros((2048, 8), np.uint64)
        rd.argtypes = [C.c_void_p, C.c_int]
        rd(P(buf), buf.size)
        s = buf[:n].astype(np.int64)
        d = lambda a, b: np.median(s[:, b] - s[:, a])
        cyc = s[:, 4] - s[:, 0]
        rt = (s[:, 6] - s[:, 5]).astype(np.float64)
        clk = np.median(cyc / np.maximum(rt, 1)) * 100.0     # MHz
        line += (f" | cycles: prologue+proj+LN {d(0, 1):7.0f}  chunk loop {d(1, 2):7.0f} ({d(1, 2) / 64:6.0f}/chunk, MFMA pipe needs 2048)  tail {d(2, 3):6.0f}"
                 f"  epilogue {d(3, 4):6.0f}  total {np.median(cyc):7.0f} | in-kernel clock {clk:5.0f} MHz"
                 f" | workgroup lifetime {np.median(rt) / 100:6.1f} us, spread of starts {(s[:, 5].max() - s[:, 5].min()) / 100:5.1f} us")
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--tiles", type=int, nargs="+", default=[256, 128, 64, 256])
    a = ap.parse_args()
    ctx = Context.get()
    for kind in ("random", "zero", "random"):
        for t in a.tiles:
            run(ctx, t * 128, a.D, kind, a.iters)


if __name__ == "__main__":
    main()
