#!/usr/bin/env python3
"""Phase timeline of qkv_attention_kernel from an instrumented variant build.  The product source holds no instrumentation:
    git apply tools/experiments/qkv_attention_timing.patch && python tools/build_variant.py qat -DDD_QA_TIMING && git apply -R tools/experiments/qkv_attention_timing.patch
    DUODIFF_LIB=duodiff_amd/libduodiff_qat.so python tools/qa_timing.py [B]
Every workgroup leaves its 100 MHz timestamps in the first output row of its (image, head); prints the mean phase durations."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from duodiff_amd.engine import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H, D, E = 8, 512, 1
    L = 256 + E
    g = np.random.default_rng(0)
    h = g.standard_normal((B * L, D), dtype=np.float32)
    w = (g.standard_normal((3 * D, D), dtype=np.float32) * 0.09).astype(np.float32)
    out = np.zeros((B * L, D), np.uint16)
    ms = C.c_float(0)
    ctx = Context.get()
    ctx.check(ctx.lib.dd_dev_qkv_attention(ctx.handle, B, L, H, E, h.ctypes.data, w.ctypes.data, None, out.ctypes.data, 20, None, C.byref(ms)))
    print(f"B={B}: {ms.value * 1e3:.1f} us/launch")
    rows = out.reshape(B, L, D)[:, E, :].reshape(B, H, 64)                  # first patch row of every (image, head)
    ts = np.ascontiguousarray(rows).view(np.uint64).reshape(B * H, 16)[:, :11].astype(np.int64)
    if ts[:, 10].min() == 0:
        print("no timestamps: not a DD_QA_TIMING build")
        return
    t0 = ts[:, 0].min()
    names = ["start->dma0 issued (xf loads issued)", "->tile0 ready (xf, dma0 landed, barrier)", "tile0", "tile1", "tile2", "tile3", "tile4",
             "tile5 + final barrier", "phase B main chunk", "split chunk + merge"]
    d = np.diff(ts, axis=1) * 10.0 / 1e3                                     # us
    for i, n in enumerate(names):
        print(f"  {n:44s} mean {d[:, i].mean():7.2f} us  (min {d[:, i].min():6.2f}, max {d[:, i].max():6.2f})")
    full = np.ascontiguousarray(rows).view(np.uint64).reshape(B * H, 16).astype(np.int64)
    if full[:, 12].min() > 0:     # tile 2 in detail: barrier exit (stamp 4) -> DMA issued (11) -> MFMAs done (12) -> epilogue done (13) -> next barrier exit (5)
        seq = np.stack([full[:, 4], full[:, 11], full[:, 12], full[:, 5], full[:, 5]], 1)
        dd = np.diff(seq, axis=1) * 10.0 / 1e3
        for i, n in enumerate(["tile2: whole tile until its wait", "tile3 head: vmcnt(0) wait (DMA of tile 3 landed)", "tile3 head: barrier", "-"]):
            print(f"  {n:44s} mean {dd[:, i].mean():7.2f} us  (min {dd[:, i].min():6.2f}, max {dd[:, i].max():6.2f})")
    tot = (ts[:, 10] - ts[:, 0]) * 10.0 / 1e3
    print(f"  workgroup total mean {tot.mean():.2f} us; first start -> last end {(ts[:, 10].max() - t0) * 10.0 / 1e3:.2f} us; "
          f"start spread {(ts[:, 0].max() - t0) * 10.0 / 1e3:.2f} us")


if __name__ == "__main__":
    main()
