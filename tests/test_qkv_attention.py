"""GPU unit test of the attention launch that computes attn.qkv itself (csrc/attention.hip qkv_attention_kernel) through its
development entry point dd_dev_qkv_attention (include/duodiff_dev.h), against a float64 reference built from the SAME
bf16-rounded operands: qkv = Linear(h), q, k, v rounded to bf16 (as the stored qkv tensor of the plain path is),
softmax(q k^T / 8) v per (image, head).  Replaces reference models/uvit.py:152-164 for D = 512 / 768 / 1024, L = 256 + 1 or 2.
Covers: both extra-token counts, with and without qkv bias, batch sizes on both workgroup -> (image, head) maps
(B % 8 == 0: XCD-grouped heads), the extra tokens as keys AND as queries.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _reference(h, w, bias, B, L, H):
    D = 64 * H
    qkv = _bf16(h).astype(np.float64) @ _bf16(w).astype(np.float64).T
    if bias is not None:
        qkv = qkv + bias.astype(np.float64)
    qkv = _bf16(qkv.astype(np.float32)).astype(np.float64).reshape(B, L, 3, H, 64)     # "B L (K H D) -> K B H L D"
    q, k, v = (qkv[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) * 0.125
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B * L, D)                               # "B H L D -> B L (H D)"


@pytest.mark.parametrize("B,extras,with_bias,H", [(3, 1, False, 8), (8, 2, False, 8), (5, 2, True, 8), (16, 1, True, 8),
                                                  (3, 2, False, 12), (8, 2, True, 12), (2, 2, True, 16), (8, 1, False, 16)])
def test_qkv_attention_against_float64_reference(B, extras, with_bias, H):
    """H = 8 / 12 / 16 = embed_dim 512 / 768 / 1024: the k range of the Linear is walked in 2 / 3 / 4 parts of 16 k-steps."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    D, L = 64 * H, 256 + extras
    g = np.random.default_rng(100 * B + extras)
    h = g.standard_normal((B * L, D), dtype=np.float32)
    # weights scaled so that the scores spread over a few units (a flat softmax would hide a wrong key order)
    w = (g.standard_normal((3 * D, D), dtype=np.float32) * (0.09 * (512.0 / D) ** 0.5)).astype(np.float32)
    bias = (g.standard_normal(3 * D, dtype=np.float32) * 0.3).astype(np.float32) if with_bias else None
    out = np.zeros((B * L, D), np.uint16)
    ms = C.c_float(0)
    ctx.check(ctx.lib.dd_dev_qkv_attention(ctx.handle, B, L, H, extras, h.ctypes.data, w.ctypes.data, bias.ctypes.data if with_bias else None,
                                           out.ctypes.data, 5, None, C.byref(ms)))
    got = (out.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    want = _reference(h, w, bias, B, L, H)
    err = np.abs(got - want)
    rows = np.arange(B * L) % L
    print(f"qkv_attention B={B} H={H} extras={extras} bias={with_bias}: max err {err.max():.3e} (patch rows {err[rows >= extras].max():.3e}, "
          f"extra rows {err[rows < extras].max():.3e}); |out| max {np.abs(want).max():.2f}; {ms.value * 1e3:.1f} us/launch")
    # bf16 output rounding (2^-9 relative) + bf16 P in the P V product + accumulation order
    assert err.max() <= 2e-2 * max(1.0, np.abs(want).max())
    assert np.sqrt((err ** 2).mean()) <= 3e-3
