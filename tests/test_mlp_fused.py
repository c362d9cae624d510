"""GPU unit tests of the fused MLP kernel (csrc/mlp_fused.hip) through its development entry point dd_dev_mlp
(include/duodiff_dev.h): x += fc2(GELU_erf(fc1(LayerNorm(x)))) against a float64 reference built from the SAME
bf16-rounded operands, so the tolerance only has to cover the kernel's own arithmetic (bf16 rounding of the hidden
activation, the GELU polynomial's 2.4e-4, fp32 accumulation order) -- not the bf16 quantisation of the inputs.
Covers: every supported width, ragged last tiles, the extra-token rows (hidden-split path + reduce kernel), the fused
LayerNorm prologue / epilogue, the bf16 copy, and the attention projection fused in front (x += proj(ao) first, then the
MLP on the result).  Replaces reference models/uvit.py:86-92, 166, 206-207.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _layernorm(x, gb):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + 1e-5) * gb[0] + gb[1]).astype(np.float32)


def _reference(h, w1, b1, w2, b2, x):
    from scipy.special import erf
    s = _bf16(h).astype(np.float64) @ _bf16(w1).astype(np.float64).T + b1.astype(np.float64)
    g = 0.5 * s * (1.0 + erf(s / np.sqrt(2.0)))                      # nn.GELU() default: exact erf
    return x.astype(np.float64) + _bf16(g.astype(np.float32)).astype(np.float64) @ _bf16(w2).astype(np.float64).T + b2.astype(np.float64)


@pytest.mark.parametrize("M,D,extras,ln,proj", [
    (300, 512, 0, True, False), (128, 512, 0, False, False), (700, 512, 1, True, False), (257 * 3, 512, 2, True, False),
    (40, 64, 0, True, False), (900, 64, 2, True, False), (700, 128, 1, False, False), (1500, 256, 0, True, False),
    (300, 512, 0, True, True), (700, 512, 1, True, True), (257 * 3, 512, 2, True, True), (700, 128, 1, True, True),
    (1500, 256, 0, True, True)])
def test_fused_mlp_against_float64_reference(M, D, extras, ln, proj):
    from duodiff_amd.engine import Context
    ctx = Context.get()
    hidden = 4 * D
    g = np.random.default_rng(M + D)
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
    x1 = x
    if proj:                                          # x1 = x + attn.proj(ao): what the MLP then normalises and adds to
        x1 = (x.astype(np.float64) + _bf16(ao).astype(np.float64) @ _bf16(wp).astype(np.float64).T + bp.astype(np.float64)).astype(np.float32)
    if ln:
        h = _layernorm(x1, ln_in)                     # what the kernel's prologue computes itself from x
    want = _reference(h, w1, b1, w2, b2, x1)
    got, out, hout = x.copy(), np.zeros((M, D), np.uint16), np.zeros((M, D), np.uint16)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ms = C.c_float(0)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, extras, P(h), P(w1), P(b1), P(w2), P(b2), P(got), P(out),
                                 P(ln_in) if ln else None, P(ln_out) if ln else None, P(hout) if ln else None, 0,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                 P(ao) if proj else None, P(wp) if proj else None, P(bp) if proj else None, None, None, None, None, None))
    scale = float(np.abs(want - x).std())             # size of the block's contribution
    err = np.abs(got - want)
    print(f"M={M} D={D} extras={extras} ln={ln} proj={proj}: max {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} (mlp std {scale:.3f})")
    assert np.isfinite(got).all()
    assert err.max() <= 1.5e-2 * max(scale, 0.1) and np.sqrt((err ** 2).mean()) <= 2e-3 * max(scale, 0.1)
    as_f32 = lambda u: torch.from_numpy(u.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(as_f32(out), _bf16(got))    # the bf16 copy is the rounding of what was stored
    if ln:                                            # next block's norm1 of the updated rows, bf16
        assert np.abs(as_f32(hout) - _layernorm(got, ln_out)).max() <= 4e-2


@pytest.mark.parametrize("offset,sigma", [(100.0, 1.0), (50.0, 0.5), (-1000.0, 2.0)])
@pytest.mark.parametrize("proj", [False, True])
def test_fused_layernorms_on_rows_with_a_large_common_offset(offset, sigma, proj):
    """Rows whose mean is far larger than their spread (a DC component in the residual stream): the one-pass form
    E[x^2] - mean^2 loses the variance to cancellation there.  The kernel's shifted statistics (ln_stats_shifted) must keep
    both fused LayerNorms -- norm2 in the prologue (seen through the MLP output) and the next norm1 from the epilogue --
    at bf16-rounding accuracy, as the two-pass kernels of the unfused path are (torch.nn.LayerNorm, models/uvit.py:185-189)."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    M, D, extras = 385, 512, 0
    hidden = 4 * D
    g = np.random.default_rng(int(abs(offset)) + int(proj))
    w1 = (g.standard_normal((hidden, D), dtype=np.float32) * 0.05).astype(np.float32)
    b1 = (g.standard_normal(hidden, dtype=np.float32) * 0.2).astype(np.float32)
    w2 = (g.standard_normal((D, hidden), dtype=np.float32) * 0.05).astype(np.float32)
    b2 = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    x = (g.standard_normal((M, D), dtype=np.float32) * sigma + offset).astype(np.float32)
    x[::7] += np.float32(3 * offset)                   # and the offset differs from row to row
    ln_in = np.stack([1 + 0.1 * g.standard_normal(D), 0.05 * g.standard_normal(D)]).astype(np.float32)
    ln_out = np.stack([1 + 0.1 * g.standard_normal(D), 0.05 * g.standard_normal(D)]).astype(np.float32)
    ao = g.standard_normal((M, D), dtype=np.float32)
    wp = (g.standard_normal((D, D), dtype=np.float32) * 0.05).astype(np.float32)
    bp = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    x1 = x
    if proj:
        x1 = (x.astype(np.float64) + _bf16(ao).astype(np.float64) @ _bf16(wp).astype(np.float64).T + bp.astype(np.float64)).astype(np.float32)
    h = _layernorm(x1, ln_in)
    want = _reference(h, w1, b1, w2, b2, x1)
    got, out, hout = x.copy(), np.zeros((M, D), np.uint16), np.zeros((M, D), np.uint16)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ms = C.c_float(0)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, extras, P(h), P(w1), P(b1), P(w2), P(b2), P(got), P(out),
                                 P(ln_in), P(ln_out), P(hout), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                 P(ao) if proj else None, P(wp) if proj else None, P(bp) if proj else None, None, None, None, None, None))
    contrib = want - x1.astype(np.float64)             # the MLP's contribution, O(1), on top of rows of size |offset|
    scale = float(contrib.std())
    ulp = float(np.spacing(np.float32(4 * abs(offset))))   # fp32 resolution of the stored rows
    err = np.abs((got.astype(np.float64) - x1.astype(np.float64)) - contrib)
    rms = float(np.sqrt((err ** 2).mean()))
    as_f32 = lambda u: torch.from_numpy(u.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
    herr = np.abs(as_f32(hout).astype(np.float64) - _layernorm(got, ln_out).astype(np.float64))
    hrms = float(np.sqrt((herr ** 2).mean()))
    print(f"offset={offset} sigma={sigma} proj={proj}: mlp max {err.max():.2e} rms {rms:.2e} (std {scale:.3f}, ulp {ulp:.1e}); "
          f"ln_out max {herr.max():.2e} rms {hrms:.2e}")
    assert np.isfinite(got).all()
    # same bounds as the zero-offset cases (+ the fp32 resolution of the rows themselves)
    # (a row of size 4000 takes 128 MFMA accumulation steps of the MLP output in fp32: a few ulp of random walk)
    assert err.max() <= 1.5e-2 * max(scale, 0.1) + 16 * ulp and rms <= 2e-3 * max(scale, 0.1) + 4 * ulp
    # next norm1: bf16 rounding of values of size <= ~4 (2^-8 * 4 = 1.6e-2 max, ~1e-3 rms); a variance off by 1 % would show as 1e-2 rms
    assert herr.max() <= 2.5e-2 and hrms <= 2.5e-3


@pytest.mark.parametrize("M,D,extras", [(300, 512, 0), (770, 512, 1), (700, 128, 1), (1500, 256, 0), (258 * 2, 512, 2)])
def test_fused_tail_with_next_skip_linear(M, D, extras):
    """The fused launch with the NEXT block's skip_linear + norm1 behind the MLP (reference models/uvit.py:196-200, 206):
    x' = cat([y, skip]) . Wskip^T + bskip with y = x1 + mlp(norm2(x1)), x1 = x + proj(ao) -- against a float64 reference in
    which y and skip are rounded to bf16 exactly where the engine rounds them (the Linear's operands).  Patch rows run the skip
    phases inside the launch, extra-token rows the small kernel behind the reduce; both paths are covered."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    hidden = 4 * D
    g = np.random.default_rng(M + D + 17)
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
    skip = (g.standard_normal((M, D), dtype=np.float32) * 1.2).astype(np.float32)
    ws = (g.standard_normal((D, 2 * D), dtype=np.float32) * 0.04).astype(np.float32)
    bs = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    x1 = (x.astype(np.float64) + _bf16(ao).astype(np.float64) @ _bf16(wp).astype(np.float64).T + bp.astype(np.float64)).astype(np.float32)
    y = _reference(_layernorm(x1, ln_in), w1, b1, w2, b2, x1)                      # float64
    cat = np.concatenate([_bf16(y.astype(np.float32)), _bf16(skip)], axis=1).astype(np.float64)
    want = cat @ _bf16(ws).astype(np.float64).T + bs.astype(np.float64)
    got, out, hout = x.copy(), np.zeros((M, D), np.uint16), np.zeros((M, D), np.uint16)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ms = C.c_float(0)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, extras, P(x), P(w1), P(b1), P(w2), P(b2), P(got), P(out),
                                 P(ln_in), P(ln_out), P(hout), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                 P(ao), P(wp), P(bp), P(skip), P(ws), P(bs), None, None))
    err = np.abs(got - want)
    scale = float(want.std())
    rows = err.max(axis=1)
    print(f"skip M={M} D={D} extras={extras}: max {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} (x' std {scale:.3f}); worst row {int(rows.argmax())}")
    assert np.isfinite(got).all()
    # y carries the MLP's own error (<= 1.5e-2 of its spread, above) into a K = 2D Linear with weights of size 0.04, plus the
    # bf16 rounding of y (a different rounding boundary than the reference's wherever y sits within that error of one)
    assert err.max() <= 3e-2 * max(scale, 0.1) and np.sqrt((err ** 2).mean()) <= 4e-3 * max(scale, 0.1)
    as_f32 = lambda u: torch.from_numpy(u.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
    herr = np.abs(as_f32(hout).astype(np.float64) - _layernorm(got, ln_out).astype(np.float64))
    assert herr.max() <= 4e-2 and np.sqrt((herr ** 2).mean()) <= 3e-3


@pytest.mark.parametrize("M,D,extras,skip", [(256, 512, 0, False), (300, 512, 0, True), (772, 512, 1, False), (516, 512, 2, True),
                                            (704, 128, 1, False), (1536, 256, 0, True)])
def test_fused_tail_with_next_qkv(M, D, extras, skip):
    """The fused launch with the NEXT block's attn.qkv behind everything else (reference models/uvit.py:152, 206): qkv =
    norm1(x_out) . Wqkv^T in head-major order, with and without that block's skip_linear in between.  Checked against a
    float64 reference built from the engine's OWN fp32 rows (so that only norm1 -> bf16 -> the Linear is under test; the rows
    themselves are covered by the tests above).  Patch rows take the in-launch phases, extra-token rows the small kernel."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    hidden = 4 * D
    g = np.random.default_rng(M + D + 31)
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
    sk = (g.standard_normal((M, D), dtype=np.float32) * 1.2).astype(np.float32)
    ws = (g.standard_normal((D, 2 * D), dtype=np.float32) * 0.04).astype(np.float32)
    bs = (g.standard_normal(D, dtype=np.float32) * 0.2).astype(np.float32)
    wq = (g.standard_normal((3 * D, D), dtype=np.float32) * 0.05).astype(np.float32)
    L = M if extras == 0 else 1 + extras
    Lp, H, B = (L + 7) // 8 * 8, D // 64, M // L
    got, out, hout = x.copy(), np.zeros((M, D), np.uint16), np.zeros((M, D), np.uint16)
    qkv = np.zeros(B * 3 * H * Lp * 64, np.uint16)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    ms = C.c_float(0)
    ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, M, D, hidden, extras, P(x), P(w1), P(b1), P(w2), P(b2), P(got), P(out),
                                 P(ln_in), P(ln_out), P(hout), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                 P(ao), P(wp), P(bp), P(sk) if skip else None, P(ws) if skip else None, P(bs) if skip else None, P(wq), P(qkv)))
    assert np.isfinite(got).all()
    as_f32 = lambda u: torch.from_numpy(u.view(np.int16)).view(torch.bfloat16).to(torch.float32).numpy()
    h = _bf16(_layernorm(got, ln_out)).astype(np.float64)                        # norm1 of the rows the launch stored, rounded as the operand is
    want = h @ _bf16(wq).astype(np.float64).T                                    # [M, 3D]
    q = as_f32(qkv).reshape(B, 3 * H, Lp, 64)                                    # head-major -> [M, 3D]
    have = q[:, :, :L, :].transpose(0, 2, 1, 3).reshape(M, 3 * D)
    err = np.abs(have - want)
    scale = float(want.std())
    print(f"qkv M={M} D={D} extras={extras} skip={skip}: max {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} (std {scale:.3f}); worst row {int(err.max(axis=1).argmax())}")
    # bf16 rounding of the output (2^-9 relative) + operand-rounding boundary effects of norm1 (a value within fp32 noise of a bf16 tie)
    assert err.max() <= 2.5e-2 * max(scale, 0.1) and np.sqrt((err ** 2).mean()) <= 3e-3 * max(scale, 0.1)
    assert np.all(q[:, :, L:, :] == 0)                                           # pad rows of the units stay untouched (attention relies on zeros there)


def test_fused_mlp_rows_do_not_depend_on_their_neighbours():
    """Which path a row takes (main tile / hidden-split) depends only on its token index, so the same row inside two
    different batches gives bit-identical results (the engine-level statement: test_batch_independence_at_full_size)."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    D, hidden, extras = 512, 2048, 1
    g = np.random.default_rng(7)
    w1 = (g.standard_normal((hidden, D), dtype=np.float32) * 0.05).astype(np.float32)
    b1 = np.zeros(hidden, np.float32)
    w2 = (g.standard_normal((D, hidden), dtype=np.float32) * 0.05).astype(np.float32)
    b2 = np.zeros(D, np.float32)
    ln = np.stack([np.ones(D), np.zeros(D)]).astype(np.float32)
    x_all = g.standard_normal((400, D), dtype=np.float32)             # 200 "images" of 1 extra + 1 patch token
    P = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(x):
        got, hout = x.copy(), np.zeros(x.shape, np.uint16)
        ms = C.c_float(0)
        ctx.check(ctx.lib.dd_dev_mlp(ctx.handle, x.shape[0], D, hidden, extras, P(x), P(w1), P(b1), P(w2), P(b2), P(got), None,
                                     P(ln), P(ln), P(hout), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms),
                                     None, None, None, None, None, None, None, None))
        return got, hout

    big, hbig = run(x_all)
    small, hsmall = run(x_all[:6].copy())
    assert np.array_equal(big[:6], small) and np.array_equal(hbig[:6], hsmall)
