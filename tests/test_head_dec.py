"""GPU unit test of the output head's first launch (csrc/rowops.hip head_dec_kernel) through its development entry point
dd_dev_head_dec (include/duodiff_dev.h), against a float64 reference: dec = decoder_pred(LayerNorm(x)) (reference
models/uvit.py:377-378), and -- early-exit heads -- the MLP probe's per-row value sigmoid(x . w + b) from the same launch
(reference models/early_exit.py:31-37).

The kernel multiplies the UN-normalised rows as they arrive and takes the LayerNorm out of the product afterwards
(dec = rstd (Wg . d - mean_d wsum) + c with d = x - x[0]); the cases below are the ones that formulation could get wrong:
rows that carry a large common offset (the shift by the row's own first element must absorb it), rows of tiny spread,
every supported width (hand-counted row loads at D <= 512, compiler-counted above), patch-rows-only launches with one and
two extra tokens, every-row launches whose last unit is ragged, more units than waves, the full batch (one unit per wave of the chip).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference(x, g, b, w, bias):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    xn = (x - mu) / np.sqrt(var + 1e-5) * g.astype(np.float64) + b.astype(np.float64)
    return xn @ w.astype(np.float64).T + bias.astype(np.float64)


def _run(M, D, pd, tok_l, tok_e, x, probe, split=0):
    from duodiff_amd.engine import Context
    ctx = Context.get()
    r = np.random.default_rng(D + pd)
    g = (1.0 + 0.2 * r.standard_normal(D)).astype(np.float32)
    b = (0.1 * r.standard_normal(D)).astype(np.float32)
    w = (r.standard_normal((pd, D)) / np.sqrt(D)).astype(np.float32)
    bias = (0.1 * r.standard_normal(pd)).astype(np.float32)
    pw = (r.standard_normal(D) / np.sqrt(D)).astype(np.float32)
    pb = np.array([0.3], np.float32)
    dec = np.zeros((M, pd), np.float32)
    srow = np.zeros(M, np.float32)
    ms = C.c_float(0)
    P = lambda a: a.ctypes.data
    ctx.check(ctx.lib.dd_dev_head_dec(ctx.handle, M, D, pd, tok_l, tok_e, P(x), P(g), P(b), P(w), P(bias), P(dec),
                                      P(pw) if probe else None, P(pb) if probe else None, P(srow) if probe else None, split, 3, None, C.byref(ms)))
    want = _reference(x, g, b, w, bias)
    with np.errstate(over="ignore"):
        want_s = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ pw.astype(np.float64) + 0.3)))
    return dec, srow, want, want_s, ms.value * 1e3


@pytest.mark.parametrize("D,pd,B,L,extras,probe", [(512, 48, 5, 257, 1, True), (512, 48, 3, 258, 2, True), (512, 48, 4, 257, 1, False),
                                                   (256, 48, 3, 65, 1, True), (512, 64, 2, 257, 1, True), (512, 12, 3, 258, 2, False),
                                                   (768, 12, 3, 257, 1, False), (768, 48, 2, 258, 2, False), (1024, 16, 3, 258, 2, False),
                                                   (512, 48, 64, 257, 1, True), (512, 48, 128, 257, 1, False)])
def test_head_dec_patch_rows_against_float64_reference(D, pd, B, L, extras, probe):
    """Patch-rows-only launches (what the models make): rows with offsets of 0, 50 and 2 000 times their spread, and rows 1e-3 wide."""
    M = B * L
    r = np.random.default_rng(M + D)
    x = r.standard_normal((M, D)).astype(np.float32)
    x[1::4] += 50.0
    x[2::4] += 2000.0                       # (fp32 rows: the offset costs the INPUT log2(2000) = 11 bits; what is left must survive the kernel)
    x[3::4] *= 1e-3
    dec, srow, want, want_s, us = _run(M, D, pd, L, extras, x, probe)
    rows = np.arange(M) % L
    patch = rows >= extras
    assert np.isnan(dec[~patch]).all(), "the extra tokens' rows are not decoded"
    err = np.abs(dec[patch] - want[patch])
    scale = np.abs(want[patch]).max()
    print(f"head_dec D={D} pd={pd} B={B} L={L}: max err {err.max():.3e} of |dec| max {scale:.2f}; rows with offset 2000: {err[(np.arange(M) % 4 == 2)[patch]].max():.3e}; {us:.1f} us/launch")
    # fp32 accumulation of D products + the statistics: a few 1e-6 of the output's scale (measured 1.1e-6 .. 2.2e-6), offset rows included --
    # x - x[0] is exact for a row whose elements share an exponent, so the offset costs nothing beyond what it cost the fp32 INPUT
    assert err.max() <= 5e-6 * scale
    if probe:
        es = np.abs(srow - want_s)
        print(f"  probe rows: max err {es.max():.3e}")
        assert es.max() <= 1e-6


@pytest.mark.parametrize("D,pd,M,probe", [(512, 48, 16 * 300 + 7, True), (256, 16, 1000, True), (512, 48, 37, False), (768, 12, 16 * 40 + 3, False)])
def test_head_dec_every_row_ragged_and_more_units_than_waves(D, pd, M, probe):
    """tok_l = 0 launches: every row, a ragged last unit, and (M = 4 807 on a grid of 38 workgroups) waves that walk a second unit
    through the prefetch path."""
    r = np.random.default_rng(M)
    x = (r.standard_normal((M, D)) * 1.5 + r.standard_normal((M, 1)) * 3.0).astype(np.float32)
    dec, srow, want, want_s, us = _run(M, D, pd, 0, 0, x, probe)
    err = np.abs(dec - want)
    print(f"head_dec every row D={D} pd={pd} M={M}: max err {err.max():.3e} of |dec| max {np.abs(want).max():.2f}; {us:.1f} us/launch")
    assert err.max() <= 5e-6 * max(1.0, np.abs(want).max())
    if probe:
        assert np.abs(srow - want_s).max() <= 2e-6


@pytest.mark.parametrize("D,pd,B,L,extras,probe", [(512, 48, 5, 257, 1, True), (512, 48, 3, 258, 2, False), (256, 48, 3, 65, 1, True), (512, 64, 2, 257, 1, True),
                                                   (512, 12, 3, 258, 2, True), (256, 16, 4, 17, 1, False), (512, 48, 64, 257, 1, True)])
def test_head_dec_split_bf16_product_against_float64_reference(D, pd, B, L, extras, probe):
    """The early-exit heads of the bf16 engine: Wg . d as hi + lo bf16 halves and three bf16 MFMAs.  The dropped lo . lo term and the low
    halves' rounding are 2^-16 .. 2^-17 of a product: a few 1e-5 of the output's scale at most (the exact kernel: 2e-6), offset rows included
    -- the split is of d = x - x[0], not of x; the probe value comes from the fp32 rows as in the exact kernel."""
    M = B * L
    r = np.random.default_rng(M + D + 1)
    x = r.standard_normal((M, D)).astype(np.float32)
    x[1::4] += 50.0
    x[2::4] += 2000.0
    x[3::4] *= 1e-3
    dec, srow, want, want_s, us = _run(M, D, pd, L, extras, x, probe, split=1)
    exact, _, _, _, us_exact = _run(M, D, pd, L, extras, x, probe, split=0)
    patch = (np.arange(M) % L) >= extras
    assert np.isnan(dec[~patch]).all()
    err = np.abs(dec[patch] - want[patch])
    scale = np.abs(want[patch]).max()
    print(f"head_dec split D={D} pd={pd} B={B} L={L}: max err {err.max():.3e} of |dec| max {scale:.2f} (exact kernel {np.abs(exact[patch] - want[patch]).max():.3e}); "
          f"{us:.1f} us/launch against {us_exact:.1f} exact")
    assert err.max() <= 4e-5 * scale
    if probe:
        assert np.abs(srow - want_s).max() <= 1e-6
