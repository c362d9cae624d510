"""Oracle (test infrastructure): DDPM noise schedule tables and the reverse update.

Restates, in numpy fp32:
  * the module-level schedule of reference ``sampler.py:40-44``
  * ``NoiseScheduler.__init__`` of reference ``ddpm_core.py:56-70`` (same tables,
    but beta-tilde is multiplied in a different order, so it rounds differently)
  * ``predict_noise_postprocessing`` reference ``sampler.py:47-56``
    (== the update of ``ddpm_core.py:190-193``)

The tables are pinned bit-for-bit against tests/golden/schedule.npz (fixture F2).
"""
import numpy as np

F32 = np.float32


def _linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in fp32 (reference sampler.py:40).

    ATen computes ``step = (end - start) / (steps - 1)`` in fp32 and fills the
    first half forward from ``start`` and the second half backward from ``end``,
    each element with ONE rounding (vectorised fused multiply-add); evaluating
    ``start + step * i`` in double and rounding once reproduces that bit for bit
    (checked against fixture F2).
    """
    start, end = F32(start), F32(end)
    step = np.float64(F32((end - start) / F32(steps - 1)))
    idx = np.arange(steps)
    half = steps // 2
    lo = np.float64(start) + step * idx
    hi = np.float64(end) - step * (steps - 1 - idx)
    return np.where(idx < half, lo, hi).astype(F32)


def _cumprod_f32(a):
    """torch.cumprod on an fp32 CPU tensor (reference sampler.py:42).

    ATen's CPU scan accumulates in double and rounds every output to fp32.
    """
    return np.cumprod(a.astype(np.float64)).astype(F32)


def _base_tables(beta_init=1e-4, beta_final=0.02, beta_steps=1000):
    betas = _linspace_f32(beta_init, beta_final, beta_steps)
    alphas = (F32(1) - betas).astype(F32)
    alphas_bar = _cumprod_f32(alphas)
    alphas_bar_prev = np.concatenate([np.ones(1, F32), alphas_bar[:-1]]).astype(F32)
    return betas, alphas, alphas_bar, alphas_bar_prev


def sampler_schedule():
    """reference sampler.py:40-44 -> dict of five fp32[1000] tables."""
    betas, alphas, alphas_bar, alphas_bar_prev = _base_tables()
    # betas * (1 - abar_prev) / (1 - abar): left-to-right, each op rounded to fp32
    betas_tilde = ((betas * (F32(1) - alphas_bar_prev)).astype(F32)
                   / (F32(1) - alphas_bar)).astype(F32)
    return dict(betas=betas, alphas=alphas, alphas_bar=alphas_bar,
                alphas_bar_previous=alphas_bar_prev, betas_tilde=betas_tilde)


def scheduler_schedule(beta_init=1e-4, beta_final=0.02, beta_steps=1000):
    """reference ddpm_core.py:64-70 (NoiseScheduler.__init__)."""
    betas, alphas, alphas_bar, alphas_bar_prev = _base_tables(beta_init, beta_final, beta_steps)
    # (1 - abar_prev) / (1 - abar) * betas
    betas_tilde = (((F32(1) - alphas_bar_prev) / (F32(1) - alphas_bar)).astype(F32)
                   * betas).astype(F32)
    return dict(betas=betas, alphas=alphas, alphas_bar=alphas_bar,
                alpha_bar_prev=alphas_bar_prev, betas_tilde=betas_tilde)


def step_coefficients(tables, t, variance="beta_tilde"):
    """The three per-step scalars of sampler.py:48-56, each rounded as torch does.

    returns (c1, c2, sigma) with  x' = c1 * (x - c2 * eps) + sigma * z
    variance: "beta_tilde" (sampler.py:50) or "beta" (ddpm_core.py:72-75 default).
    """
    alpha_t = tables["alphas"][t]
    alpha_bar_t = tables["alphas_bar"][t]
    var_t = tables["betas_tilde"][t] if variance == "beta_tilde" else tables["betas"][t]
    c1 = np.sqrt(F32(1) / alpha_t, dtype=F32)
    c2 = F32((F32(1) - alpha_t) / np.sqrt(F32(1) - alpha_bar_t, dtype=F32))
    sigma = np.sqrt(var_t, dtype=F32)
    return F32(c1), F32(c2), F32(sigma)


def ddpm_step(x, eps, z, t, tables=None, variance="beta_tilde"):
    """reference sampler.py:47-56: x <- sqrt(1/a_t)(x - (1-a_t)/sqrt(1-abar_t) eps) + sigma_t z.

    ``z`` is ignored (treated as 0) when t == 0, as the reference does.
    """
    if tables is None:
        tables = sampler_schedule()
    c1, c2, sigma = step_coefficients(tables, t, variance)
    x = np.asarray(x, F32)
    eps = np.asarray(eps, F32)
    mean = (c1 * (x - (c2 * eps).astype(F32)).astype(F32)).astype(F32)
    if t > 0 and z is not None:
        return (mean + (sigma * np.asarray(z, F32)).astype(F32)).astype(F32)
    return mean


def predict_original_step(x, m, z, t, tables=None):
    """reference sampler.py:59-72 (model predicts x_0)."""
    tb = tables or sampler_schedule()
    a_t, ab_t, ab_p, b_t = tb["alphas"][t], tb["alphas_bar"][t], tb["alphas_bar_previous"][t], tb["betas"][t]
    sigma = np.sqrt(tb["betas_tilde"][t], dtype=F32)
    x, m = np.asarray(x, F32), np.asarray(m, F32)
    t1 = ((np.sqrt(ab_p, dtype=F32) * b_t).astype(F32) * m).astype(F32) / (F32(1) - ab_t)
    t2 = ((np.sqrt(a_t, dtype=F32) * (F32(1) - ab_p)).astype(F32) * x).astype(F32) / (F32(1) - ab_t)
    out = (t1.astype(F32) + t2.astype(F32)).astype(F32)
    if t > 0 and z is not None:
        out = (out + (sigma * np.asarray(z, F32)).astype(F32)).astype(F32)
    return out


def predict_previous_step(x, m, z, t, tables=None):
    """reference sampler.py:75-79 (model predicts x_{t-1})."""
    tb = tables or sampler_schedule()
    sigma = np.sqrt(tb["betas_tilde"][t], dtype=F32)
    out = np.asarray(m, F32)
    if t > 0 and z is not None:
        out = (out + (sigma * np.asarray(z, F32)).astype(F32)).astype(F32)
    return out


def ddim_step(x, m, z, t, s, eta, tables=None):
    """One DDIM update t -> s of reference sampler.py:112-120 (note: noise is scaled by sigma^2, as there)."""
    tb = tables or sampler_schedule()
    ab = tb["alphas_bar"]
    sig2 = F32(tb["betas_tilde"][t] * F32(eta))
    x, m = np.asarray(x, F32), np.asarray(m, F32)
    mean = (np.sqrt(ab[s] / ab[t], dtype=F32) * (x - (np.sqrt(F32(1) - ab[t], dtype=F32) * m).astype(F32)).astype(F32)).astype(F32)
    mean = (mean + (np.sqrt(F32(1) - ab[s] - sig2, dtype=F32) * m).astype(F32)).astype(F32)
    if s > 0 and z is not None:
        return (mean + (sig2 * np.asarray(z, F32)).astype(F32)).astype(F32)
    return mean
