"""CPU: the numpy oracle against golden vectors produced by the reference itself.

These pin the oracle (SURVEY section 8c fixtures F1-F6).  Tolerances: schedule tables
bit-exact; everything else fp32 re-association noise (<= 2e-5 abs on O(1) values).
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import FULL_NAMES, REPO, tiny_cfg_from_fixture
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict


def _params(mp, seed):
    return {k: v.numpy() for k, v in synthetic_state_dict(mp, seed).items()}


def test_schedule_bit_exact(golden):
    fx = golden("schedule.npz")
    s = oracle.sampler_schedule()
    for k in ("betas", "alphas", "alphas_bar", "alphas_bar_previous", "betas_tilde"):
        assert np.array_equal(s[k], fx["sampler_" + k]), k
    d = oracle.scheduler_schedule()
    for k in ("betas", "alphas", "alphas_bar", "alpha_bar_prev", "betas_tilde"):
        assert np.array_equal(d[k], fx["sched_" + k]), k
    # the two files round beta-tilde differently
    assert (s["betas_tilde"] != d["betas_tilde"]).sum() > 100
    assert s["betas_tilde"][0] == 0.0


def test_step_matches_reference(golden):
    fx = golden("step.npz")
    for t in fx["ts"]:
        t = int(t)
        got = oracle.ddpm_step(fx["x"], fx["eps"], fx[f"z_{t}"], t)
        np.testing.assert_allclose(got, fx[f"xnext_{t}"], rtol=0, atol=5e-7)


def test_rng_stream(golden):
    fx = golden("rng.npz")
    oracle.seed_everything(0)
    assert np.array_equal(torch.randn(64).numpy(), fx["first64"])
    oracle.seed_everything(0)
    x = torch.randn(2, 3, 8, 8).numpy()
    z = torch.randn(2, 3, 8, 8).numpy()
    assert np.array_equal(x, fx["x_T"]) and np.array_equal(z, fx["z_first"])
    # contiguity: one draw of 2N == two draws of N
    oracle.seed_everything(0)
    both = torch.randn(2 * x.size).numpy()
    assert np.array_equal(both[: x.size], x.ravel()) and np.array_equal(both[x.size:], z.ravel())


@pytest.mark.parametrize("name", ["uncond_norm", "uncond_raw", "cond_raw", "cond_norm_h2", "timemlp_qkvbias", "cond_timemlp"])
def test_tiny_per_op(golden, name):
    fx = golden(f"uvit_tiny_{name}.npz")
    cfg = tiny_cfg_from_fixture(fx)
    mp = ModelParams.from_dict(cfg)
    m = oracle.UViTOracle(cfg, _params(mp, int(fx["seed"])))
    taps = {}
    y = fx["y"] if "y" in fx.files else None
    eps = m(fx["x"], fx["t"], y, taps=taps)
    checked = 0
    for k in fx.files:
        if not k.startswith("tap_"):
            continue
        key = k[4:]
        if key in taps:
            np.testing.assert_allclose(taps[key], fx[k], rtol=0, atol=2e-5, err_msg=key)
            checked += 1
    assert checked >= mp.depth + 2
    np.testing.assert_allclose(eps, fx["eps"], rtol=0, atol=2e-5)
    # sub-module taps of the first block
    p = m.p
    tok = fx["tap_tokens"]
    a = oracle.attention(oracle.layer_norm(tok, p["in_blocks.0.norm1.weight"], p["in_blocks.0.norm1.bias"]),
                         p, "in_blocks.0.attn.", mp.num_heads)
    np.testing.assert_allclose(a, fx["tap_in_blocks.0.attn"], rtol=0, atol=2e-5)


def test_rollout_tiny(golden):
    fx = golden("rollout_tiny.npz")
    from conftest import TINY
    mp_s = ModelParams.from_dict(dict(TINY, depth=1))
    mp_f = ModelParams.from_dict(dict(TINY, depth=3))
    m_s = oracle.UViTOracle(mp_s.as_dict(), _params(mp_s, int(fx["seed_first"])))
    m_f = oracle.UViTOracle(mp_f.as_dict(), _params(mp_f, int(fx["seed_late"])))
    rec = {"_want": (999, 998, 700, 699, 1, 0)}
    samples, _ = oracle.get_samples(m_s, 2, 0, 3, 8, 8, late_model=m_f, t_switch=300, record=rec)
    assert m_s.calls == int(fx["calls_first"]) == 300
    assert m_f.calls == int(fx["calls_late"]) == 700
    np.testing.assert_allclose(rec[999], fx["x_after_999"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rec[998], fx["x_after_998"], rtol=0, atol=1e-5)
    # free-running fp32 trajectories drift by re-association; scale tolerance with |x|
    for t in (700, 699, 1, 0):
        ref = fx[f"x_after_{t}"]
        tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(rec[t], ref, rtol=0, atol=tol)
    assert samples.shape == (2, 8, 8, 3) and samples.dtype == np.float32
    np.testing.assert_allclose(samples, fx["samples"], rtol=0,
                               atol=2e-3 * max(1.0, float(np.abs(fx["samples"]).max())))


def test_scheduler_tiny(golden):
    fx = golden("scheduler_tiny.npz")
    from conftest import TINY
    mp = ModelParams.from_dict(dict(TINY, depth=1))
    m = oracle.UViTOracle(mp.as_dict(), _params(mp, int(fx["seed"])))
    # 50-step schedule in both variance modes, and the default 1000-step schedule (sigma^2 = beta)
    for tag, steps, mode in (("", 50, "beta"), ("_bt50", 50, "beta_tilde"), ("_b1000", 1000, "beta")):
        x0, over = oracle.scheduler_sample(m, steps, (3, 8, 8), 2, seed=5, variance_mode=mode)
        np.testing.assert_allclose(over[0], fx["x_after_first" + tag], rtol=0, atol=1e-5)
        for got, key in ((over[steps // 2], "x_mid" + tag), (x0, "x0" + tag)):
            np.testing.assert_allclose(got, fx[key], rtol=0, atol=2e-4 * max(1.0, float(np.abs(fx[key]).max())), equal_nan=False)
    assert not np.array_equal(fx["x0"], fx["x0_bt50"])          # the variance mode is really exercised


@pytest.mark.parametrize("name", FULL_NAMES)
def test_full_size_forward(golden, name):
    fx = golden(f"uvit_full_{name}.npz")
    cfg = load_config(REPO / "configs" / f"{name}.yaml")
    mp = ModelParams.from_dict(cfg)
    m = oracle.UViTOracle(mp.as_dict(), _params(mp, int(fx["seed"])))
    B = fx["x"].shape[0]
    t = np.full((B,), float(fx["t"]), np.float32)
    y = fx["y"] if fx["y"].size else None
    eps = m(fx["x"], t, y)
    np.testing.assert_allclose(eps[:, :, :16, :16], fx["eps_slice"], rtol=0, atol=5e-5)
    st = fx["stats"]
    assert abs(eps.mean(dtype=np.float64) - st[0]) < 1e-5
    assert abs(eps.std(dtype=np.float64) - st[1]) < 1e-5
    assert abs(np.abs(eps.astype(np.float64)).sum() - float(fx["abs_checksum"])) < 1e-6 * eps.size * 50


@pytest.mark.parametrize("name", ["uvit_celeba", "uvit_imagenet64_3", "uvit_imagenet256_3"])
def test_torch_functional_oracle_full_size(golden, name):
    """The torch-functional variant (bench.py's CPU baseline) is pinned by the same vectors."""
    fx = golden(f"uvit_full_{name}.npz")
    mp = ModelParams.from_dict(load_config(REPO / "configs" / f"{name}.yaml"))
    m = oracle.UViTTorchOracle(mp.as_dict(), synthetic_state_dict(mp, int(fx["seed"])))
    B = fx["x"].shape[0]
    eps = m(fx["x"], np.full((B,), float(fx["t"]), np.float32), fx["y"] if fx["y"].size else None)
    np.testing.assert_allclose(eps[:, :, :16, :16], fx["eps_slice"], rtol=0, atol=2e-5)
    assert abs(eps.astype(np.float64).sum() - float(fx["checksum"])) < 1e-5 * eps.size


@pytest.mark.parametrize("name", ["uncond_norm", "cond_raw"])
def test_torch_functional_oracle_tiny(golden, name):
    fx = golden(f"uvit_tiny_{name}.npz")
    cfg = tiny_cfg_from_fixture(fx)
    mp = ModelParams.from_dict(cfg)
    m = oracle.UViTTorchOracle(cfg, synthetic_state_dict(mp, int(fx["seed"])))
    eps = m(fx["x"], fx["t"], fx["y"] if "y" in fx.files else None)
    np.testing.assert_allclose(eps, fx["eps"], rtol=0, atol=1e-5)


def test_other_parametrizations_and_ddim_vs_reference(golden):
    """predict_original / predict_previous updates and the DDIM branch."""
    fx = golden("param_steps.npz")
    for t in fx["ts"]:
        t = int(t)
        np.testing.assert_allclose(oracle.predict_original_step(fx["x"], fx["m"], fx[f"z_{t}"], t), fx[f"orig_{t}"],
                                   rtol=0, atol=2e-6 * max(1.0, float(np.abs(fx[f"orig_{t}"]).max())))
        np.testing.assert_allclose(oracle.predict_previous_step(fx["x"], fx["m"], fx[f"z_{t}"], t), fx[f"prev_{t}"],
                                   rtol=0, atol=1e-6)
    fd = golden("ddim_tiny.npz")
    from conftest import TINY
    mp_s, mp_f = ModelParams.from_dict(dict(TINY, depth=1)), ModelParams.from_dict(dict(TINY, depth=3))
    for tag in ("a", "b", "c", "nan"):
        steps, eta, tsw = fd[f"cfg_{tag}"]
        m_s = oracle.UViTOracle(mp_s.as_dict(), _params(mp_s, 300))
        m_f = oracle.UViTOracle(mp_f.as_dict(), _params(mp_f, 301))
        with np.errstate(invalid="ignore"):
            samples, inter = oracle.get_samples_ddim(m_s, 2, 3, 3, 8, 8, ddim_steps=int(steps), ddim_eta=float(eta),
                                                     timesteps_save=[1], late_model=m_f, t_switch=int(tsw))
        assert m_s.calls + m_f.calls == int(steps) - 1 and m_f.calls > 0
        np.testing.assert_allclose(inter[0], fd[f"first_{tag}"], rtol=0, atol=1e-5, equal_nan=False)
        if tag == "nan":
            # the reference's sigma^2-for-sigma quirk: sqrt(1 - abar_0 - betas_tilde[t]*eta) < 0 at the last pair -> all NaN
            assert float(fd["nan_fraction"]) == 1.0 and np.isnan(samples).all()
            continue
        scale = max(1.0, float(np.abs(fd[f"samples_{tag}"]).max()))
        # 499 free-running steps (case b) amplify fp32 rounding differences: drift-scaled tolerance
        np.testing.assert_allclose(samples, fd[f"samples_{tag}"], rtol=0, atol=(2e-3 if tag == "b" else 1e-4) * scale, equal_nan=False)


def test_affine_coefficients_reproduce_reference_updates(golden):
    """Host-side scalar coefficients (duodiff_amd.sampler.affine_coefficients) against the reference's outputs."""
    from duodiff_amd import sampler
    fx = golden("param_steps.npz")
    for t in fx["ts"]:
        t = int(t)
        for kind, key in (("predict_original", "orig"), ("predict_previous", "prev")):
            a, b, c = sampler.affine_coefficients(kind, t)
            got = a * fx["x"] + b * fx["m"] + (c * fx[f"z_{t}"] if t > 0 else 0)
            ref = fx[f"{key}_{t}"]
            np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 * max(1.0, float(np.abs(ref).max())))
    a, b, c = sampler.affine_coefficients("ddim", 999, 946, 0.5)
    x, m, z = fx["x"], fx["m"], fx["z_999"]
    np.testing.assert_allclose(a * x + b * m + c * z, oracle.ddim_step(x, m, z, 999, 946, 0.5), rtol=0, atol=2e-5)
