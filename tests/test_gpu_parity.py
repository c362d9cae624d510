"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors
and the numpy oracle, on the same seeded inputs.

Tolerances (stated per SURVEY section 7.3 H3 / BASELINE north_star "1e-3 max-abs"):
  fp32 mode  (f32 MFMA, exact fp32 products): eps within 1e-4 of the reference (observed ~1e-5)
  bf16 mode  (bf16 MFMA operands, fp32 accumulate / residual / LN / softmax):
             one sampling step x_{t-1} within 1e-3 of the reference (teacher-forced);
             eps itself within 3e-2 * std(eps) max-abs (observed 1.6 - 2.7e-2 * std; the reference's own bf16
             autocast misses 1e-3 on eps too: 8.4e-3) -- reported, not hidden;
             the benchmarked path (dd_sample, hipGraph replay, CelebA pair, B = 128) free-running for K = 1, 10, 100
             steps against the fp32 engine: test_benchmarked_path_drift_bf16_vs_fp32.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import FULL_NAMES, REPO, TINY, tiny_cfg_from_fixture
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu

STEP_TOL = 1e-3


def eps_tol(precision, ref, full_size=False):
    """max-abs bound on eps: absolute in the exact-fp32 mode; in bf16 mode a LOOSE guard relative to the spread of the reference
    output -- 3e-2 sigma for the tiny fixtures, 5e-2 sigma for the full-size cases (observed 1.4 .. 3.4e-2 sigma: the maximum of a
    1536-value slice moves by +-5 % with every change of accumulation order in a kernel, so it is not the regression gate).
    The gate is the rms error against the error model of eps_rms_bound (fixed margin, no per-config constants)."""
    return 1e-4 if precision == "fp32" else (5e-2 if full_size else 3e-2) * float(np.asarray(ref, np.float64).std())


# bf16 engine vs the reference: the gate on rms(eps - ref) / sigma(ref) is an ERROR MODEL, not a fit to the last run.  Every block rounds
# ~6 GEMM operands to bf16 (norm1 rows, the attention output, norm2 rows, the GELU'd hidden rows, the long-skip copy, P of the attention
# core), each with relative rms 2^-9 / sqrt(3) (uniform rounding error of an 8-bit significand), and the roundings of the `depth` blocks
# add in quadrature on a residual stream of spread ~sigma:  model = 2^-9 / sqrt(3) * sqrt(6 depth)  =  4.8e-3 (depth 3) .. 9.9e-3 (13)
# .. 1.27e-2 (21).  Observed on MI355X (profiles/r05/parity_numbers.txt): 0.47 .. 0.96 of the model.  The bound is the model x 1.5, fixed.
EPS_RMS_MODEL_MARGIN = 1.5


def eps_rms_bound(depth):
    return EPS_RMS_MODEL_MARGIN * 2.0 ** -9 / np.sqrt(3.0) * np.sqrt(6.0 * depth)


def _uvit(cfg, seed, precision, max_batch=None):
    from duodiff_amd.uvit import UViT
    mp = ModelParams.from_dict(cfg)
    m = UViT(**mp.as_dict(), precision=precision, max_batch=max_batch)
    m.load_state_dict(synthetic_state_dict(mp, seed))
    return m.eval().to("cuda"), mp


def _oracle(cfg, seed):
    mp = ModelParams.from_dict(cfg)
    return oracle.UViTOracle(mp.as_dict(), {k: v.numpy() for k, v in synthetic_state_dict(mp, seed).items()})


def test_library_loaded_in_tree():
    from duodiff_amd import _lib
    lib = _lib.load()
    assert str(_lib.LIB_PATH).endswith("duodiff_amd/libduodiff.so") and lib.dd_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["uncond_norm", "uncond_raw", "cond_raw", "cond_norm_h2", "timemlp_qkvbias", "cond_timemlp"])
def test_forward_tiny_vs_reference(golden, name, precision):
    fx = golden(f"uvit_tiny_{name}.npz")
    cfg = tiny_cfg_from_fixture(fx)
    m, mp = _uvit(cfg, int(fx["seed"]), precision)
    y = torch.from_numpy(fx["y"]) if "y" in fx.files else None
    eps = m(torch.from_numpy(fx["x"]), torch.from_numpy(fx["t"]), y).cpu().numpy()
    err = np.abs(eps - fx["eps"]).max()
    print(f"tiny {name} {precision}: max|eps - ref| = {err:.3e} (std {fx['eps'].std():.3f})")
    assert err <= eps_tol(precision, fx["eps"])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", FULL_NAMES)
def test_forward_full_size_vs_reference(golden, name, precision):
    fx = golden(f"uvit_full_{name}.npz")
    cfg = load_config(REPO / "configs" / f"{name}.yaml")
    m, mp = _uvit(cfg, int(fx["seed"]), precision)
    B = fx["x"].shape[0]
    t = torch.full((B,), float(fx["t"]))
    y = torch.from_numpy(fx["y"]) if fx["y"].size else None
    eps = m(torch.from_numpy(fx["x"]), t, y).cpu().numpy()
    assert np.isfinite(eps).all()
    diff = eps[:, :, :16, :16] - fx["eps_slice"]
    err, rms, sigma = np.abs(diff).max(), float(np.sqrt((diff.astype(np.float64) ** 2).mean())), float(fx["eps_slice"].std())
    st = fx["stats"]
    print(f"{name} {precision}: max|eps - ref| (slice) = {err:.3e}, rms {rms:.3e} ({rms / sigma:.2e} sigma); std {eps.std():.4f} vs {st[1]:.4f}")
    assert err <= eps_tol(precision, fx["eps_slice"], full_size=True)
    if precision == "bf16":
        assert rms <= eps_rms_bound(mp.depth) * sigma, f"rms {rms / sigma:.3e} sigma vs the error model's bound {eps_rms_bound(mp.depth):.2e} (depth {mp.depth})"
    assert abs(eps.std(dtype=np.float64) - st[1]) <= (1e-4 if precision == "fp32" else 5e-3)
    assert abs(eps.astype(np.float64).sum() - float(fx["checksum"])) <= (1e-5 if precision == "fp32" else 2e-3) * eps.size


def test_ddpm_step_matches_reference(golden):
    from duodiff_amd.engine import Context
    fx = golden("step.npz")
    ctx = Context.get()
    x, eps = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["eps"]).cuda()
    for t in fx["ts"]:
        t = int(t)
        z = torch.from_numpy(fx[f"z_{t}"]).cuda()
        got = ctx.ddpm_step(x, eps, z, t).cpu().numpy()
        want_o = oracle.ddpm_step(fx["x"], fx["eps"], fx[f"z_{t}"], t)
        assert np.array_equal(got, want_o), f"t={t}: device update differs from the oracle bitwise"
        np.testing.assert_allclose(got, fx[f"xnext_{t}"], rtol=0, atol=5e-7)
    # ddpm_core variance mode (sigma^2 = beta)
    got = ctx.ddpm_step(x, eps, z, 500, variance="beta").cpu().numpy()
    want = oracle.ddpm_step(fx["x"], fx["eps"], z.cpu().numpy(), 500, oracle.scheduler_schedule(), variance="beta")
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", FULL_NAMES)
def test_teacher_forced_step_within_1e3(golden, name, precision):
    """x_{t-1} = postprocessing(model(x_t, t), x_t, t) from the SAME x_t, z: the boundary the
    reference loop composes 1000 times.  1e-3 max-abs in both precisions."""
    fx = golden(f"uvit_full_{name}.npz")
    cfg = load_config(REPO / "configs" / f"{name}.yaml")
    seed = int(fx["seed"])
    m, mp = _uvit(cfg, seed, precision)
    orc = _oracle(cfg, seed)
    B = 2
    x = fx["x"]
    y = fx["y"] if fx["y"].size else None
    g = torch.Generator().manual_seed(5)
    for t in (999, 699, 1, 0):
        z = torch.randn(x.shape, generator=g).numpy()
        eps_o = orc(x, np.full((B,), t, np.float32), y)
        want = oracle.ddpm_step(x, eps_o, z, t)
        xd = torch.from_numpy(x).cuda().contiguous()
        em = m.engine_model(B)
        em.sample_step(xd, t, y=(torch.from_numpy(y).cuda() if y is not None else None),
                       z=torch.from_numpy(z).cuda(), noise="buffer")
        err = np.abs(xd.cpu().numpy() - want).max()
        print(f"{name} {precision} t={t}: max|x' - ref| = {err:.3e}")
        assert err <= STEP_TOL


def test_rollout_tiny_fp32_vs_reference(golden):
    """get_samples with late_model, t_switch=300, seed 0, B=2 against the reference's own rollout."""
    from duodiff_amd import sampler
    fx = golden("rollout_tiny.npz")
    m_s, _ = _uvit(dict(TINY, depth=1), int(fx["seed_first"]), "fp32")
    m_f, _ = _uvit(dict(TINY, depth=3), int(fx["seed_late"]), "fp32")
    samples, inter = sampler.get_samples(m_s, 2, sampler.predict_noise_postprocessing, 0, 3, 8, 8,
                                         timesteps_save=[1, 2, 301], late_model=m_f, t_switch=300,
                                         noise="torch_cpu")
    assert samples.shape == (2, 8, 8, 3) and samples.dtype == np.float32
    # intermediates are appended in loop order: after t=999 (1000-t=1), t=998, t=699 (301)
    to_img = lambda a: ((a + 1) / 2).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(inter[0], to_img(fx["x_after_999"]), rtol=0, atol=1e-5)
    np.testing.assert_allclose(inter[1], to_img(fx["x_after_998"]), rtol=0, atol=1e-5)
    scale = max(1.0, float(np.abs(fx["x_after_699"]).max()))
    np.testing.assert_allclose(inter[2], to_img(fx["x_after_699"]), rtol=0, atol=2e-3 * scale)
    scale = max(1.0, float(np.abs(fx["samples"]).max()))
    np.testing.assert_allclose(samples, fx["samples"], rtol=0, atol=2e-3 * scale)


def test_switch_and_graph_replay_match_manual_steps():
    """dd_sample (hipGraph replay, device Philox noise, backbone switch) == the same steps issued
    one by one through dd_sample_step: bit-exact, and the switch happens AFTER t == 1000 - t_switch."""
    from duodiff_amd.engine import sample_loop
    cfg_s, cfg_f = dict(TINY, depth=1), dict(TINY, depth=3)
    m_s, _ = _uvit(cfg_s, 11, "bf16")
    m_f, _ = _uvit(cfg_f, 12, "bf16")
    B = 4
    es, ef = m_s.engine_model(B), m_f.engine_model(B)
    x0 = torch.randn(B, 3, 8, 8, generator=torch.Generator().manual_seed(3)).cuda()
    stream = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(stream):
        for use_graph in (True, False):
            x = x0.clone()
            sample_loop(es.ctx, es, ef, x, t_switch=5, t_start=999, t_end=988, seed=77, noise="philox",
                        use_graph=use_graph, stream=stream)
            stream.synchronize()
            outs.append(x.clone())
        xm = x0.clone()
        for t in range(999, 987, -1):
            (es if t >= 995 else ef).sample_step(xm, t, noise="philox", seed=77, stream=stream)
        stream.synchronize()
    assert torch.equal(outs[0], outs[1]), "graph replay differs from eager launches"
    assert torch.equal(outs[0], xm), "loop differs from manual steps (switch placement?)"
    assert torch.isfinite(xm).all()


@pytest.mark.parametrize("case", ["tiny_forced", "tiny_cond_forced", "celeba_default", "imagenet64_gemm_path"])
def test_half_batch_chains_equal_the_single_chain(case):
    """dd_sample runs an even batch of >= 32 images as two half-batch chains on two streams (each with its own workspace, step
    state and captured graphs; images are independent and a row's path through the kernels does not depend on the batch, the Philox
    pixel ids carry the image offset).  The result must equal the single-chain loop bit for bit: backbone switch inside the run,
    device noise; tiny models with the split forced at B = 6 (unconditional and class-conditional: the label half moves too),
    the CelebA pair at its benchmark batch with the default policy, and two ImageNet-64 width models (embed_dim 768: the GEMM
    sequence) at B = 256, where both chains' persistent GEMM grids are sized for half the CUs (another row partition, same values)."""
    from duodiff_amd import _lib as L
    from duodiff_amd.engine import sample_loop
    if case == "celeba_default":
        B, S, C_, steps, tsw, force = 128, 64, 3, 6, 3, 0
        cfg_s, cfg_f = load_config(REPO / "configs" / "uvit_celeba_3.yaml"), load_config(REPO / "configs" / "uvit_celeba.yaml")
    elif case == "imagenet64_gemm_path":
        B, S, C_, steps, tsw, force = 256, 64, 3, 4, 2, 0
        cfg_s = cfg_f = load_config(REPO / "configs" / "uvit_imagenet64_3.yaml")
    else:
        B, S, C_, steps, tsw, force = 6, 8, 3, 12, 5, L.DD_DEV_FORCE_CHAINS
        nc = 10 if case == "tiny_cond_forced" else -1
        cfg_s, cfg_f = dict(TINY, depth=1, num_classes=nc), dict(TINY, depth=3, num_classes=nc)
    m_s, _ = _uvit(cfg_s, 31, "bf16", max_batch=B)
    m_f, mp_f = _uvit(cfg_f, 32, "bf16", max_batch=B)
    es, ef = m_s.engine_model(B), m_f.engine_model(B)
    ctx = es.ctx
    x0 = torch.randn(B, C_, S, S, generator=torch.Generator().manual_seed(4)).cuda()
    ncls = int(mp_f.num_classes)
    y = torch.randint(0, ncls, (B,), generator=torch.Generator().manual_seed(5)).cuda() if ncls > 0 else None
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    outs = {}
    try:
        with torch.cuda.stream(stream):
            for name, flags in (("chained", force), ("single", L.DD_DEV_NO_CHAINS)):
                ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
                n0 = ctx.lib.dd_dev_graph_captures(ctx.handle)
                x = x0.clone()
                sample_loop(ctx, es, ef, x, t_switch=tsw, t_start=999, t_end=1000 - steps, y=y, seed=9, noise="philox", use_graph=True, stream=stream)
                stream.synchronize()
                outs[name] = (x.clone(), ctx.lib.dd_dev_graph_captures(ctx.handle) - n0)
    finally:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
    assert outs["chained"][1] == 4 and outs["single"][1] == 2, f"graph captures: {outs['chained'][1]} chained, {outs['single'][1]} single"
    assert torch.isfinite(outs["single"][0]).all() and not torch.equal(outs["single"][0], x0)
    assert torch.equal(outs["chained"][0], outs["single"][0]), "two half-batch chains differ from the single chain"


@pytest.mark.parametrize("case", ["tiny_forced", "tiny_cond_forced", "celeba_default", "width768_forced", "width1024_forced", "tiny_fp32_single"])
def test_no_kernel_depends_on_stale_workspace_bytes(case):
    """The class of bug behind the round-4 chain-workspace race: kernels that read bytes of the activation workspace which nothing in the
    call wrote (zero-initialised padding, slabs of an earlier call).  Both chains' workspaces are filled with NaN bytes ON THE LAUNCH
    STREAM right before a model pair's first dd_sample (dd_dev_poison_workspaces allocates the second chain's workspace first, so the
    call does not zero it again); the samples must equal those of fresh, zero-initialised models bit for bit."""
    from duodiff_amd import _lib as L
    from duodiff_amd.engine import sample_loop
    prec, flags = "bf16", L.DD_DEV_FORCE_CHAINS
    if case == "celeba_default":
        B, S, C_, steps, tsw, flags = 128, 64, 3, 4, 2, 0
        cfg_s, cfg_f = load_config(REPO / "configs" / "uvit_celeba_3.yaml"), load_config(REPO / "configs" / "uvit_celeba.yaml")
    elif case in ("width768_forced", "width1024_forced"):     # the row-resident / split-K launches and their slabs, one and two extra tokens
        B, S, C_, steps, tsw = 4, 32, 3, 3, 1
        D = 768 if case == "width768_forced" else 1024
        base = dict(img_size=32, patch_size=2, in_chans=3, embed_dim=D, num_heads=D // 64, mlp_ratio=4, qkv_bias=False, mlp_time_embed=False,
                    num_classes=10 if D == 768 else -1, normalize_timesteps=True)
        cfg_s, cfg_f = dict(base, depth=1), dict(base, depth=3)
    else:
        B, S, C_, steps, tsw = 6, 8, 3, 8, 3
        nc = 10 if case == "tiny_cond_forced" else -1
        cfg_s, cfg_f = dict(TINY, depth=1, num_classes=nc), dict(TINY, depth=3, num_classes=nc)
        if case == "tiny_fp32_single":
            prec, flags = "fp32", L.DD_DEV_NO_CHAINS
    x0 = torch.randn(B, C_, S, S, generator=torch.Generator().manual_seed(14)).cuda()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    outs = []
    for poison in (False, True):
        m_s, _ = _uvit(cfg_s, 131, prec, max_batch=B)
        m_f, mp_f = _uvit(cfg_f, 132, prec, max_batch=B)
        es, ef = m_s.engine_model(B), m_f.engine_model(B)
        ctx = es.ctx
        ncls = int(mp_f.num_classes)
        y = torch.randint(0, ncls, (B,), generator=torch.Generator().manual_seed(15)).cuda() if ncls > 0 else None
        try:
            ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
            with torch.cuda.stream(stream):
                if poison:
                    for e in (es, ef):
                        ctx.check(ctx.lib.dd_dev_poison_workspaces(ctx.handle, e.handle, stream.cuda_stream))
                x = x0.clone()
                sample_loop(ctx, es, ef, x, t_switch=tsw, t_start=999, t_end=1000 - steps, y=y, seed=19, noise="philox", use_graph=True, stream=stream)
                stream.synchronize()
            outs.append((x.clone(), ctx.lib.dd_dev_last_sample_chains(ctx.handle)))
        finally:
            ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
        del es, ef, m_s, m_f
    assert outs[0][1] == outs[1][1] == (1 if flags == L.DD_DEV_NO_CHAINS else 2)
    assert torch.isfinite(outs[0][0]).all() and not torch.equal(outs[0][0], x0)
    bad = (outs[0][0] != outs[1][0]).flatten(1).any(1).nonzero().flatten().tolist()
    assert not bad, f"{case}: images {bad[:8]}... differ after the workspaces were poisoned (a kernel reads bytes no launch of the call wrote)"


def test_profile_steps_chained_counts_and_grid():
    """dd_profile_steps_chained (bench.py's `chained` roofline leg) brackets every launch of the selected kernel in BOTH chains:
    2 x steps x depth event pairs for the block tail, and it runs on the chain-sized GEMM grids dd_sample uses."""
    cfg = load_config(REPO / "configs" / "uvit_celeba_3.yaml")
    m, mp = _uvit(cfg, 7, "bf16", max_batch=64)
    em = m.engine_model(64)
    x = torch.randn(64, 3, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        ms1, n1 = em.profile_steps(x.clone(), t_start=699, steps=3, stream=stream)
        ms2, n2 = em.profile_steps_chained(x.clone(), t_start=699, steps=3, stream=stream)
        ms3, n3 = em.profile_steps(x.clone(), t_start=699, steps=2, stream=stream, kind="qkv_attention")
        ms4, n4 = em.profile_steps(x.clone(), t_start=699, steps=2, stream=stream, kind="rowlin")     # (no such launch in this model)
    assert n1 == 3 * mp.depth and n2 == 2 * 3 * mp.depth and n3 == 2 * mp.depth and n4 == 0
    assert 0 < ms2 < ms1 * 1.5 and ms3 > 0 and ms4 == 0


def test_half_batch_chains_in_the_table_driven_loops():
    """dd_sample_affine (DDIM with eta > 0, predict_original, predict_previous: device Philox noise, backbone switch) split into two
    half-batch chains must equal the single-chain loop bit for bit (both chains read the one step table; the Philox ids carry the
    image offset)."""
    from duodiff_amd import _lib as L
    from duodiff_amd import sampler
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "bf16", max_batch=6)
    m_f, _ = _uvit(dict(TINY, depth=3), 301, "bf16", max_batch=6)
    ctx = m_s.engine_model(6).ctx
    run = lambda **kw: sampler.get_samples(m_s, 6, kw.pop("post", sampler.predict_noise_postprocessing), 5, 3, 8, 8,
                                           late_model=m_f, noise="device", **kw)
    try:
        for kw in (dict(use_ddim=True, ddim_steps=30, ddim_eta=0.05, t_switch=400),
                   dict(post=sampler.predict_original_postprocessing, num_steps=40, t_switch=980),
                   dict(post=sampler.predict_previous_postprocessing, num_steps=30, t_switch=985, timesteps_save=[4, 20])):
            outs = []
            for flags in (L.DD_DEV_FORCE_CHAINS, L.DD_DEV_NO_CHAINS):
                ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
                g, gi = run(**dict(kw))
                outs.append((g, gi, ctx.lib.dd_dev_last_sample_chains(ctx.handle)))
            assert outs[0][2] == 2 and outs[1][2] == 1
            assert np.isfinite(outs[1][0]).all() and np.array_equal(outs[0][0], outs[1][0]), f"{kw}: chained loop differs"
            assert len(outs[0][1]) == len(outs[1][1]) and all(np.array_equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    finally:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))


def test_graphs_are_not_recaptured_for_new_tensors():
    """reference get_samples allocates a fresh x per call (sampler.py:98); dd_sample's graphs run on context-owned staging
    buffers, so a second call with other tensors of the same shape replays the captured graphs (and still writes its
    result into the caller's tensor)."""
    from duodiff_amd.engine import sample_loop
    m_s, _ = _uvit(dict(TINY, depth=1), 21, "bf16")
    m_f, _ = _uvit(dict(TINY, depth=3), 22, "bf16")
    B = 3
    es, ef = m_s.engine_model(B), m_f.engine_model(B)
    lib, h = es.ctx.lib, es.ctx.handle
    stream = torch.cuda.Stream()
    g = torch.Generator().manual_seed(9)
    xa, xb = torch.randn(B, 3, 8, 8, generator=g).cuda(), torch.randn(B, 3, 8, 8, generator=g).cuda()
    with torch.cuda.stream(stream):
        ra = xa.clone()
        sample_loop(es.ctx, es, ef, ra, t_switch=3, t_start=999, t_end=994, seed=5, noise="philox", use_graph=True, stream=stream)
        stream.synchronize()
        n1 = lib.dd_dev_graph_captures(h)
        rb = xb.clone()                                  # a different allocation
        sample_loop(es.ctx, es, ef, rb, t_switch=3, t_start=999, t_end=994, seed=5, noise="philox", use_graph=True, stream=stream)
        stream.synchronize()
        n2 = lib.dd_dev_graph_captures(h)
        eb = xb.clone()
        sample_loop(es.ctx, es, ef, eb, t_switch=3, t_start=999, t_end=994, seed=5, noise="philox", use_graph=False, stream=stream)
        stream.synchronize()
    assert n2 == n1, f"{n2 - n1} graph captures for a second call of the same shape"
    assert torch.equal(rb, eb) and not torch.equal(rb, xb) and torch.isfinite(rb).all()


def test_philox_noise_is_standard_normal():
    m, mp = _uvit(dict(TINY), 21, "bf16")
    B = 64
    em = m.engine_model(B)
    x0 = torch.randn(B, 3, 8, 8, generator=torch.Generator().manual_seed(1)).cuda()
    sig = oracle.schedule_oracle.step_coefficients(oracle.sampler_schedule(), 500)[2]
    zs = []
    for t, seed in ((500, 1), (500, 2), (499, 1)):
        a, b = x0.clone(), x0.clone()
        em.sample_step(a, t, noise="none")
        em.sample_step(b, t, noise="philox", seed=seed)
        s = oracle.schedule_oracle.step_coefficients(oracle.sampler_schedule(), t)[2]
        zs.append(((b - a) / float(s)).cpu().numpy().ravel())
    z = zs[0]
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    assert abs(np.mean(z ** 3)) < 0.1 and abs(np.mean(z ** 4) - 3.0) < 0.3
    assert abs(np.corrcoef(zs[0], zs[1])[0, 1]) < 0.05 and abs(np.corrcoef(zs[0], zs[2])[0, 1]) < 0.05
    assert sig > 0


def test_batch_independence_at_full_size():
    """Images are independent (no cross-sample op on the path): image i of a B=128 CelebA batch
    equals the same image run at B=2, bit for bit.  Also exercises the BASELINE batch size."""
    cfg = load_config(REPO / "configs" / "uvit_celeba_3.yaml")
    m, mp = _uvit(cfg, 1237, "bf16", max_batch=128)
    x = torch.randn(128, 3, 64, 64, generator=torch.Generator().manual_seed(8)).cuda()
    t = torch.full((128,), 640.0)
    eps_big = m(x, t)
    eps_small = m(x[:2].contiguous(), t[:2])
    assert torch.isfinite(eps_big).all()
    assert torch.equal(eps_big[:2], eps_small)
    eps_last = m(x[126:].contiguous(), t[:2])
    assert torch.equal(eps_big[126:], eps_last)


def test_full_size_bf16_close_to_fp32_at_batch_128():
    cfg = load_config(REPO / "configs" / "uvit_celeba.yaml")
    mb, _ = _uvit(cfg, 1236, "bf16", max_batch=128)
    x = torch.randn(128, 3, 64, 64, generator=torch.Generator().manual_seed(9)).cuda()
    t = torch.full((128,), 123.0)
    eb = mb(x, t)
    del mb
    mf, _ = _uvit(cfg, 1236, "fp32", max_batch=128)
    ef = mf(x, t)
    err = (eb - ef).abs().max().item()
    rms, sd = ((eb - ef) ** 2).mean().sqrt().item(), ef.std().item()
    print(f"celeba B=128: max|eps_bf16 - eps_fp32| = {err:.3e}, rms {rms:.3e} (std {sd:.3f})")
    assert err <= 4e-2 * sd and rms <= 8e-3 * sd      # the maximum over 1.6 M values (the 3e-2 * std bound is stated on 1.5 k-value slices)


# observed on MI355X (profiles/r03/parity_numbers.txt) x 1.2: free-running drift of the bf16 product against the fp32 engine,
# normalised by rms(x): K -> (max, rms)
# (observed: 2.31e-4 / 4.41e-5, 1.45e-3 / 2.74e-4, 4.90e-3 / 9.83e-4)
DRIFT_BOUND = {1: (2.9e-4, 5.5e-5), 10: (1.8e-3, 3.4e-4), 100: (6.1e-3, 1.2e-3)}


@pytest.mark.parametrize("B,qkv_bias", [(8, False), (3, True)])
def test_class_conditional_width_512_vs_oracle(B, qkv_bias):
    """No shipped config is class-conditional at embed_dim 512, so nothing full-size puts TWO extra tokens (label + time,
    L = 258) through the launches the headline path is made of: the attention launch that computes attn.qkv itself (extra
    tokens as keys, as the split query chunk, their norm1 + qkv inside), the fused block tail with 2 B extra-token rows
    (hidden-split tiles, reduce, column-split skip_linear rows), the fragment-order LayerNorm of the first block.  A synthetic
    5-block model of that shape: BOTH engines against the numpy oracle (reference models/uvit.py:351-383 restated; pinned to the
    reference by tests/test_oracle_golden.py) -- the fp32 engine runs gemm256 + attention_kernel at this shape, which no
    fixture covers either.  B = 8 takes the XCD-grouped workgroup map, B = 3 the plain one; with and without qkv bias."""
    cfg = dict(img_size=32, patch_size=2, in_chans=3, embed_dim=512, depth=5, num_heads=8, mlp_ratio=4, qkv_bias=qkv_bias,
               mlp_time_embed=False, num_classes=10, normalize_timesteps=True)
    g = torch.Generator().manual_seed(77 + B)
    x = torch.randn(B, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t = torch.full((B,), 417.0)
    want = _oracle(cfg, 4242)(x.numpy(), t.numpy(), y.numpy())
    sigma = float(want.std())
    for prec in ("fp32", "bf16"):
        m, _ = _uvit(cfg, 4242, prec, max_batch=B)
        got = m(x, t, y).cpu().numpy()
        del m
        assert np.isfinite(got).all()
        err, rms = float(np.abs(got - want).max()), float(np.sqrt(((got - want).astype(np.float64) ** 2).mean()))
        print(f"class-conditional D=512 L=258 B={B} qkv_bias={qkv_bias} {prec}: vs oracle max {err:.3e} rms {rms:.3e} (sigma {sigma:.3f})")
        if prec == "fp32":
            assert err <= 1e-4
        else:
            # The maximum is over the whole output (25 k / 9 k values, not a 1.5 k-value slice), and this synthetic model's output is
            # small (sigma 0.28): its ABSOLUTE rms error, 3.0e-3, equals that of its unconditional twin (3.1e-3 at sigma 0.42 = 7e-3
            # sigma).  Observed (round 3, against the fp32 engine): max 4.1e-2 sigma, rms 1.06e-2 sigma; gate = rms x 1.25, max = loose guard
            assert rms <= 1.25 * 1.06e-2 * sigma and err <= 6e-2 * sigma


@pytest.mark.parametrize("K", [1, 10, 100])
def test_benchmarked_path_drift_bf16_vs_fp32(K):
    """The thing bench.py times -- dd_sample, one hipGraph per backbone replayed, CelebA shallow + full pair, B = 128, device
    Philox noise, the 30 / 70 shallow / full mix -- free-running for K steps in bf16 against the SAME loop on the exact-fp32
    engine (same x_T, same Philox seed => identical noise): the north_star's "1e-3 max-abs" question asked of the product
    path itself.  (The fp32 engine is pinned to the oracle / the reference's own rollouts by the tests above.)"""
    from duodiff_amd.engine import sample_loop
    B = 128
    cfg_s, cfg_f = load_config(REPO / "configs" / "uvit_celeba_3.yaml"), load_config(REPO / "configs" / "uvit_celeba.yaml")
    x_T = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(77)).cuda().contiguous()
    k_switch = max(1, round(0.3 * K))
    out = {}
    for prec in ("bf16", "fp32"):
        ms, _ = _uvit(cfg_s, 1237, prec, max_batch=B)
        mf, _ = _uvit(cfg_f, 1236, prec, max_batch=B)
        es, ef = ms.engine_model(B), mf.engine_model(B)
        x = x_T.clone()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            sample_loop(es.ctx, es, ef, x, t_switch=k_switch, t_start=999, t_end=1000 - K, seed=5, noise="philox",
                        use_graph=True, stream=stream)
        stream.synchronize()
        out[prec] = x.cpu().numpy().astype(np.float64)
        del ms, mf, es, ef
    assert np.isfinite(out["bf16"]).all() and np.isfinite(out["fp32"]).all()
    scale = float(np.sqrt((out["fp32"] ** 2).mean()))
    d = np.abs(out["bf16"] - out["fp32"]) / scale
    dmax, drms = float(d.max()), float(np.sqrt((d ** 2).mean()))
    print(f"benchmarked path K={K} ({k_switch} shallow + {K - k_switch} full steps, B={B}): bf16 vs fp32 engine max {dmax:.3e} rms {drms:.3e} (rms x {scale:.3f})")
    assert dmax <= DRIFT_BOUND[K][0] and drms <= DRIFT_BOUND[K][1]


def test_error_behaviour_matches_reference_classes():
    from duodiff_amd.uvit import UViT
    cfg = dict(TINY, num_classes=10)
    m, mp = _uvit(cfg, 5, "fp32")
    x = torch.zeros(2, 3, 8, 8)
    with pytest.raises(RuntimeError):          # quirk Q5: conditional model without y
        m(x, torch.zeros(2))
    with pytest.raises(IndexError):            # quirk Q4: label out of range for nn.Embedding
        m(x, torch.zeros(2), torch.tensor([1, 10]))
    with pytest.raises(RuntimeError):          # wrong image shape
        m(torch.zeros(2, 3, 4, 4), torch.zeros(2), torch.tensor([1, 2]))
    u = UViT(**ModelParams.from_dict(dict(TINY)).as_dict())
    with pytest.raises(RuntimeError):
        u(torch.zeros(1, 3, 8, 8), torch.zeros(1))     # no weights loaded


def test_ddim_and_other_parametrizations_vs_reference(golden):
    """SURVEY section 8f next-2 on the GPU (fp32 mode): DDIM rollouts incl. late-model switch against the reference's
    own outputs, and the predict_original / predict_previous updates against the reference."""
    from duodiff_amd import sampler
    fd = golden("ddim_tiny.npz")
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "fp32")
    m_f, _ = _uvit(dict(TINY, depth=3), 301, "fp32")
    for tag in ("a", "b", "c", "nan"):
        steps, eta, tsw = fd[f"cfg_{tag}"]
        with np.errstate(invalid="ignore"):
            samples, inter = sampler.get_samples(m_s, 2, sampler.predict_noise_postprocessing, 3, 3, 8, 8, use_ddim=True,
                                                 ddim_steps=int(steps), ddim_eta=float(eta), timesteps_save=[1],
                                                 late_model=m_f, t_switch=int(tsw), noise="torch_cpu")
        np.testing.assert_allclose(inter[0], fd[f"first_{tag}"], rtol=0, atol=1e-5, equal_nan=False)
        if tag == "nan":   # the reference's own output for (10 steps, eta 0.5) is all-NaN (sigma^2-for-sigma quirk, sampler.py:112-120)
            assert float(fd["nan_fraction"]) == 1.0 and np.isnan(samples).all()
            continue
        assert np.isfinite(samples).all()
        scale = max(1.0, float(np.abs(fd[f"samples_{tag}"]).max()))
        # case b: 499 free-running steps with eta = 0.5 and the late-model switch; fp32 rounding differences are amplified
        np.testing.assert_allclose(samples, fd[f"samples_{tag}"], rtol=0, atol=(2e-3 if tag == "b" else 2e-4) * scale, equal_nan=False)
    fx = golden("param_steps.npz")
    x, m = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["m"]).cuda()
    for t in fx["ts"]:
        t = int(t)
        z = torch.from_numpy(fx[f"z_{t}"]).cuda()
        got = sampler.predict_original_postprocessing(m, x, t, z).cpu().numpy()
        ref = fx[f"orig_{t}"]
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 * max(1.0, float(np.abs(ref).max())))
        got = sampler.predict_previous_postprocessing(m, x, t, z).cpu().numpy()
        np.testing.assert_allclose(got, fx[f"prev_{t}"], rtol=0, atol=1e-6)
    # a short predict_previous loop runs end to end through get_samples
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "bf16")
    s, _ = sampler.get_samples(m_s, 2, sampler.predict_previous_postprocessing, 0, 3, 8, 8, num_steps=5, noise="torch_cpu")
    assert s.shape == (2, 8, 8, 3) and np.isfinite(s).all()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_device_resident_ddim_and_parametrization_loops(precision):
    """dd_sample_affine (DDIM, predict_original / predict_previous as device-resident loops; reference sampler.py:103-126,
    59-79, 128-139): (1) the hipGraph replay equals the eager launches of the same loop bit for bit, with device noise and a
    backbone switch; (2) with a noise term that is exactly zero (DDIM eta = 0: the reference still adds sigma^2 z = 0 z) the
    loop equals the step-by-step Python loop (dd_forward + dd_affine_step, the path the reference fixtures pin) bit for bit,
    save points included; (3) the noisy DDIM / predict_* runs are finite and differ from the noise-free ones."""
    from duodiff_amd import sampler
    m_s, _ = _uvit(dict(TINY, depth=1), 300, precision)
    m_f, _ = _uvit(dict(TINY, depth=3), 301, precision)
    run = lambda **kw: sampler.get_samples(m_s, 3, kw.pop("post", sampler.predict_noise_postprocessing), 3, 3, 8, 8,
                                           late_model=m_f, **kw)
    # (2) DDIM eta = 0, switch inside the run, one save point: device loop == Python loop
    for tsw in (300, 700):
        want, winter = run(use_ddim=True, ddim_steps=20, ddim_eta=0.0, timesteps_save=[1, 500], t_switch=tsw, noise="torch_cpu")
        for graph in (True, False):
            got, ginter = run(use_ddim=True, ddim_steps=20, ddim_eta=0.0, timesteps_save=[1, 500], t_switch=tsw, noise="device",
                              use_graph=graph)
            assert np.array_equal(got, want), f"DDIM eta=0 t_switch={tsw} graph={graph}: device loop differs from the Python loop"
            assert len(ginter) == len(winter) and all(np.array_equal(a, b) for a, b in zip(ginter, winter))
    # (1) + (3) noisy loops: graph == eager, finite, noise matters
    for kw in (dict(use_ddim=True, ddim_steps=50, ddim_eta=0.01, t_switch=300),
               dict(post=sampler.predict_original_postprocessing, num_steps=60, t_switch=970),
               dict(post=sampler.predict_previous_postprocessing, num_steps=40, t_switch=980, timesteps_save=[3, 25])):
        g, gi = run(noise="device", use_graph=True, **dict(kw))
        e, ei = run(noise="device", use_graph=False, **dict(kw))
        assert np.isfinite(g).all() and np.array_equal(g, e) and all(np.array_equal(a, b) for a, b in zip(gi, ei))
        assert len(gi) == len(kw.get("timesteps_save", []))
    g0, _ = run(use_ddim=True, ddim_steps=50, ddim_eta=0.0, t_switch=300, noise="device")
    g1, _ = run(use_ddim=True, ddim_steps=50, ddim_eta=0.01, t_switch=300, noise="device")
    assert not np.array_equal(g0, g1)


def test_save_points_do_not_change_the_final_samples():
    """Cutting a device-resident loop at intermediate save points (timesteps_save) must not change the final samples: every
    segment keeps the seed and continues the Philox counter at its first step's index (dd_affine_sample_args.counter_base),
    so step k draws the same z whether or not the loop was cut before it.  Noisy DDIM (eta > 0), predict_original and
    predict_previous with a backbone switch; and the DDPM loop (dd_sample: counter = t) for completeness."""
    from duodiff_amd import sampler
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "fp32")
    m_f, _ = _uvit(dict(TINY, depth=3), 301, "fp32")
    run = lambda **kw: sampler.get_samples(m_s, 3, kw.pop("post", sampler.predict_noise_postprocessing), 7, 3, 8, 8,
                                           late_model=m_f, noise="device", **kw)
    # (DDIM visits t = linspace(0, 999, 40).astype(int) reversed: 1000 - t takes the values 1, 27, ..., 258, ..., 642, ...)
    for kw, saves in ((dict(use_ddim=True, ddim_steps=40, ddim_eta=0.02, t_switch=400), [1, 258, 642]),
                      (dict(post=sampler.predict_original_postprocessing, num_steps=50, t_switch=975), [2, 30]),
                      (dict(post=sampler.predict_previous_postprocessing, num_steps=40, t_switch=980), [5, 6, 39]),
                      (dict(num_steps=60, t_switch=30), [10, 45])):
        plain, _ = run(**dict(kw))
        cut, inter = run(timesteps_save=saves, **dict(kw))
        assert np.isfinite(plain).all() and len(inter) >= 2
        assert np.array_equal(plain, cut), f"{kw}: final samples depend on the save points"
        assert not np.array_equal(inter[0], inter[-1])
    # and the noise is really there: another seed gives other samples
    a, _ = run(use_ddim=True, ddim_steps=40, ddim_eta=0.02, t_switch=400)
    b, _ = sampler.get_samples(m_s, 3, sampler.predict_noise_postprocessing, 8, 3, 8, 8, late_model=m_f, noise="device",
                               use_ddim=True, ddim_steps=40, ddim_eta=0.02, t_switch=400)
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("num_classes", [-1, 10])
@pytest.mark.parametrize("D,H", [(128, 2), (256, 4)])
def test_fused_branches_at_small_widths_vs_oracle(D, H, num_classes):
    """embed_dim 128 / 256 take launches no shipped config reaches through a model: the fused block tail with the NEXT block's
    attn.qkv inside it (QKV instantiation, ragged tiles: qkv_dump), launch_qkv_rows on the extra-token rows, the bf16
    head-major qkv GEMM of the first block, attention_kernel on that tensor.  7-block models (3 in / mid / 3 out: skip_linear
    phases too), 64 patches + 1 or 2 extra tokens, B = 5 (ragged 128-row tile), both precisions against the numpy oracle."""
    cfg = dict(img_size=16, patch_size=2, in_chans=3, embed_dim=D, depth=7, num_heads=H, mlp_ratio=4, qkv_bias=False,
               mlp_time_embed=False, num_classes=num_classes, normalize_timesteps=True)
    B = 5
    g = torch.Generator().manual_seed(D + B)
    x = torch.randn(B, 3, 16, 16, generator=g)
    y = torch.randint(0, 10, (B,), generator=g) if num_classes > 0 else None
    t = torch.full((B,), 250.0)
    want = _oracle(cfg, 99)(x.numpy(), t.numpy(), y.numpy() if y is not None else None)
    sigma = float(want.std())
    for prec in ("fp32", "bf16"):
        m, _ = _uvit(cfg, 99, prec, max_batch=B)
        got = m(x, t, y).cpu().numpy()
        del m
        err, rms = float(np.abs(got - want).max()), float(np.sqrt(((got - want).astype(np.float64) ** 2).mean()))
        print(f"D={D} classes={num_classes} {prec}: vs oracle max {err:.3e} rms {rms:.3e} (sigma {sigma:.3f})")
        assert np.isfinite(got).all()
        assert err <= (1e-4 if prec == "fp32" else 6e-2 * sigma) and (prec == "fp32" or rms <= 1.5e-2 * sigma)


@pytest.mark.parametrize("flags", [32, 64, 128, 32 | 128, 64 | 128, 1, 2, 4])
def test_kernel_variant_flags_agree_with_the_default_path(flags):
    """Every dd_dev_set_flags setting selects another launch sequence for the same arithmetic (run_backbone's h_ready /
    skip_done / qkv_done / qa_ready state machine): on a 5-block class-conditional model at embed_dim 512 each variant must
    stay within bf16 rounding of the oracle, like the default path (32 = no fused skip_linear, 64 = no qkv in the block tail,
    128 = no qkv inside the attention launch -> the QKV instantiation / the qkv GEMM, and their combinations)."""
    from duodiff_amd.engine import Context
    cfg = dict(img_size=32, patch_size=2, in_chans=3, embed_dim=512, depth=5, num_heads=8, mlp_ratio=4, qkv_bias=False,
               mlp_time_embed=False, num_classes=10, normalize_timesteps=True)
    B = 3
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t = torch.full((B,), 600.0)
    want = _oracle(cfg, 4243)(x.numpy(), t.numpy(), y.numpy())
    sigma = float(want.std())
    ctx = Context.get()
    try:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
        m, _ = _uvit(cfg, 4243, "bf16", max_batch=B)
        got = m(x, t, y).cpu().numpy()
        del m
    finally:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
    err, rms = float(np.abs(got - want).max()), float(np.sqrt(((got - want).astype(np.float64) ** 2).mean()))
    print(f"dev_flags {flags}: vs oracle max {err:.3e} rms {rms:.3e} (sigma {sigma:.3f})")
    assert np.isfinite(got).all() and err <= 6e-2 * sigma and rms <= 1.4e-2 * sigma


@pytest.mark.parametrize("img,B,flags,ncls", [(32, 3, 0, 10), (32, 3, 128, 10), (32, 3, 1024, 10), (32, 3, 2048, 10), (32, 3, 16384, 10), (16, 5, 0, 10), (16, 5, 1024, 10),
                                               (32, 4, 0, -1), (16, 5, 0, -1)])
def test_row_resident_fc2_at_embed_dim_768(img, B, flags, ncls):
    """embed_dim 768 (the ImageNet-64 width) has no fused block tail; its mlp.fc2 + residual + the next block's norm1 run as one
    row-resident launch (rowlin.hip), the extra-token rows K-split into slabs.  5-block class-conditional models: 256 patches
    (norm1 leaves in the attention launch's fragment order; flags 128: row-major, the qkv GEMM reads it) and 64 patches at B = 5
    (a ragged 128-row tile), class-conditional (two extra tokens per image) and unconditional (one), each against the oracle like the GEMM + LayerNorm pairs it replaces (flags 1024; 2048: attn.proj + norm2 alone stay a GEMM pair; 16384: the out-blocks' skip_linear + norm1 do)."""
    from duodiff_amd.engine import Context
    cfg = dict(img_size=img, patch_size=2, in_chans=3, embed_dim=768, depth=5, num_heads=12, mlp_ratio=4, qkv_bias=False,
               mlp_time_embed=False, num_classes=ncls, normalize_timesteps=True)
    g = torch.Generator().manual_seed(77 + img)
    x = torch.randn(B, 3, img, img, generator=g)
    y = torch.randint(0, 10, (B,), generator=g) if ncls > 0 else None
    t = torch.full((B,), 420.0)
    want = _oracle(cfg, 4244)(x.numpy(), t.numpy(), y.numpy() if y is not None else None)
    sigma = float(want.std())
    ctx = Context.get()
    try:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
        m, _ = _uvit(cfg, 4244, "bf16", max_batch=B)
        got = m(x, t, y).cpu().numpy()
        alone = m(x[B - 1:].contiguous(), t[:1], y[B - 1:].contiguous() if y is not None else None).cpu().numpy()
        del m
    finally:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
    err, rms = float(np.abs(got - want).max()), float(np.sqrt(((got - want).astype(np.float64) ** 2).mean()))
    print(f"D=768 img {img} B={B} classes {ncls} dev_flags {flags}: vs oracle max {err:.3e} rms {rms:.3e} (sigma {sigma:.3f})")
    assert np.isfinite(got).all() and err <= 6e-2 * sigma and rms <= 1.4e-2 * sigma
    assert np.array_equal(alone, got[B - 1:])        # the same image computed alone: bit-identical (no batch-dependent split)


@pytest.mark.parametrize("img,B,flags,ncls", [(32, 3, 0, 10), (32, 3, 128, 10), (32, 3, 8192, 10), (16, 5, 0, 10), (32, 4, 0, -1)])
def test_split_k_linears_at_embed_dim_1024(img, B, flags, ncls):
    """embed_dim 1024 at small batches (the ImageNet-256 latent models): skip_linear, attn.proj and mlp.fc2 run as split-K halves of the
    256 x 256 kernel + one row pass (slabs + bias + residual + the LayerNorm behind the Linear).  5-block class-conditional models
    against the oracle, like the whole-K GEMM + LayerNorm launches they replace (flags 8192); flags 128: norm1 row-major for the
    qkv GEMM; 64 patches: sequence lengths the 256 x 256 row partition does not take keep the whole-K launches.  The same image
    computed alone is bit-identical."""
    from duodiff_amd.engine import Context
    cfg = dict(img_size=img, patch_size=2, in_chans=4, embed_dim=1024, depth=5, num_heads=16, mlp_ratio=4, qkv_bias=False,
               mlp_time_embed=False, num_classes=ncls, normalize_timesteps=True)
    g = torch.Generator().manual_seed(78 + img)
    x = torch.randn(B, 4, img, img, generator=g)
    y = torch.randint(0, 10, (B,), generator=g) if ncls > 0 else None
    t = torch.full((B,), 130.0)
    want = _oracle(cfg, 4245)(x.numpy(), t.numpy(), y.numpy() if y is not None else None)
    sigma = float(want.std())
    ctx = Context.get()
    try:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, flags))
        m, _ = _uvit(cfg, 4245, "bf16", max_batch=B)
        got = m(x, t, y).cpu().numpy()
        alone = m(x[B - 1:].contiguous(), t[:1], y[B - 1:].contiguous() if y is not None else None).cpu().numpy()
        del m
    finally:
        ctx.check(ctx.lib.dd_dev_set_flags(ctx.handle, 0))
    err, rms = float(np.abs(got - want).max()), float(np.sqrt(((got - want).astype(np.float64) ** 2).mean()))
    print(f"D=1024 img {img} B={B} classes {ncls} dev_flags {flags}: vs oracle max {err:.3e} rms {rms:.3e} (sigma {sigma:.3f})")
    assert np.isfinite(got).all() and err <= 6e-2 * sigma and rms <= 1.4e-2 * sigma
    assert np.array_equal(alone, got[B - 1:])


def test_cli_end_to_end(tmp_path):
    """The sampler CLI with the reference's flags: YAML configs + checkpoint files (bare state_dict and the
    trainer's {"model_state_dict": ...} format) -> statistics.txt and samples, DuoDiff switch included."""
    import subprocess, sys, yaml
    cfg_s, cfg_f = dict(TINY, depth=1, img_size=16), dict(TINY, depth=3, img_size=16)
    for name, cfg, seed, wrap in (("s", cfg_s, 1, False), ("f", cfg_f, 2, True)):
        (tmp_path / f"{name}.yaml").write_text(yaml.safe_dump({"model_params": dict(cfg, classifier_type="x")}))
        sd = synthetic_state_dict(ModelParams.from_dict(cfg), seed)
        torch.save({"model_state_dict": sd, "step": 3} if wrap else dict(sd), tmp_path / f"{name}.pth")
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "duodiff_amd.sampler", "--seed", "4", "--checkpoint_path", str(tmp_path / "s.pth"),
           "--checkpoint_path_late", str(tmp_path / "f.pth"), "--config_path", str(tmp_path / "s.yaml"),
           "--config_path_late", str(tmp_path / "f.yaml"), "--t_switch", "300", "--batch_size", "3",
           "--parametrization", "predict_noise", "--output_folder", str(out), "--no_png", "--precision", "fp32",
           "--noise", "torch_cpu"]
    r = subprocess.run(cmd, cwd=str(REPO), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (out / "statistics.txt").read_text().startswith("Elapsed time: ")
    got = np.load(out / "samples.npy")
    assert got.shape == (3, 16, 16, 3) and got.dtype == np.float32
    # same run through the oracle: same seeds, same torch CPU noise stream, fp32 -> same images up to drift
    o_s = _oracle(cfg_s, 1); o_f = _oracle(cfg_f, 2)
    want, _ = oracle.get_samples(o_s, 3, 4, 3, 16, 16, late_model=o_f, t_switch=300)
    assert o_s.calls == 300 and o_f.calls == 700
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-3 * scale)


@pytest.mark.parametrize("B", [1, 3, 5, 17, 40])
def test_ragged_batch_sizes_row_partition(B):
    """Odd batch sizes drive the GEMM row partition through its edge cases (q = B main tiles, one tail row each;
    B = 1 is a single 256-row tile + 1 tail row).  bf16 engine vs the oracle on image 0 and B-1, and every image
    bit-identical to the same image computed alone (B = 1)."""
    cfg = load_config(REPO / "configs" / "uvit_celeba_3.yaml")
    m, mp = _uvit(cfg, 1237, "bf16", max_batch=64)
    orc = _oracle(cfg, 1237)
    x = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(100 + B))
    t = torch.full((B,), 321.0)
    eps = m(x, t).cpu().numpy()
    assert np.isfinite(eps).all()
    for i in sorted({0, B - 1}):
        want = orc(x[i:i + 1].numpy(), np.full((1,), 321.0, np.float32))
        assert np.abs(eps[i:i + 1] - want).max() <= eps_tol("bf16", want) * 4 / 3     # (whole image: 12 k values)
        alone = m(x[i:i + 1].contiguous(), t[:1]).cpu().numpy()
        assert np.array_equal(alone, eps[i:i + 1])


def test_conditional_model_full_size_labels():
    """Class-conditional path at full size (ImageNet-64 shallow, L = 258: two tail rows per tile) with distinct labels."""
    cfg = load_config(REPO / "configs" / "uvit_imagenet64_3.yaml")
    m, mp = _uvit(cfg, 1239, "bf16", max_batch=8)
    orc = _oracle(cfg, 1239)
    B = 6
    x = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    y = torch.tensor([0, 1, 500, 998, 999, 3])
    t = torch.full((B,), 12.0)
    eps = m(x, t, y).cpu().numpy()
    want = orc(x.numpy(), t.numpy(), y.numpy())
    err = np.abs(eps - want).max()
    print(f"imagenet64_3 cond B=6: max|eps - oracle| = {err:.3e} (std {want.std():.3f})")
    assert err <= eps_tol("bf16", want) * 4 / 3       # (six whole images)
    # a different label changes that image's output and only that image's
    y2 = y.clone(); y2[2] = 501
    eps2 = m(x, t, y2).cpu().numpy()
    assert np.array_equal(np.delete(eps, 2, 0), np.delete(eps2, 2, 0)) and not np.array_equal(eps[2], eps2[2])


def test_get_samples_device_noise_on_default_stream():
    """noise="device" replays hipGraphs; called from the default stream (as the CLI does) the loop moves to a side stream
    instead of failing in hipStreamBeginCapture.  Same seeds -> identical images; intermediate saves split the loop."""
    from duodiff_amd import sampler
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "fp32")
    m_f, _ = _uvit(dict(TINY, depth=3), 301, "fp32")
    kw = dict(late_model=m_f, t_switch=300, noise="device", num_steps=40)
    a, _ = sampler.get_samples(m_s, 3, sampler.predict_noise_postprocessing, 7, 3, 8, 8, **kw)
    b, inter = sampler.get_samples(m_s, 3, sampler.predict_noise_postprocessing, 7, 3, 8, 8, timesteps_save=[20], **kw)
    assert a.shape == (3, 8, 8, 3) and np.isfinite(a).all()
    assert np.array_equal(a, b) and len(inter) == 1 and inter[0].shape == (3, 8, 8, 3)


def test_sharded_cli_single_rank(tmp_path):
    """python -m duodiff_amd.dist with WORLD_SIZE unset = one rank: same flags as the sampler, rank 0 writes the output."""
    import subprocess, sys, yaml
    cfg = dict(TINY, depth=1, img_size=16)
    (tmp_path / "m.yaml").write_text(yaml.safe_dump({"model_params": dict(cfg)}))
    torch.save(dict(synthetic_state_dict(ModelParams.from_dict(cfg), 5)), tmp_path / "m.pth")
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "duodiff_amd.dist", "--seed", "3", "--checkpoint_path", str(tmp_path / "m.pth"), "--config_path",
           str(tmp_path / "m.yaml"), "--batch_size", "4", "--parametrization", "predict_noise", "--output_folder", str(out), "--no_png",
           "--precision", "fp32", "--noise", "device", "--use_ddim", "--ddim_steps", "5"]
    r = subprocess.run(cmd, cwd=str(REPO), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out / "samples.npy")
    assert got.shape == (4, 16, 16, 3) and np.isfinite(got).all()


def test_workspace_grows_with_batch():
    """No max_batch given: the engine model is rebuilt when a later call brings a larger batch; results do not depend on it."""
    cfg = dict(TINY)
    m, _ = _uvit(cfg, 77, "fp32")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(9, 3, 8, 8, generator=g)
    t = torch.full((9,), 123.0)
    e2 = m(x[:2].contiguous(), t[:2]).cpu()
    e9 = m(x, t).cpu()
    e2b = m(x[:2].contiguous(), t[:2]).cpu()
    assert torch.equal(e2, e9[:2]) and torch.equal(e2, e2b)
    # graph replay after growth: a sampling loop at the larger batch on the default stream
    from duodiff_amd import sampler
    s, _ = sampler.get_samples(m, 9, sampler.predict_noise_postprocessing, 1, 3, 8, 8, noise="device", num_steps=10)
    assert s.shape == (9, 8, 8, 3) and np.isfinite(s).all()


def test_noise_scheduler_sample_vs_reference(golden):
    """NoiseScheduler.sample (ddpm_core.py:106-214, uvit branch) on the HIP path against the reference's own runs:
    a 50-step schedule in both variance modes (forward + explicit-coefficient update, tables from dd_schedule_build)
    and the default 1000-step schedule with sigma^2 = beta through the FUSED sampling step (DD_VAR_BETA)."""
    from duodiff_amd.ddpm_core import NoiseScheduler
    fx = golden("scheduler_tiny.npz")
    m, _ = _uvit(dict(TINY, depth=1), int(fx["seed"]), "fp32")
    for tag, steps, mode in (("", 50, "beta"), ("_bt50", 50, "beta_tilde"), ("_b1000", 1000, "beta")):
        sch = NoiseScheduler(beta_steps=steps, variance_mode=mode)
        x0, log = sch.sample(m, num_steps=steps, data_shape=(3, 8, 8), num_samples=2, seed=5, model_type="uvit")
        over = [v.cpu().numpy() for v in log["samples_over_time"]]
        assert len(over) == steps
        np.testing.assert_allclose(over[0], fx["x_after_first" + tag], rtol=0, atol=1e-5, equal_nan=False)
        for got, key in ((over[steps // 2], "x_mid" + tag), (x0.cpu().numpy(), "x0" + tag)):
            np.testing.assert_allclose(got, fx[key], rtol=0, atol=5e-4 * max(1.0, float(np.abs(fx[key]).max())), equal_nan=False)
    # the fused step and the explicit-coefficient path are the same arithmetic: bit-identical trajectories
    sch = NoiseScheduler()
    a, _ = sch.sample(m, 1000, (3, 8, 8), 2, seed=5, fused=True, keep_samples_over_time=False)
    b, _ = sch.sample(m, 1000, (3, 8, 8), 2, seed=5, fused=False, keep_samples_over_time=False)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        NoiseScheduler(variance_mode="beta_tilde").sample(m, 10, (3, 8, 8), 2, seed=5, fused=True)
    with pytest.raises(IndexError):
        NoiseScheduler(beta_steps=50).sample(m, 51, (3, 8, 8), 2, seed=5)


@pytest.mark.parametrize("t_switch", [0, 1, 2, 300, 1000, 1200, -5, np.inf])
def test_backbone_per_step_matches_reference_rule(t_switch):
    """Which backbone runs each step: the reference switches AFTER the step at t == 1000 - t_switch (sampler.py:135-136),
    so a t_switch outside [1, 1000] never switches.  Both noise paths of get_samples against manual stepping."""
    from duodiff_amd import sampler
    m_s, _ = _uvit(dict(TINY, depth=1), 300, "fp32")
    m_f, _ = _uvit(dict(TINY, depth=3), 301, "fp32")
    n, B, seed = 4, 2, 9
    sw = 1000 - t_switch if np.isfinite(t_switch) else None
    pick = lambda t: (m_s if (sw is None or not (0 <= sw <= 999) or t >= sw) else m_f).engine_model(B)
    sampler.seed_everything(seed)
    x_T = torch.randn(B, 3, 8, 8).cuda().contiguous()
    # device noise: philox seeded with `seed`
    xm = x_T.clone()
    for t in range(999, 999 - n, -1):
        pick(t).sample_step(xm, t, noise="philox", seed=seed)
    got, _ = sampler.get_samples(m_s, B, sampler.predict_noise_postprocessing, seed, 3, 8, 8, late_model=m_f,
                                 t_switch=t_switch, noise="device", num_steps=n, return_device_tensor=True)
    assert torch.equal(got, m_s.engine_model(B).ctx.to_images(xm))
    # host noise: the torch CPU stream after seed_everything
    sampler.seed_everything(seed)
    xh = torch.randn(B, 3, 8, 8).cuda().contiguous()
    for t in range(999, 999 - n, -1):
        pick(t).sample_step(xh, t, z=torch.randn(xh.shape).cuda(), noise="buffer")
    got, _ = sampler.get_samples(m_s, B, sampler.predict_noise_postprocessing, seed, 3, 8, 8, late_model=m_f,
                                 t_switch=t_switch, noise="torch_cpu", num_steps=n, return_device_tensor=True)
    assert torch.equal(got, m_s.engine_model(B).ctx.to_images(xh))


def test_to_images_matches_reference_convention():
    """dd_to_images == rearrange((x + 1) / 2, "b c h w -> b h w c") (sampler.py:145-146), bit for bit, C = 3 and 4."""
    from duodiff_amd.engine import Context
    ctx = Context.get()
    for shape in ((5, 3, 64, 64), (2, 4, 32, 32), (1, 3, 8, 8)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 3
        want = ((x + 1) / 2).permute(0, 2, 3, 1).contiguous()
        assert torch.equal(ctx.to_images(x.cuda().contiguous()).cpu(), want)


@pytest.mark.parametrize("name,B", [("uvit_imagenet64", 256), ("uvit_imagenet256", 32), ("uvit_celeba", 128)])
def test_full_batch_vs_oracle(name, B):
    """The BASELINE batch sizes (row partitions M = 66 048 x D 768, M = 8 256 x D 1024, M = 32 896 x D 512) against the
    oracle on image 0 and the last image, bf16 engine; every other image must at least be finite."""
    cfg = load_config(REPO / "configs" / f"{name}.yaml")
    seed = 1234 + FULL_NAMES.index(name)
    m, mp = _uvit(cfg, seed, "bf16", max_batch=B)
    orc = _oracle(cfg, seed)
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(B, mp.in_chans, mp.img_size, mp.img_size, generator=g)
    y = torch.randint(0, mp.num_classes, (B,), generator=g) if mp.num_classes > 0 else None
    t = torch.full((B,), 250.0)
    eps = m(x, t, y).cpu().numpy()
    assert np.isfinite(eps).all()
    for i in (0, B - 1):
        want = orc(x[i:i + 1].numpy(), np.full((1,), 250.0, np.float32), y[i:i + 1].numpy() if y is not None else None)
        err = np.abs(eps[i:i + 1] - want).max()
        print(f"{name} B={B} image {i}: max|eps - oracle| = {err:.3e} (std {want.std():.3f})")
        assert err <= eps_tol("bf16", want) * 4 / 3   # (a whole image)


def test_two_ranks_equal_two_single_rank_runs(tmp_path):
    """Engine-level multi-GPU contract on one GPU: two fresh processes run `python -m duodiff_amd.dist` as ranks 0 / 1 of a
    world of 2 (transport forced to gloo, both on this GPU; the driver runs the RCCL path on the 8-GPU node).  The
    gathered images must equal, bit for bit, the two single-rank runs with seeds base and base + 1, in rank order."""
    import os, socket, subprocess, sys, yaml
    from duodiff_amd import sampler
    cfg_s, cfg_f = dict(TINY, depth=1, img_size=16), dict(TINY, depth=3, img_size=16)
    for name, cfg, seed in (("s", cfg_s, 21), ("f", cfg_f, 22)):
        (tmp_path / f"{name}.yaml").write_text(yaml.safe_dump({"model_params": dict(cfg)}))
        torch.save(dict(synthetic_state_dict(ModelParams.from_dict(cfg), seed)), tmp_path / f"{name}.pth")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "out"
    base, B = 11, 3
    cmd = [sys.executable, "-m", "duodiff_amd.dist", "--seed", str(base), "--checkpoint_path", str(tmp_path / "s.pth"),
           "--checkpoint_path_late", str(tmp_path / "f.pth"), "--config_path", str(tmp_path / "s.yaml"),
           "--config_path_late", str(tmp_path / "f.yaml"), "--t_switch", "300", "--batch_size", str(B),
           "--parametrization", "predict_noise", "--output_folder", str(out), "--no_png", "--precision", "bf16", "--noise", "device"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DUODIFF_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(cmd, cwd=str(REPO), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
    got = np.load(out / "samples.npy")
    assert got.shape == (2 * B, 16, 16, 3) and np.isfinite(got).all()
    m_s, _ = _uvit(cfg_s, 21, "bf16")
    m_f, _ = _uvit(cfg_f, 22, "bf16")
    for r in range(2):
        want, _ = sampler.get_samples(m_s, B, sampler.predict_noise_postprocessing, base + r, 3, 16, 16, late_model=m_f,
                                      t_switch=300, noise="device")
        assert np.array_equal(got[r * B:(r + 1) * B], want), f"rank {r} shard differs from the single-rank run with seed {base + r}"
