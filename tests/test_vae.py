"""KL-VAE decode: oracle vs the reference-generated fixture on CPU; the HIP decoder
(dd_vae_decode through the C ABI) vs the same fixture on the GPU."""
import numpy as np
import pytest
import torch

from duodiff_amd.autoencoder import FrozenAutoencoderKL, synthetic_vae_state_dict, vae_param_shapes
from oracle import vae_decode

from conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN / "vae_decode.npz")


@pytest.fixture(scope="module")
def sd(gold):
    return synthetic_vae_state_dict(int(gold["seed"]))


def test_param_inventory():
    shp = vae_param_shapes()
    assert len(shp) == 140 and sum(int(np.prod(s)) for s in shp.values()) == 49_490_199
    assert shp["decoder.up.1.block.0.nin_shortcut.weight"] == (256, 512, 1, 1)
    assert shp["decoder.up.0.block.0.nin_shortcut.weight"] == (128, 256, 1, 1)
    assert "decoder.up.0.upsample.conv.weight" not in shp and shp["decoder.conv_out.weight"] == (3, 128, 3, 3)


def test_oracle_matches_reference_small(gold, sd):
    y = vae_decode(gold["z8"], sd)
    assert y.shape == (2, 3, 64, 64)
    np.testing.assert_allclose(y, gold["y8"], rtol=0, atol=2e-5)


def test_oracle_matches_reference_full(gold, sd):
    y = vae_decode(gold["z32"], sd)
    np.testing.assert_allclose(y[:, :, 96:160, 96:160], gold["y32_slice"], rtol=0, atol=5e-5)
    assert abs(float(y.astype(np.float64).sum()) - float(gold["y32_checksum"])) < 1.0


def test_state_dict_errors(sd):
    bad = dict(sd)
    bad.pop("decoder.conv_in.weight")
    with pytest.raises(RuntimeError, match="missing keys"):
        FrozenAutoencoderKL(bad)
    bad = dict(sd)
    bad["decoder.conv_in.weight"] = torch.zeros(512, 4, 1, 1)
    with pytest.raises(RuntimeError, match="size mismatch"):
        FrozenAutoencoderKL(bad)
    with pytest.raises(NotImplementedError):
        FrozenAutoencoderKL(sd)(torch.zeros(1, 3, 8, 8), fn="encode")


# ---------------------------------------------------------------- GPU
def _err(a, b):
    return float(np.abs(a - b).max())


@pytest.mark.gpu
def test_vae_decode_fp32_small(gold, sd):
    ae = FrozenAutoencoderKL(sd, precision="fp32", max_chunk=2, max_latent=8).to("cuda:0")
    y = ae.decode(torch.from_numpy(gold["z8"])).cpu().numpy()
    assert y.shape == (2, 3, 64, 64)
    assert _err(y, gold["y8"]) <= 2e-4, _err(y, gold["y8"])


@pytest.mark.gpu
def test_vae_decode_fp32_full(gold, sd):
    ae = FrozenAutoencoderKL(sd, precision="fp32", max_chunk=1, max_latent=32).to("cuda:0")
    y = ae.decode(torch.from_numpy(gold["z32"])).cpu().numpy()
    assert y.shape == (1, 3, 256, 256)
    e = _err(y[:, :, 96:160, 96:160], gold["y32_slice"])
    assert e <= 3e-4, e
    st = gold["y32_stats"]
    assert abs(y.mean(dtype=np.float64) - st[0]) < 1e-5 and abs(y.std(dtype=np.float64) - st[1]) < 1e-5
    assert abs(float(y.astype(np.float64).sum()) - float(gold["y32_checksum"])) < 2.0


@pytest.mark.gpu
def test_vae_decode_bf16(gold, sd):
    """bf16 operands / fp32 accumulation; GroupNorm statistics, softmax and residual trunk in fp32."""
    ae = FrozenAutoencoderKL(sd, precision="bf16", max_chunk=2, max_latent=32).to("cuda:0")
    y8 = ae.decode(torch.from_numpy(gold["z8"])).cpu().numpy()
    e8 = _err(y8, gold["y8"])
    y32 = ae.decode(torch.from_numpy(gold["z32"])).cpu().numpy()
    e32 = _err(y32[:, :, 96:160, 96:160], gold["y32_slice"])
    print(f"vae bf16 max|err| 8x8 {e8:.3e}  32x32 {e32:.3e}  (output std {gold['y32_stats'][1]:.3f})")
    assert e8 <= 6e-2 and e32 <= 6e-2, (e8, e32)
    rms = float(np.sqrt(np.mean((y8 - gold["y8"]) ** 2)))
    assert rms <= 1e-2, rms


@pytest.mark.gpu
def test_vae_chunking_and_batch_independence(gold, sd):
    """B larger than the workspace chunk: every image decodes as it does alone (bitwise)."""
    ae = FrozenAutoencoderKL(sd, precision="fp32", max_chunk=2, max_latent=8).to("cuda:0")
    g = torch.Generator().manual_seed(3)
    z = torch.randn(5, 4, 8, 8, generator=g)
    y = ae.decode(z).cpu().numpy()
    for i in (0, 3, 4):
        yi = ae.decode(z[i:i + 1]).cpu().numpy()
        assert np.array_equal(yi[0], y[i]), i


@pytest.mark.gpu
def test_vae_errors(sd):
    ae = FrozenAutoencoderKL(sd, precision="fp32", max_chunk=1, max_latent=8).to("cuda:0")
    with pytest.raises(ValueError, match="max_latent"):
        ae.decode(torch.zeros(1, 4, 16, 16))       # larger than the workspace was sized for
    with pytest.raises(RuntimeError):
        ae.decode(torch.zeros(1, 3, 8, 8))
    assert ae.decode(torch.zeros(0, 4, 8, 8)).shape == (0, 3, 64, 64)


@pytest.mark.gpu
def test_latent_sampler_cli_end_to_end(tmp_path, sd):
    """The latent-diffusion branch of the sampler (reference sampler.py:141-150, 320-325): a config with an
    `autoencoder` section loads the KL-VAE checkpoint and decodes the final latents; compared with the oracle's
    DDIM rollout + oracle decode on the same seeds."""
    import subprocess, sys, yaml
    import oracle
    from conftest import REPO, TINY
    from duodiff_amd.config import ModelParams
    from duodiff_amd.weights import synthetic_state_dict
    cfg = dict(TINY, in_chans=4)
    torch.save(dict(sd, **{"encoder.conv_in.bias": torch.zeros(128)}), tmp_path / "ae.pth")   # encode-side keys are skipped
    (tmp_path / "m.yaml").write_text(yaml.safe_dump({
        "model_params": dict(cfg, classifier_type="x"),
        "autoencoder": {"autoencoder_checkpoint_path": str(tmp_path / "ae.pth")}}))
    usd = synthetic_state_dict(ModelParams.from_dict(cfg), 9)
    torch.save(dict(usd), tmp_path / "m.pth")
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "duodiff_amd.sampler", "--seed", "2", "--checkpoint_path", str(tmp_path / "m.pth"),
           "--config_path", str(tmp_path / "m.yaml"), "--batch_size", "2", "--parametrization", "predict_noise",
           "--output_folder", str(out), "--no_png", "--precision", "fp32", "--use_ddim", "--ddim_steps", "6"]
    r = subprocess.run(cmd, cwd=str(REPO), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Decode the images..." in r.stdout
    got = np.load(out / "samples.npy")
    assert got.shape == (2, 64, 64, 3)
    orc = oracle.UViTOracle(ModelParams.from_dict(cfg).as_dict(), {k: v.numpy() for k, v in usd.items()})
    want, _ = oracle.get_samples_ddim(orc, 2, 2, 4, 8, 8, ddim_steps=6, autoencoder=lambda z: vae_decode(z, sd))
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3 * scale)
