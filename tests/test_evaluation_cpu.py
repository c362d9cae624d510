"""Output plumbing: PNG dump / grid / statistics.txt / read_samples round trip, Frechet distance."""
import numpy as np
import torch

from duodiff_amd import evaluation, sampler


def test_dump_and_read_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    samples = rng.uniform(-0.2, 1.2, size=(5, 8, 8, 3)).astype(np.float32)       # out-of-range values get clipped
    sampler.dump_samples(samples, tmp_path)
    sampler.dump_samples(samples[:2], tmp_path, timestep=300)
    sampler.dump_statistics(2.0, tmp_path, batch_size=5)
    names = sorted(p.name for p in tmp_path.glob("*.png"))
    assert names == sorted([f"{i}.png" for i in range(5)] + ["0_300.png", "1_300.png", "grid_image.png"])
    assert (tmp_path / "statistics.txt").read_text().splitlines()[0] == "Elapsed time: 2.0 s"
    got = evaluation.read_samples(tmp_path)                                         # the grid image is skipped
    assert got.shape == (7, 3, 8, 8) and got.dtype == torch.float32
    by_name = {p.name: i for i, p in enumerate(sorted(q for q in tmp_path.rglob("*.png") if "grid" not in q.name))}
    want = np.clip(samples[3], 0, 1).transpose(2, 0, 1)
    assert np.abs(got[by_name["3.png"]].numpy() - want).max() <= 1.0 / 255 + 1e-6


def test_save_images_round_trip(tmp_path):
    imgs = torch.rand(3, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    evaluation.save_images(imgs, tmp_path / "x")
    got = evaluation.read_samples(tmp_path / "x")
    assert got.shape == imgs.shape and (got - imgs).abs().max() <= 0.5 / 255 + 1e-6


def test_frechet_distance_closed_forms():
    mu, s = np.array([1.0, -2.0, 0.5]), np.diag([1.0, 4.0, 9.0])
    assert abs(evaluation.frechet_distance(mu, s, mu, s)) < 1e-9
    # commuting covariances: Tr(S1 + S2 - 2 sqrt(S1 S2)) = sum (sqrt(a) - sqrt(b))^2
    s2 = np.diag([4.0, 1.0, 1.0])
    want = ((mu - (mu + 1.0)) ** 2).sum() + ((np.sqrt(np.diag(s)) - np.sqrt(np.diag(s2))) ** 2).sum()
    assert abs(evaluation.frechet_distance(mu, s, mu + 1.0, s2) - want) < 1e-9
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(4000, 6)), rng.normal(size=(4000, 6)) * 2.0 + 1.0
    d = evaluation.frechet_distance_from_features(a, b)
    assert abs(d - (6 * 1.0 + 6 * (2.0 - 1.0) ** 2)) < 0.6
