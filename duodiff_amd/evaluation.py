"""Output / evaluation plumbing around the sampler.

``read_samples`` mirrors reference utils/evaluation_utils.py:13-25 (every ``*.png`` below a folder except the grid image,
as a float [N,3,H,W] tensor in [0,1]); ``save_images`` mirrors :47-51.  The FID number itself (reference fid.py:34-39,
torchmetrics' FrechetInceptionDistance) needs the Inception-v3 weights and the datasets, neither of which exists offline;
what can be stated without them is the Frechet distance between two feature sets, ``frechet_distance`` -- the statistic
torchmetrics computes after the Inception forward -- so a user with their own feature extractor can close the loop.
"""
from pathlib import Path

import numpy as np
import torch


def read_samples(path):
    from PIL import Image
    tensors = []
    for p in sorted(Path(path).rglob("*.png")):
        if "grid" in p.name:
            continue
        img = np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8)
        tensors.append(torch.from_numpy(img.copy()).permute(2, 0, 1).to(torch.float32) / 255.0)   # ToTensor()
    if not tensors:
        raise RuntimeError(f"no sample PNGs under {path}")
    out = torch.stack(tensors, dim=0)
    print(f"Read {len(out)} images")
    return out


def save_images(images, path):
    """images: iterable of [3,H,W] float tensors in [0,1] -> <path>/<idx>.png (8-bit, round-half-up like save_image)."""
    from PIL import Image
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    for idx, img in enumerate(images):
        arr = (torch.as_tensor(img).detach().cpu().float().clamp(0, 1) * 255 + 0.5).to(torch.uint8)
        Image.fromarray(arr.permute(1, 2, 0).numpy()).save(path / f"{idx}.png")


def feature_statistics(features):
    f = np.asarray(features, dtype=np.float64)
    return f.mean(axis=0), np.cov(f, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2):
    """|mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2)); the trace of the square root from the eigenvalues of S1 S2."""
    mu1, mu2 = np.asarray(mu1, np.float64), np.asarray(mu2, np.float64)
    s1, s2 = np.atleast_2d(np.asarray(sigma1, np.float64)), np.atleast_2d(np.asarray(sigma2, np.float64))
    eig = np.linalg.eigvals(s1 @ s2)
    tr_sqrt = np.sqrt(np.clip(eig.real, 0.0, None)).sum()
    d = mu1 - mu2
    return float(d @ d + np.trace(s1) + np.trace(s2) - 2.0 * tr_sqrt)


def frechet_distance_from_features(real_features, generated_features):
    return frechet_distance(*feature_statistics(real_features), *feature_statistics(generated_features))
