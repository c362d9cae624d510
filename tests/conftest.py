import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden():
    """Loader for tests/golden/*.npz.  Every floating-point array of a fixture must be finite: a NaN expected value
    would make assert_allclose (equal_nan=True by default) pass vacuously.  The reference's NaN quirks are pinned
    through explicit scalar keys (e.g. ddim_tiny.npz::nan_fraction), never through NaN arrays."""
    def load(name):
        fx = np.load(GOLDEN / name, allow_pickle=False)
        for k in fx.files:
            a = fx[k]
            if np.issubdtype(a.dtype, np.floating):
                assert np.isfinite(a).all(), f"{name}::{k} holds non-finite values"
        return fx
    return load


def tiny_cfg_from_fixture(fx):
    keys = [str(k) for k in fx["cfg_keys"]]
    vals = fx["cfg_vals"]
    cfg = {}
    for k, v in zip(keys, vals):
        if k in ("qkv_bias", "mlp_time_embed", "normalize_timesteps"):
            cfg[k] = bool(v)
        else:
            cfg[k] = int(v)
    return cfg


TINY = dict(img_size=8, patch_size=2, in_chans=3, embed_dim=64, depth=3, num_heads=1, mlp_ratio=4,
            qkv_bias=False, mlp_time_embed=False, num_classes=-1, normalize_timesteps=True)

FULL_NAMES = ["uvit_cifar10", "uvit_cifar10_3", "uvit_celeba", "uvit_celeba_3",
              "uvit_imagenet64", "uvit_imagenet64_3", "uvit_imagenet256", "uvit_imagenet256_3"]
