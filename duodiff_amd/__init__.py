"""duodiff_amd: MI355X-native DuoDiff sampling engine.

One hot path, nothing else: the 1000-step DDPM denoising loop of the reference's
``sampler.py`` / ``ddpm_core.py`` driving the U-ViT forward of ``models/uvit.py``
with the shallow<->full backbone switch at ``t_switch``.  The arithmetic runs in
hand-written HIP kernels for gfx950 behind a C ABI (``include/duodiff.h``,
``libduodiff.so``); this Python package is the host-side mirror of the reference's
call surface (UViT, get_samples, NoiseScheduler, the sampler CLI) and never
computes on the CPU: if the HIP library is missing, construction fails loudly.
"""
from .config import ModelParams, load_config  # noqa: F401
from .weights import param_shapes, synthetic_state_dict, num_params  # noqa: F401

__all__ = ["ModelParams", "load_config", "param_shapes", "synthetic_state_dict", "num_params"]
__version__ = "0.1.0"
