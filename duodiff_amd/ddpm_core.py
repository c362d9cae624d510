"""Host-side mirror of the reference's ``NoiseScheduler`` (ddpm_core.py:55-214), sampling part.

Tables for any ``(beta_init, beta_final, beta_steps)`` come from the engine's host arithmetic
(``dd_schedule_build``: torch.linspace / cumprod restated bit for bit); ``sample`` drives the HIP path:
the fused sampling step for the default 1000-step schedule with sigma^2 = beta (the class default), and
``dd_forward`` + ``dd_ddpm_step_coef`` with host-derived scalars for every other schedule / variance mode
(beta-tilde in the ddpm_core rounding order).  ``add_noise`` (training) and the early-exit
branches are out of scope.
"""
from collections import defaultdict

import numpy as np
import torch

from .engine import Context, build_schedule

_F32 = np.float32


class NoiseScheduler:
    def __init__(self, beta_init=1e-4, beta_final=0.02, beta_steps=1000, variance_mode="beta"):
        self.beta_init, self.beta_final, self.beta_steps = beta_init, beta_final, int(beta_steps)
        self.variance_mode = variance_mode  # validated lazily by sigma_squared(), like the reference (ddpm_core.py:72-79)
        t = build_schedule(beta_init, beta_final, beta_steps)
        self.betas = torch.from_numpy(t["betas"])
        self.alphas = torch.from_numpy(t["alphas"])
        self.alphas_bar = torch.from_numpy(t["alphas_bar"])
        self.alpha_bar_prev = torch.from_numpy(t["alpha_bar_prev"])
        self.betas_tilde = torch.from_numpy(t["betas_tilde"])  # ddpm_core.py:68-70 rounding order

    def sigma_squared(self):
        if self.variance_mode == "beta":
            return self.betas
        if self.variance_mode == "beta_tilde":
            return self.betas_tilde
        raise ValueError("Invalid variance mode. Choose 'beta' or 'beta_tilde'.")

    def set_device(self, device):
        for n in ("betas", "alphas", "alphas_bar", "alpha_bar_prev", "betas_tilde"):
            setattr(self, n, getattr(self, n).to(device))

    def _is_engine_default(self):
        return (float(self.beta_init), float(self.beta_final), self.beta_steps) == (1e-4, 0.02, 1000)

    def step_coefficients(self, t):
        """(c1, c2, sigma) of ddpm_core.py:167-193 in fp32, each operation rounded as torch rounds it."""
        alpha_t = _F32(self.alphas[t].item())
        alpha_bar_t = _F32(self.alphas_bar[t].item())
        c1 = np.sqrt(_F32(1) / alpha_t, dtype=_F32)
        c2 = _F32((_F32(1) - alpha_t) / np.sqrt(_F32(1) - alpha_bar_t, dtype=_F32))
        sigma = np.sqrt(_F32(self.sigma_squared()[t].item()), dtype=_F32)
        return float(c1), float(c2), float(sigma)

    def sample(self, model, num_steps, data_shape, num_samples, seed, model_type="uvit", generator_device="cpu",
               keep_samples_over_time=True, fused=None, **_unused):
        """ddpm_core.py:106-214, uvit branch.  Returns (x_0, logging_dict).

        generator_device="cpu" draws x_T and z from a torch CPU generator seeded with ``seed``
        (the stream the reference produces on a CPU device); "cuda" uses torch's device generator
        as the reference does on a GPU.
        fused: None = use the fused sampling step when the schedule is the engine's built-in one and
        sigma^2 = beta; False = always forward + explicit-coefficient update; True = require the fused step.
        """
        if model_type != "uvit":
            raise NotImplementedError("only model_type='uvit' is on the DuoDiff sampling path")
        if num_steps > self.beta_steps:
            raise IndexError(f"num_steps {num_steps} exceeds the schedule's {self.beta_steps} entries")
        self.sigma_squared()   # raises for an invalid variance mode before any device work
        can_fuse = self._is_engine_default() and self.variance_mode == "beta"
        if fused and not can_fuse:
            raise ValueError("the fused step covers the default 1000-step schedule with variance_mode='beta'")
        use_fused = can_fuse if fused is None else bool(fused)
        dev = model.device
        ctx = Context.get(dev)
        gen = torch.Generator(device=generator_device).manual_seed(seed)
        logging_dict = defaultdict(list)
        x = torch.randn((num_samples, *data_shape), generator=gen, device=generator_device).to(dev).contiguous()
        m = model.engine_model(num_samples)
        eps = None if use_fused else torch.empty_like(x)
        for t in range(num_steps - 1, -1, -1):
            z = None
            if t > 0:
                z = torch.randn(x.size(), generator=gen, device=generator_device).to(dev)
            if use_fused:
                m.sample_step(x, t, z=z, noise="buffer", variance="beta")
            else:
                m.forward(x, float(t), out=eps)                          # time_tensor = [t] * B (ddpm_core.py:149-152)
                c1, c2, sigma = self.step_coefficients(t)
                ctx.ddpm_step_coef(x, eps, z, c1, c2, sigma, out=x)      # ddpm_core.py:190-193
            if keep_samples_over_time:
                logging_dict["samples_over_time"].append(x.clone())
        return x, logging_dict
