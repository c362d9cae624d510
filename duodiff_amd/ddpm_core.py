"""Host-side mirror of the reference's ``NoiseScheduler`` (ddpm_core.py:55-214), sampling part.

Tables come from the engine (``dd_schedule_table``) and are bit-equal to the reference's;
``sample`` drives the fused HIP step.  ``add_noise`` (training) and the early-exit samplers
are out of scope.
"""
from collections import defaultdict

import torch

from .engine import schedule_tables


class NoiseScheduler:
    def __init__(self, beta_init=1e-4, beta_final=0.02, beta_steps=1000, variance_mode="beta"):
        if (beta_init, beta_final, beta_steps) != (1e-4, 0.02, 1000):
            raise NotImplementedError("the engine's schedule is the reference default linspace(1e-4, 0.02, 1000)")
        if variance_mode not in ("beta", "beta_tilde"):
            raise ValueError("Invalid variance mode. Choose 'beta' or 'beta_tilde'.")
        self.beta_init, self.beta_final, self.beta_steps = beta_init, beta_final, beta_steps
        self.variance_mode = variance_mode
        t = schedule_tables()
        self.betas = torch.from_numpy(t["betas"].copy())
        self.alphas = torch.from_numpy(t["alphas"].copy())
        self.alphas_bar = torch.from_numpy(t["alphas_bar"].copy())
        self.alpha_bar_prev = torch.from_numpy(t["alphas_bar_previous"].copy())
        self.betas_tilde = torch.from_numpy(t["betas_tilde_scheduler"].copy())  # ddpm_core.py:68-70 rounding order

    def sigma_squared(self):
        if self.variance_mode == "beta":
            return self.betas
        if self.variance_mode == "beta_tilde":
            return self.betas_tilde
        raise ValueError("Invalid variance mode. Choose 'beta' or 'beta_tilde'.")

    def set_device(self, device):
        for n in ("betas", "alphas", "alphas_bar", "alpha_bar_prev", "betas_tilde"):
            setattr(self, n, getattr(self, n).to(device))

    def sample(self, model, num_steps, data_shape, num_samples, seed, model_type="uvit", generator_device="cpu",
               keep_samples_over_time=True, **_unused):
        """ddpm_core.py:106-214, uvit branch.  Returns (x_0, logging_dict).

        generator_device="cpu" draws x_T and z from a torch CPU generator seeded with ``seed``
        (the stream the reference produces on a CPU device); "cuda" uses torch's device generator
        as the reference does on a GPU.
        """
        if model_type != "uvit":
            raise NotImplementedError("only model_type='uvit' is on the DuoDiff sampling path")
        if num_steps != self.beta_steps:
            raise NotImplementedError("num_steps must equal beta_steps (1000)")
        dev = model.device
        gen = torch.Generator(device=generator_device).manual_seed(seed)
        logging_dict = defaultdict(list)
        x = torch.randn((num_samples, *data_shape), generator=gen, device=generator_device).to(dev).contiguous()
        m = model.engine_model(num_samples)
        for t in range(num_steps - 1, -1, -1):
            z = None
            if t > 0:
                z = torch.randn(x.size(), generator=gen, device=generator_device).to(dev)
            m.sample_step(x, t, z=z, noise="buffer", variance=self.variance_mode)
            if keep_samples_over_time:
                logging_dict["samples_over_time"].append(x.clone())
        return x, logging_dict
