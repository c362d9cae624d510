"""Oracle (test infrastructure): the 1000-step DDPM sampling loops.

Restates reference ``sampler.py:82-101,128-139,145-155`` (``get_samples``, DDPM
branch with the DuoDiff backbone switch) and ``ddpm_core.py:106-214``
(``NoiseScheduler.sample``, uvit branch).  ``model`` is any callable
``model(x, t, y) -> eps`` over numpy arrays (normally ``UViTOracle``).

Randomness: the reference draws everything from torch's CPU mt19937 stream
(``seed_everything`` utils/train_utils.py:8-12 then ``torch.randn``); torch is used
here only as that random-number source.
"""
import random

import numpy as np
import torch

from .schedule_oracle import (ddim_step, ddpm_step, predict_original_step, predict_previous_step, sampler_schedule,
                              scheduler_schedule)

F32 = np.float32


def seed_everything(seed):
    """utils/train_utils.py:8-12."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def get_samples(model, batch_size, seed, num_channels, sample_height, sample_width,
                timesteps_save=(), y=None, late_model=None, t_switch=np.inf,
                num_steps=1000, record=None):
    """sampler.py:82-155, DDPM branch (use_ddim=False, predict_noise, no autoencoder).

    ``num_steps`` < 1000 runs only the first ``num_steps`` iterations (t = 999 ...),
    for bounded CPU baselines; the reference always runs 1000.
    ``record`` (optional dict): x after selected t (keys = t) for fixture F4.
    Returns (samples[B,H,W,C] fp32 = (x+1)/2, intermediate list).
    """
    tables = sampler_schedule()
    seed_everything(seed)                                                    # :99
    x = torch.randn(batch_size, num_channels, sample_height, sample_width).numpy()  # :100
    inter = []
    steps_done = 0
    for t in range(999, -1, -1):                                             # :129
        if steps_done >= num_steps:
            break
        time_tensor = (F32(t) * np.ones(batch_size, F32)).astype(F32)        # :130
        eps = model(x, time_tensor, y)                                       # :131-132
        z = torch.randn(x.shape).numpy() if t > 0 else None                  # :52
        x = ddpm_step(x, eps, z, t, tables)                                  # :133
        if t == 1000 - t_switch:                                             # :135-136
            model = late_model
        if 1000 - t in timesteps_save:                                       # :138-139
            inter.append(x)
        if record is not None and t in record.get("_want", ()):
            record[t] = x.copy()
        steps_done += 1
    samples = ((x + F32(1)) / F32(2)).astype(F32).transpose(0, 2, 3, 1)      # :145-146
    inter = [((v + F32(1)) / F32(2)).astype(F32).transpose(0, 2, 3, 1) for v in inter]
    return np.ascontiguousarray(samples), inter


def scheduler_sample(model, num_steps, data_shape, num_samples, seed, variance_mode="beta"):
    """ddpm_core.py:106-214, ``model_type="uvit"``: own generator, int timesteps, sigma^2 = beta.

    Returns (x_0, samples_over_time list) like the reference's (x_t, logging_dict).
    """
    tables = scheduler_schedule(beta_steps=num_steps)
    gen = torch.Generator(device="cpu").manual_seed(seed)                    # :138
    x = torch.randn((num_samples, *data_shape), generator=gen).numpy()       # :143-145
    over_time = []
    variance = "beta" if variance_mode == "beta" else "beta_tilde"
    for t in range(num_steps - 1, -1, -1):                                   # :147
        time_tensor = np.full((num_samples,), t, np.int64)                   # :150
        eps = model(x, time_tensor, None)                                    # :152
        z = torch.randn(x.shape, generator=gen).numpy() if t > 0 else None   # :167-172
        x = ddpm_step(x, eps, z, t, tables, variance=variance)               # :190-193
        over_time.append(x)                                                  # :210
    return x, over_time


def get_samples_ddim(model, batch_size, seed, num_channels, sample_height, sample_width, ddim_steps=50, ddim_eta=0.0,
                     timesteps_save=(), y=None, late_model=None, t_switch=np.inf, autoencoder=None):
    """sampler.py:103-126: the DDIM branch of get_samples, including its two quirks (sigma^2-scaled noise;
    the late model takes over once t < 1000 - t_switch, tested AFTER the step)."""
    tables = sampler_schedule()
    seed_everything(seed)
    x = torch.randn(batch_size, num_channels, sample_height, sample_width).numpy()
    inter = []
    ts = np.linspace(0, 999, ddim_steps).astype(int)[::-1]
    for t, s in zip(ts[:-1], ts[1:]):
        t, s = int(t), int(s)
        eps = model(x, (F32(t) * np.ones(batch_size, F32)).astype(F32), y)
        z = torch.randn(x.shape).numpy() if s > 0 else None
        x = ddim_step(x, eps, z, t, s, ddim_eta, tables)
        if t < 1000 - t_switch:
            model = late_model
        if 1000 - t in timesteps_save:
            inter.append(x)
    if autoencoder is not None:          # sampler.py:141-143,149-150: `autoencoder` is a callable latents -> images
        x = autoencoder(x)
        inter = [autoencoder(v) for v in inter]
    samples = ((x + F32(1)) / F32(2)).astype(F32).transpose(0, 2, 3, 1)
    inter = [((v + F32(1)) / F32(2)).astype(F32).transpose(0, 2, 3, 1) for v in inter]
    return np.ascontiguousarray(samples), inter


POSTPROCESSING = {"predict_noise": ddpm_step, "predict_original": predict_original_step,
                  "predict_previous": predict_previous_step}
