"""DuoDiff sampling driver: the reference's ``sampler.py`` surface on the MI355X engine.

Same command line (reference sampler.py:192-252), YAML configs and checkpoint formats; the
1000-step loop of ``get_samples`` (sampler.py:82-155) runs as fused HIP steps -- either one
hipGraph replay per step entirely on the device (``--noise device``), or step by step with z
drawn from the torch CPU stream exactly as the reference draws it (``--noise torch_cpu``,
bit-identical noise for parity runs).

    python -m duodiff_amd.sampler --config_path configs/uvit_celeba_3.yaml --checkpoint_path s.pth \
        --config_path_late configs/uvit_celeba.yaml --checkpoint_path_late f.pth --t_switch 300 \
        --batch_size 128 --parametrization predict_noise --output_folder out

DDIM (--use_ddim) and the predict_original / predict_previous parametrizations run
the U-ViT forward on the engine plus one fused affine update per step; ImageNet-256 latents are decoded by the
engine's KL-VAE decoder when the YAML carries an ``autoencoder`` block (next-1).
"""
import math
import random
import time
from argparse import ArgumentParser
from pathlib import Path
from typing import List

import numpy as np
import torch

from .config import ModelParams, load_config
from .autoencoder import get_autoencoder
from .engine import Context, sample_affine_loop, sample_loop, schedule_tables
from .uvit import UViT


def get_device():
    """reference sampler.py:25-38 picks cuda:0 > mps > cpu; this engine exists only for the GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("duodiff_amd needs an MI355X GPU; there is no CPU sampling path")
    return f"cuda:{torch.cuda.current_device()}"


def seed_everything(seed):
    """reference utils/train_utils.py:8-12."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


class _Schedule:
    """Module-level tables of reference sampler.py:40-44, materialised lazily from the engine."""

    def __getattr__(self, name):
        t = schedule_tables()
        self.betas = torch.from_numpy(t["betas"].copy())
        self.alphas = torch.from_numpy(t["alphas"].copy())
        self.alphas_bar = torch.from_numpy(t["alphas_bar"].copy())
        self.alphas_bar_previous = torch.from_numpy(t["alphas_bar_previous"].copy())
        self.betas_tilde = torch.from_numpy(t["betas_tilde"].copy())
        if name in self.__dict__:
            return self.__dict__[name]
        raise AttributeError(name)


schedule = _Schedule()


def predict_noise_postprocessing(model_output, x, t, z=None):
    """reference sampler.py:47-56 on device tensors.  ``z``: the noise to use (default: drawn from
    the torch CPU stream, like ``randn_like`` on a CPU reference run); ignored at t == 0."""
    ctx = Context.get(x.device)
    if t > 0 and z is None:
        z = torch.randn(x.shape).to(x.device)
    return ctx.ddpm_step(x.contiguous(), model_output, z if t > 0 else None, t)


def _f32(v):
    return np.float32(v)


def affine_coefficients(kind, t, s=None, eta=0.0):
    """Scalar (a, b, c) with x' = a*x + b*model_output + c*z for the reference's other updates, in fp32
    from the engine's (bit-exact) tables:
      "predict_original"  sampler.py:59-72      "predict_previous"  sampler.py:75-79
      "ddim" (t -> s)     sampler.py:112-120 (noise scaled by sigma^2 = betas_tilde[t]*eta, as the reference does)
    """
    tb = schedule_tables()
    one = _f32(1)
    if kind == "predict_previous":
        return _f32(0), one, np.sqrt(tb["betas_tilde"][t])
    if kind == "predict_original":
        a_t, ab_t, ab_p, b_t = tb["alphas"][t], tb["alphas_bar"][t], tb["alphas_bar_previous"][t], tb["betas"][t]
        a = _f32(np.sqrt(a_t) * (one - ab_p)) / (one - ab_t)
        b = _f32(np.sqrt(ab_p) * b_t) / (one - ab_t)
        return _f32(a), _f32(b), np.sqrt(tb["betas_tilde"][t])
    if kind == "ddim":
        ab = tb["alphas_bar"]
        sig2 = _f32(tb["betas_tilde"][t] * _f32(eta))
        a = np.sqrt(_f32(ab[s] / ab[t]))
        b = _f32(np.sqrt(_f32(one - ab[s] - sig2)) - _f32(a * np.sqrt(_f32(one - ab[t]))))
        return _f32(a), b, sig2
    raise ValueError(kind)


def _affine_post(kind, model_output, x, t, z=None):
    ctx = Context.get(x.device)
    a, b, c = affine_coefficients(kind, t)
    if t > 0 and z is None:
        z = torch.randn(x.shape).to(x.device)          # randn_like on the torch CPU stream (sampler.py:67,77)
    return ctx.affine_step(x, model_output, z if t > 0 else None, a, b, c)


def predict_original_postprocessing(model_output, x, t, z=None):
    """reference sampler.py:59-72 (model predicts x_0) on device tensors."""
    return _affine_post("predict_original", model_output, x, t, z)


def predict_previous_postprocessing(model_output, x, t, z=None):
    """reference sampler.py:75-79 (model predicts x_{t-1}) on device tensors."""
    return _affine_post("predict_previous", model_output, x, t, z)


def get_samples(model, batch_size: int, postprocessing: callable, seed: int, num_channels: int,
                sample_height: int, sample_width: int, use_ddim: bool = False, ddim_steps: int = 50,
                ddim_eta: float = 0.0, timesteps_save: List[int] = (), y=None, autoencoder=None,
                late_model=None, t_switch=np.inf, *, noise: str = "torch_cpu", use_graph: bool = True,
                num_steps: int = 1000, return_device_tensor: bool = False):
    """reference sampler.py:82-155.  Returns (samples[B,H,W,C] float32 numpy = (x+1)/2, intermediates).

    noise="torch_cpu": x_T and every z come from the torch CPU generator after seed_everything(seed),
        in the reference's order -> identical random numbers to a CPU reference run.
    noise="device": x_T as above, z from the device Philox generator inside the graph-replayed loop.
    num_steps < 1000 runs only the first steps (t = 999 ...), for bounded benchmarks.
    """
    device = model.device
    seed_everything(seed)                                                    # sampler.py:99
    x = torch.randn(batch_size, num_channels, sample_height, sample_width).to(device).contiguous()  # :100
    if y is not None:
        y = torch.as_tensor(y).to(device, torch.int64).contiguous()
    intermediate = []
    saves = set(int(v) for v in timesteps_save)
    t_last = 1000 - int(num_steps)
    first = model.engine_model(batch_size)
    late = late_model.engine_model(batch_size) if late_model is not None else None
    ctx = first.ctx
    # the DDPM loop switches AFTER the step at t == 1000 - t_switch (sampler.py:135-136): a t_switch outside
    # [1, 1000] (0, negative, > 1000, inf) never matches a t in 999..0, i.e. the first model runs every step
    switch_t = None
    if late is not None and np.isfinite(t_switch) and 0 <= 1000 - int(t_switch) <= 999:
        switch_t = 1000 - int(t_switch)

    def draw(shape):
        return torch.randn(shape).to(device) if noise == "torch_cpu" else torch.randn(shape, device=device)

    def affine_segments(steps, switch_after):
        """noise == "device": the table-driven loop on the device (dd_sample_affine: one hipGraph replay per step, Philox z),
        cut at the save points.  steps: [(t, a, b, c, draws_noise, saves_after)].  Every segment keeps the seed and passes its
        first step's index as the Philox counter base: the final samples do not depend on where the loop is cut."""
        k0 = 0
        while k0 < len(steps):
            k1 = next((k + 1 for k in range(k0, len(steps)) if steps[k][5]), len(steps))
            seg = steps[k0:k1]
            sw = None if switch_after is None else min(max(switch_after - k0, 0), len(seg))
            seg_first, seg_late = (first, late) if (sw is None or sw > 0) else (late, None)
            sample_affine_loop(ctx, seg_first, seg_late if sw is not None and 0 < sw < len(seg) else None, x,
                               [v[0] for v in seg], [v[1] for v in seg], [v[2] for v in seg], [v[3] for v in seg],
                               [int(v[4]) for v in seg], switch_after=sw, y=y, seed=seed, counter_base=k0, noise="philox", use_graph=use_graph)
            if seg[-1][5]:
                intermediate.append(x.clone())
            k0 = k1

    if use_ddim:
        # reference sampler.py:103-126.  U-ViT forward on the engine + one fused affine update per step.
        ts = np.linspace(0, 999, ddim_steps).astype(int)[::-1]
        pairs = [(int(t), int(s_)) for t, s_ in zip(ts[:-1], ts[1:])]
        if noise == "device":
            steps, switch_after = [], None
            for k, (t, s_) in enumerate(pairs):
                a, b, c = affine_coefficients("ddim", t, s_, ddim_eta)
                steps.append((float(t), a, b, c, s_ > 0, (1000 - t) in saves))
                if switch_after is None and late is not None and t < 1000 - t_switch:   # :122-123: from the NEXT step on
                    switch_after = k + 1
            affine_segments(steps, switch_after)
        else:
            eps = torch.empty_like(x)
            cur = first
            for t, s_ in pairs:
                cur.forward(x, float(t), y, out=eps)                             # :108-110
                a, b, c = affine_coefficients("ddim", t, s_, ddim_eta)           # :112-117
                z = draw(x.shape) if s_ > 0 else None                            # :119
                ctx.affine_step(x, eps, z, a, b, c, out=x)                       # :120
                if late is not None and t < 1000 - t_switch:                     # :122-123
                    cur = late
                if 1000 - t in saves:                                            # :125-126
                    intermediate.append(x.clone())
    elif postprocessing is not predict_noise_postprocessing:
        # predict_original / predict_previous (sampler.py:59-79): same loop, affine update
        kind = {predict_original_postprocessing: "predict_original",
                predict_previous_postprocessing: "predict_previous"}.get(postprocessing)
        if kind is None:
            raise ValueError("postprocessing must be one of this module's predict_*_postprocessing functions")
        if noise == "device":
            steps, switch_after = [], None
            for k, t in enumerate(range(999, t_last - 1, -1)):
                a, b, c = affine_coefficients(kind, t)
                steps.append((float(t), a, b, c, t > 0, (1000 - t) in saves))
                if switch_t is not None and t == switch_t:
                    switch_after = k + 1
            affine_segments(steps, switch_after)
        else:
            eps = torch.empty_like(x)
            cur = first
            for t in range(999, t_last - 1, -1):
                cur.forward(x, float(t), y, out=eps)
                a, b, c = affine_coefficients(kind, t)
                ctx.affine_step(x, eps, draw(x.shape) if t > 0 else None, a, b, c, out=x)
                if switch_t is not None and t == switch_t:
                    cur = late
                if 1000 - t in saves:
                    intermediate.append(x.clone())
    elif noise == "device":
        # segments between save points; each segment is one dd_sample call (graph replays)
        stops = sorted({1000 - s for s in saves if t_last <= 1000 - s <= 999}, reverse=True)
        t = 999
        cur_first, cur_late, cur_switch = first, late, (int(t_switch) if switch_t is not None else 0)
        while t >= t_last:
            seg_end = next((s for s in stops if s <= t), t_last)
            if switch_t is not None and t <= switch_t - 1 and cur_late is not None:
                cur_first, cur_late, cur_switch = late, None, 0  # already past the switch
            sample_loop(ctx, cur_first, cur_late, x, t_switch=cur_switch, t_start=t, t_end=seg_end, y=y,
                        seed=seed, noise="philox", use_graph=use_graph)
            if seg_end in stops:
                intermediate.append(x.clone())
            t = seg_end - 1
    elif noise == "torch_cpu":
        cur = first
        for t in range(999, t_last - 1, -1):                                 # :129
            z = torch.randn(x.shape).to(device) if t > 0 else None           # :52 (CPU stream)
            cur.sample_step(x, t, y=y, z=z, noise="buffer")                  # :130-133
            if switch_t is not None and t == switch_t:                       # :135-136
                cur = late
            if 1000 - t in saves:                                            # :138-139
                intermediate.append(x.clone())
    else:
        raise ValueError("noise must be 'torch_cpu' or 'device'")

    if autoencoder is not None:                                              # :141-143, :149-150
        print("Decode the images...")
        x = autoencoder.decode(x)
        intermediate = [autoencoder.decode(v) for v in intermediate]
    samples = ctx.to_images(x.contiguous())                                  # :145-146, library kernel
    inter = [ctx.to_images(v.contiguous()).cpu().numpy() for v in intermediate]
    if return_device_tensor:
        return samples, inter
    return samples.cpu().numpy(), inter                                      # :155 (D2H boundary)


def draw_labels(batch_size: int, num_classes: int):
    """reference sampler.py:314-318: labels are randint(1, 1001) whatever --class_id says (quirk Q4); a label the embedding
    table does not hold raises the IndexError nn.Embedding raises there (the engine would read past label_emb)."""
    y = torch.randint(1, 1001, (batch_size,))
    if int(y.max()) >= num_classes or int(y.min()) < 0:
        raise IndexError("index out of range in self")
    return y


def dump_samples(samples, output_folder: Path, timestep=1000):
    """reference sampler.py:158-184: per-image PNG + grid, values clipped to [0, 1]."""
    from matplotlib import pyplot as plt
    n = len(samples)
    grid = math.ceil(math.sqrt(n))
    h, w = samples[0].shape[:2]
    grid_img = np.zeros((grid * h, grid * w, 3))
    for i, s in enumerate(samples):
        s = np.clip(s, 0, 1)
        name = f"{i}_{timestep}.png" if timestep != 1000 else f"{i}.png"
        plt.imsave(output_folder / name, s)
        r, c = divmod(i, grid)
        grid_img[r * h:(r + 1) * h, c * w:(c + 1) * w, :] = s[..., :3]
    plt.imsave(output_folder / "grid_image.png", grid_img)


def dump_statistics(elapsed_time, output_folder: Path, batch_size=None):
    with open(output_folder / "statistics.txt", "w") as f:
        f.write(f"Elapsed time: {elapsed_time} s\n")                          # reference sampler.py:187-189
        if batch_size:
            f.write(f"Images per second: {batch_size / elapsed_time}\n")


def get_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--checkpoint_path", type=str, required=True, help="Path to checkpoint of the model")
    p.add_argument("--autoencoder_checkpoint_path", type=str, default=None,
                   help="(engine option) overrides config['autoencoder']['autoencoder_checkpoint_path']")
    p.add_argument("--checkpoint_path_late", type=str, default=None,
                   help="Path to checkpoint of the model to be used in the latest steps")
    p.add_argument("--batch_size", type=int, required=True)
    p.add_argument("--parametrization", type=str, required=True,
                   choices=["predict_noise", "predict_original", "predict_previous"])
    p.add_argument("--output_folder", type=str, required=True)
    p.add_argument("--config_path", type=str, required=True, help="Path to yaml config file")
    p.add_argument("--config_path_late", type=str, default=None,
                   help="Path to yaml config file of the model to be used in the latest steps")
    p.add_argument("--t_switch", type=int, default=np.inf,
                   help="Sampling timestep where the model should be replaced by the late model")
    p.add_argument("--class_id", type=int, default=None, help="Number up to 1000 that corresponds to a class")
    p.add_argument("--use_ddim", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--timesteps_save", type=int, nargs="+", default=[])
    # engine options (not in the reference)
    p.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    p.add_argument("--noise", choices=["torch_cpu", "device"], default="device")
    p.add_argument("--no_graph", action="store_true", help="launch kernels eagerly instead of hipGraph replay")
    p.add_argument("--no_png", action="store_true", help="write samples.npy instead of PNG files")
    return p.parse_args(argv)


def load_checkpoint(path):
    """reference sampler.py:289-291: a bare state_dict or {"model_state_dict": ...}."""
    sd = torch.load(path, map_location="cpu")
    return sd["model_state_dict"] if "model_state_dict" in sd else sd


def build_model(config, checkpoint_path, precision, max_batch):
    mp = ModelParams.from_dict(config)
    m = UViT(**mp.as_dict(), precision=precision, max_batch=max_batch)
    m.load_state_dict(load_checkpoint(checkpoint_path))
    return m.eval().to(get_device()), mp


def main(argv=None):
    args = get_args(argv)
    out = Path(args.output_folder)
    out.mkdir(parents=True, exist_ok=True)
    post = {"predict_noise": predict_noise_postprocessing, "predict_original": predict_original_postprocessing,
            "predict_previous": predict_previous_postprocessing}[args.parametrization]

    config = load_config(args.config_path)
    model, mp = build_model(config, args.checkpoint_path, args.precision, args.batch_size)
    model_late = None
    if args.checkpoint_path_late:
        config = load_config(args.config_path_late)
        model_late, _ = build_model(config, args.checkpoint_path_late, args.precision, args.batch_size)

    seed_everything(args.seed)
    y = None
    if args.class_id is not None:
        y = draw_labels(args.batch_size, mp.num_classes)
    autoencoder = None
    if "autoencoder" in config:                                              # reference sampler.py:320-325
        ae_path = args.autoencoder_checkpoint_path or config["autoencoder"]["autoencoder_checkpoint_path"]
        autoencoder = get_autoencoder(ae_path, precision=args.precision).to(model.device)

    tic = time.time()
    samples, inter = get_samples(model=model, batch_size=args.batch_size, postprocessing=post, seed=args.seed,
                                 num_channels=mp.in_chans, sample_height=mp.img_size, sample_width=mp.img_size,
                                 use_ddim=args.use_ddim, ddim_steps=args.ddim_steps, ddim_eta=args.ddim_eta,
                                 y=y, autoencoder=autoencoder, late_model=model_late, t_switch=args.t_switch,
                                 timesteps_save=args.timesteps_save, noise=args.noise, use_graph=not args.no_graph)
    tac = time.time()
    dump_statistics(tac - tic, out, args.batch_size)
    if args.no_png:
        np.save(out / "samples.npy", samples)
    else:
        dump_samples(samples, out)
        for ts, smp in zip(args.timesteps_save, inter):
            dump_samples(smp, out, ts)
    print(f"Elapsed time: {tac - tic} s  ({args.batch_size / (tac - tic):.3f} images/s)")


if __name__ == "__main__":
    main()
