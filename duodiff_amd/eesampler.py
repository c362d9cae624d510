"""Mirror of the reference's early-exit sampler CLI (eesampler.py): same flags, same loop, device work in libduodiff.so.

    python -m duodiff_amd.eesampler --threshold 0.1 --checkpoint_path ee.pth --batch_size 128 \\
        --output_folder out --config_path configs/deediff_celeba.yaml

Per step (eesampler.py:56-81): EarlyExitUViT.forward -> per-sample exit selection with the global threshold ->
DDPM update with sigma_t = sqrt(beta-tilde_t).  As in the reference the exit is *simulated*: every layer runs and
the per-sample output is gathered afterwards.  z comes from the torch CPU generator (``--noise torch_cpu``, the
stream a CPU run of the reference draws from) or the whole loop runs on the device (``--noise device``: dd_sample_early_exit, one
hipGraph replay per step, Philox noise).
"""
import time
from argparse import ArgumentParser
from pathlib import Path

import numpy as np
import torch

from .autoencoder import get_autoencoder
from .config import load_config
from .early_exit import EarlyExitUViT
from .engine import Context
from .sampler import dump_samples, load_checkpoint, seed_everything
from .uvit import UViT


def get_samples(model, batch_size: int, seed: int, num_channels: int, sample_height: int, sample_width: int,
                threshold: float, depth: int, y=None, autoencoder=None, *, noise: str = "torch_cpu",
                num_steps: int = 1000):
    """reference eesampler.py:40-89 -> (samples [B,H,W,C] numpy, error_prediction_by_timestep [1000,depth],
    indices_by_timestep [1000,B]) (the last two as torch CPU tensors, like the reference)."""
    device = model.device
    ctx = Context.get(device)
    seed_everything(seed)
    x = torch.randn(batch_size, num_channels, sample_height, sample_width).to(device).contiguous()
    if y is not None:
        y = torch.as_tensor(y).to(device, torch.int64).contiguous()
    err_dev = torch.zeros(1000, depth, device=device)
    ind_dev = torch.zeros(1000, batch_size, device=device, dtype=torch.int32)
    if noise in ("device", "device_eager", "device_none"):
        # the whole loop on the device (dd_sample_early_exit): one hipGraph replay per step, z from the device Philox
        # generator ("device_none": no noise term, "device_eager": the same launches without the graph -- tests / timing)
        from .engine import sample_early_exit_loop
        sample_early_exit_loop(ctx, model.engine_model(batch_size), x, threshold, t_start=999, t_end=1000 - int(num_steps), y=y,
                               seed=seed, noise="none" if noise == "device_none" else "philox", err=err_dev, idx=ind_dev,
                               use_graph=noise != "device_eager")
    else:
        for t in range(999, 999 - int(num_steps), -1):
            time_tensor = t * torch.ones(batch_size, device=device)
            eps, cls, outs = model.forward_device(x, time_tensor, y)                       # :58-59
            mo, idx, err = ctx.early_exit_select(outs, eps, cls, threshold)                  # :61-67
            err_dev[t] = err                                                                  # :70-71 (D2D row copies)
            ind_dev[t] = idx
            if t > 0 and noise != "none":                                                     # :77
                z = torch.randn(x.shape).to(device) if noise == "torch_cpu" else torch.randn(x.shape, device=device)
            else:
                z = None
            ctx.ddpm_step(x, mo, z, t, variance="beta_tilde", out=x)                          # :73-81
    if autoencoder is not None:
        x = autoencoder.decode(x)
    samples = ((x + 1) / 2).permute(0, 2, 3, 1).contiguous()
    return samples.cpu().numpy(), err_dev.cpu(), ind_dev.to(torch.float32).cpu()


def dump_statistics(elapsed_time, error_prediction_by_timestep, indices_by_timestep, output_folder: Path):
    with open(output_folder / "statistics.txt", "w") as f:                                # eesampler.py:105-113
        f.write(f"Elapsed time: {elapsed_time} s\n")
    torch.save(error_prediction_by_timestep, output_folder / "error_prediction_by_timestep.pt")
    torch.save(indices_by_timestep, output_folder / "indices_by_timestep.pt")


def get_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--threshold", type=float, required=True)
    p.add_argument("--checkpoint_path", type=str, required=True)
    p.add_argument("--batch_size", type=int, required=True)
    p.add_argument("--output_folder", type=str, required=True)
    p.add_argument("--config_path", type=str, required=True, help="Path to yaml config file")
    p.add_argument("--class_id", type=int, default=None, help="Number up to 1000 that corresponds to a class")
    # engine options (not in the reference)
    p.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    p.add_argument("--noise", choices=["torch_cpu", "device"], default="torch_cpu")
    p.add_argument("--autoencoder_checkpoint_path", type=str, default=None)
    p.add_argument("--no_png", action="store_true", help="write samples.npy instead of PNG files")
    return p.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    out = Path(args.output_folder)
    out.mkdir(parents=True, exist_ok=True)
    config = load_config(args.config_path)
    mp = dict(config["model_params"])
    classifier_type = mp.pop("classifier_type")                                          # eesampler.py:158
    base = UViT(**mp, precision=args.precision, max_batch=args.batch_size)
    model = EarlyExitUViT(base, classifier_type)
    model.load_state_dict(load_checkpoint(args.checkpoint_path))
    model = model.eval().to("cuda")
    seed_everything(args.seed)
    y = torch.randint(1, 1001, (args.batch_size,)) if args.class_id is not None else None   # :177-181
    if y is not None and int(y.max()) >= base.num_classes:
        raise IndexError("index out of range in self")
    autoencoder = None
    if "autoencoder" in config:
        path = args.autoencoder_checkpoint_path or config["autoencoder"]["autoencoder_checkpoint_path"]
        autoencoder = get_autoencoder(path, precision=args.precision).to(model.device)
    tic = time.time()
    samples, err, ind = get_samples(model, args.batch_size, args.seed, base.in_chans, base.params.img_size,
                                    base.params.img_size, args.threshold, base.depth, y=y, autoencoder=autoencoder,
                                    noise=args.noise)
    tac = time.time()
    dump_statistics(tac - tic, err, ind, out)
    if args.no_png:
        np.save(out / "samples.npy", samples)
    else:
        dump_samples(samples, out)
    print(f"Elapsed time: {tac - tic} s")


if __name__ == "__main__":
    main()
