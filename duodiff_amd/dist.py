"""Multi-GPU sampling: one process per GPU, batches sharded with no collective inside the loop.

Images are independent (no cross-sample reduction anywhere on the path: LayerNorm and softmax are
per token / per row), so the 1000-step loop shards embarrassingly: rank r samples its own batch
with ``seed = base_seed + r``, which is bit-for-bit what the r-th independent reference invocation
``sampler.py --seed base_seed+r --batch_size B`` computes (the reference has no notion of a global
batch).  Exactly ONE collective closes a run: the gather of the finished images
(RCCL over xGMI when the backend is "nccl", gloo on CPU for tests).

    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m duodiff_amd.dist --config_path ... (same flags as sampler)
"""
import os
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank()
    if world == 1:
        return rank, world, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        # DUODIFF_DIST_BACKEND=gloo forces the CPU transport (rehearsing N ranks on one GPU; tests)
        backend = os.environ.get("DUODIFF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def rank_seed(base_seed: int, rank: int) -> int:
    """Seed of rank r's shard: the r-th independent single-GPU run."""
    return int(base_seed) + int(rank)


def gather_images(local: torch.Tensor, world: int, dst: Optional[int] = None):
    """The single collective: all ranks' [B,H,W,C] images, in rank order.

    dst=None -> all_gather (every rank gets [world*B,H,W,C]); dst=k -> gather to rank k only
    (others get None).  With the nccl backend this is ncclAllGather / ncclGather over xGMI:
    6.3 MB per rank for CelebA B=128, once per >= 1000 steps.
    """
    if world == 1:
        return local
    local = local.contiguous()
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()          # gloo moves host memory; the payload is one image batch per >= 1000 steps
    if dst is None:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local)
        return torch.cat(out, dim=0)
    rank = dist.get_rank()
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, out, dst=dst)
    else:
        out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, gather_list=out, dst=dst)
    return torch.cat(out, dim=0) if rank == dst else None


def sample_sharded(sample_fn: Callable[[int], torch.Tensor], base_seed: int, dst: Optional[int] = None,
                   barrier: bool = True):
    """Run ``sample_fn(seed)`` (one rank's whole sampling loop, returning [B,H,W,C] images) on every
    rank with its own seed, then gather.  No communication happens inside ``sample_fn``.

    Returns (gathered images or None on non-destination ranks, this rank's local images).
    """
    rank, world, _ = env_rank()
    if world > 1 and not dist.is_initialized():
        raise RuntimeError("call duodiff_amd.dist.init() first")
    local = sample_fn(rank_seed(base_seed, rank))
    if not torch.is_tensor(local):
        local = torch.from_numpy(np.asarray(local))
    if world > 1 and barrier:
        dist.barrier()
    return gather_images(local, world, dst), local


def main(argv=None):
    """Sharded version of the sampler CLI: same flags; --batch_size is per GPU; rank 0 writes the output."""
    import time
    from pathlib import Path

    from . import sampler
    args = sampler.get_args(argv)
    rank, world, local_rank = init()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())   # (ranks may share a GPU when rehearsed on one)
    config = sampler.load_config(args.config_path)
    model, mp = sampler.build_model(config, args.checkpoint_path, args.precision, args.batch_size)
    late = None
    if args.checkpoint_path_late:
        late, _ = sampler.build_model(sampler.load_config(args.config_path_late), args.checkpoint_path_late,
                                      args.precision, args.batch_size)
    post = {"predict_noise": sampler.predict_noise_postprocessing, "predict_original": sampler.predict_original_postprocessing,
            "predict_previous": sampler.predict_previous_postprocessing}[args.parametrization]
    autoencoder = None
    if "autoencoder" in config:
        from .autoencoder import get_autoencoder
        path = args.autoencoder_checkpoint_path or config["autoencoder"]["autoencoder_checkpoint_path"]
        autoencoder = get_autoencoder(path, precision=args.precision).to(model.device)

    def one_rank(seed):
        y = None
        if args.class_id is not None:
            sampler.seed_everything(seed)
            y = sampler.draw_labels(args.batch_size, mp.num_classes)      # same range check as the single-GPU CLI
        s, _ = sampler.get_samples(model, args.batch_size, post, seed, mp.in_chans, mp.img_size, mp.img_size,
                                   use_ddim=args.use_ddim, ddim_steps=args.ddim_steps, ddim_eta=args.ddim_eta,
                                   timesteps_save=[], y=y, autoencoder=autoencoder, late_model=late, t_switch=args.t_switch,
                                   noise=args.noise, use_graph=not args.no_graph, return_device_tensor=True)
        return s

    tic = time.time()
    allimgs, _ = sample_sharded(one_rank, args.seed, dst=0)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    tac = time.time()
    if rank == 0:
        out = Path(args.output_folder)
        out.mkdir(parents=True, exist_ok=True)
        imgs = allimgs.cpu().numpy()
        sampler.dump_statistics(tac - tic, out, imgs.shape[0])
        if args.no_png:
            np.save(out / "samples.npy", imgs)
        else:
            sampler.dump_samples(imgs, out)
        print(f"Elapsed time: {tac - tic} s  ({imgs.shape[0] / (tac - tic):.3f} images/s on {world} GPU(s))")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
