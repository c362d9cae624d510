"""CPU, world_size 2 (gloo): the multi-GPU sharding contract of duodiff_amd.dist.

The engine itself needs a GPU, so the per-rank sampler here is the numpy oracle on a tiny model
(the oracle is the checker the GPU path is held to elsewhere).  What is tested is the host-side
distributed logic: rank r samples with seed base+r, nothing is exchanged inside the loop, and the
single final gather returns the shards in rank order -- i.e. the sharded run equals two independent
single-process runs with seeds base and base+1.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]
TINY = dict(img_size=8, patch_size=2, in_chans=3, embed_dim=64, depth=1, num_heads=1, mlp_ratio=4,
            qkv_bias=False, mlp_time_embed=False, num_classes=-1, normalize_timesteps=True)
STEPS = 6


def _sample(seed):
    sys.path.insert(0, str(REPO))
    import oracle
    from duodiff_amd.config import ModelParams
    from duodiff_amd.weights import synthetic_state_dict
    mpar = ModelParams.from_dict(TINY)
    m = oracle.UViTOracle(TINY, {k: v.numpy() for k, v in synthetic_state_dict(mpar, 5).items()})
    imgs, _ = oracle.get_samples(m, 2, seed, 3, 8, 8, num_steps=STEPS)
    return torch.from_numpy(imgs)


def _worker(rank, world, port, dst, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, str(REPO))
    torch.set_num_threads(1)
    from duodiff_amd import dist as ddist
    r, w, _ = ddist.init("gloo")
    assert (r, w) == (rank, world)
    gathered, local = ddist.sample_sharded(_sample, base_seed=40, dst=dst)
    np.save(Path(out_dir) / f"local_{rank}.npy", local.numpy())
    if gathered is not None:
        np.save(Path(out_dir) / f"gathered_{rank}.npy", gathered.numpy())
    torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dst", [None, 0])
def test_two_rank_sharding_matches_independent_runs(tmp_path, dst):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), dst, str(tmp_path)), nprocs=world, join=True)
    want = [_sample(40 + r).numpy() for r in range(world)]
    assert not np.array_equal(want[0], want[1])                  # different seeds -> different images
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"local_{r}.npy"), want[r])
    holders = range(world) if dst is None else [dst]
    for r in holders:
        g = np.load(tmp_path / f"gathered_{r}.npy")
        assert g.shape == (4, 8, 8, 3)
        assert np.array_equal(g, np.concatenate(want, axis=0))   # rank order, bit-exact
    if dst is not None:
        assert not (tmp_path / "gathered_1.npy").exists()


def test_single_process_path_is_a_no_op():
    from duodiff_amd import dist as ddist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert ddist.init() == (0, 1, 0)
    g, local = ddist.sample_sharded(_sample, base_seed=40)
    assert torch.equal(g, local) and ddist.rank_seed(40, 3) == 43


def test_bench_launch_plumbing_reaches_the_process_group(tmp_path):
    """bench.py under the driver's multi-GPU launch (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun): both ranks
    must reach init_process_group and agree on the world.  On this box there is no GPU, so the transport is forced to
    gloo and each rank then stops loudly at the engine (no CPU path) -- which is exactly the plumbing under test."""
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DUODIFF_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                                      cwd=str(REPO), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert f"process group initialised: rank {rank} of world 2, backend gloo" in se, se[-2000:]
        if not torch.cuda.is_available():
            assert p.returncode != 0 and "no CPU path" in se          # loud failure, no silent fallback


def test_bench_self_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun -- the shape of the driver's single-process command -- starts its two ranks
    itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for each child), waits for them and fails if a rank fails.  Without a
    GPU both children rendezvous over gloo and then stop loudly at the engine; the parent must report exactly that."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DUODIFF_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"], cwd=str(REPO),
                       capture_output=True, text=True, env=env, timeout=600)
    for rank in range(2):
        assert f"process group initialised: rank {rank} of world 2, backend gloo" in r.stderr, r.stderr[-3000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "ranks failed" in r.stderr and "no CPU path" in r.stderr
        assert not any(l.lstrip().startswith("{") for l in r.stdout.splitlines())   # no JSON line from a failed run
    # under torchrun (WORLD_SIZE set) a --gpus mismatch is still refused before anything else happens
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2"], cwd=str(REPO), capture_output=True, text=True,
                       env=env1, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
