#!/usr/bin/env python3
"""Headline benchmark: images/sec of 1000-step DDPM DuoDiff sampling on MI355X, bf16 MFMA operands, synthetic seeded
weights and noise.  Default workload = BASELINE.json's metric: CelebA 64x64 (uvit_celeba_3.yaml shallow for t=999..700 +
uvit_celeba.yaml full for t=699..0, t_switch=300), batch 128 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload celeba|imagenet64|imagenet256]

With --gpus N > 1 and no torchrun environment the script launches its N ranks ITSELF (child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before this process touches the GPU), relays rank 0's JSON line and exits
non-zero if any rank does; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one
rank of that launch.

A "step" is one sampling step (U-ViT forward + fused DDPM update) over the rank's batch.
K = 1000 (default) is one complete sampling run and `value` is then measured, not extrapolated.
For K < 1000 the K timed steps keep the 30 % shallow / 70 % full mix (switch after
round(0.3 K) steps) and value = images / (T_K * 1000 / K).
Every rank samples its own batch (seed + rank): weak scaling, no collective inside the loop;
one RCCL all_gather of the finished images closes the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.weights import synthetic_state_dict  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
def profile_file(name):
    """profiles/rNN/<name> of the newest round that holds it (tools/collect_profiles.sh writes one directory per round), or None."""
    for d in sorted((REPO / "profiles").glob("r[0-9][0-9]"), reverse=True):
        if (d / name).exists():
            return d / name
    return None


# kernel families the library can bracket with event pairs (include/duodiff.h DD_PROF_*): name fragment in rocprofv3's table -> kind
FAMILIES = (("mlp_fused_kernel", "block_tail"), ("qkv_attention_kernel", "qkv_attention"), ("rowlin768_kernel", "rowlin"),
            ("gemm256_kernel<1>", "fc1"), ("gemm256_kernel<5>", "splitk"))

# BASELINE.json configs[1], [3], [4]: (label, shallow yaml, full yaml, batch per GPU, CPU-baseline sample (images, steps))
WORKLOADS = {
    "celeba": ("CelebA-64", "uvit_celeba_3", "uvit_celeba", 128, (16, 30)),
    "imagenet64": ("ImageNet-64", "uvit_imagenet64_3", "uvit_imagenet64", 256, (8, 20)),
    "imagenet256": ("ImageNet-256 (32x32x4 latents)", "uvit_imagenet256_3", "uvit_imagenet256", 32, (8, 20)),
}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="celeba", choices=sorted(WORKLOADS))
    p.add_argument("--batch", type=int, default=0, help="images per GPU (default: the workload's BASELINE batch)")
    p.add_argument("--t_switch", type=int, default=300)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no_graph", action="store_true")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_batch", type=int, default=0)
    p.add_argument("--cpu_steps", type=int, default=0)
    p.add_argument("--launch_timeout", type=float, default=3000.0, help="--gpus N self-launch: seconds before the parent gives up and stops its ranks")
    p.add_argument("--num_cus", type=int, default=0, help="dd_set_num_cus: CU count the persistent GEMM grids are sized for (experiments)")
    p.add_argument("--dev_flags", type=int, default=0, help="dd_dev_set_flags value (include/duodiff_dev.h): same-process kernel-variant A/B runs only")
    return p.parse_args(argv)


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = min(len(os.sched_getaffinity(0)), os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(mp_s, mp_f, sd_s, sd_f, batch, steps, t_switch, y=None):
    """The oracle (CPU port of the reference path) on the host cores, on a bounded sample of the
    same workload: `batch` images, `steps` sampling steps with the 30/70 shallow/full mix."""
    import oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    # torch-functional oracle: the same ATen CPU kernels the reference's nn.Modules dispatch to
    m_s = oracle.UViTTorchOracle(mp_s.as_dict(), sd_s)
    m_f = oracle.UViTTorchOracle(mp_f.as_dict(), sd_f)
    n_shallow = max(1, round(steps * t_switch / 1000.0))
    tables = oracle.sampler_schedule()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, mp_f.in_chans, mp_f.img_size, mp_f.img_size, generator=g).numpy()
    t0 = time.perf_counter()
    for i in range(steps):
        t = 999 - i
        model = m_s if i < n_shallow else m_f
        eps = model(x, np.full((batch,), t, np.float32), y)
        z = torch.randn(x.shape, generator=g).numpy()
        x = oracle.ddpm_step(x, eps, z, t, tables)
    dt = time.perf_counter() - t0
    value = batch / (dt * 1000.0 / steps)
    return {"value": value, "unit": "images/sec", "cores": int(cores), "kind": "port",
            "sample": f"torch-CPU functional oracle (fp32, {cores} threads), {batch} images x {steps} steps ({n_shallow} shallow + {steps - n_shallow} full) "
                      f"in {dt:.1f} s, scaled to 1000 steps"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(a, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as children (one process per GPU, rendezvous on 127.0.0.1), before
    this process has made any GPU call; relay rank 0's stdout (the JSON line); fail if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(a.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None, text=True))
    log(f"launched {a.gpus} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    # rank 0's stdout is drained by a thread (a full pipe must not block it) while ALL children are polled: the first rank that
    # exits non-zero takes the others down with it (they would otherwise sit in rendezvous / all_reduce until the
    # torch.distributed timeout, holding their GPUs), and the whole launch is bounded by --launch_timeout
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + a.launch_timeout
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"ranks failed (rank, exit code): {bad}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = f"no result after {a.launch_timeout} s; still running: ranks {[r for r, c in enumerate(codes) if c is None]}"
            break
        time.sleep(0.2)
    if failed:
        for p in procs:                       # exactly the children started above, by pid
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    for line in ("".join(c or "" for c in chunks)).splitlines():   # rank 0's JSON line goes to stdout; anything a transport library printed there, to stderr
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)
    if failed:
        raise SystemExit(f"bench.py: {failed}")
    return 0


def committed_pmc(build_id, workload, kernel_substr):
    """Counter values of the dominant kernel from the committed rocprofv3 --pmc summaries of this workload -- only if they were
    collected on THIS build of the library (the summaries record dd_build_id()); otherwise ({}, reason).
    Returns {"traffic": HBM bytes per launch, "mfma_busy": MFMA-busy clocks / clocks of the launch, "sclk_mhz": clock held under it}."""
    sfx = "" if workload == "celeba" else f"_{workload}"
    try:
        pmc = json.load(open(profile_file(f"pmc_traffic{sfx}.json")))
        sq = json.load(open(profile_file(f"pmc_sq{sfx}.json")))
    except Exception as e:
        return {}, f"no committed PMC profile of {workload} under profiles/ ({type(e).__name__})"
    have = pmc.get("_build_id"), sq.get("_build_id")
    if have[0] != build_id or have[1] != build_id:
        return {}, (f"committed PMC profile is of another build ({have[0]} / {have[1]}, running {build_id}): counters not quoted")
    # the dominant kernel may run as several instantiations (the block tail with / without the next block's skip_linear phases): the
    # timed launches are a mix of them, so the counters are averaged over the profile's launches of all of them
    ks = [v for n, v in pmc.items() if kernel_substr in n and isinstance(v, dict)]
    calls = sum(v["calls"] for v in ks)
    if not ks or not calls:
        return {}, f"committed PMC profile holds no {kernel_substr}"
    out = {"traffic": sum((v["fetch_MB_corrected"] + v["write_MB"]) * v["calls"] for v in ks) / calls * 1e6}
    bs = [v for n, v in sq.items() if kernel_substr in n and isinstance(v, dict) and v.get("gui_active_clk")]
    if bs:   # launch-level: MFMA-busy clocks per SIMD / clocks the launch took (GRBM_GUI_ACTIVE / 8)
        out["mfma_busy"] = sum(v["mfma_clk_per_simd"] for v in bs) / sum(v["gui_active_clk"] for v in bs)
        out["sclk_mhz"] = sum(v["gui_active_clk"] for v in bs) / sum(v["avg_us_under_pmc"] for v in bs)
    return out, None


def kernel_shares(workload):
    """[(kind, name fragment, share of the GPU kernel time)] of the timeable kernel families, largest first, from the committed
    rocprofv3 --kernel-trace --stats table of this workload's bench run (instantiations of one kernel are summed); [] if absent."""
    import csv
    sfx = "" if workload == "celeba" else f"_{workload}"
    try:
        rows = list(csv.DictReader(open(profile_file(f"kernel_stats{sfx}.csv"))))
    except Exception:
        return []
    total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    fam = {}
    for frag, kind in FAMILIES:
        t = sum(float(r["TotalDurationNs"]) for r in rows if frag in r["Name"])
        if t > 0:
            fam[kind] = (frag, t / total)
    return sorted(((k, f, sh) for k, (f, sh) in fam.items()), key=lambda e: -e[2])


def family_model(kind, mp, B, dev_flags=0):
    """Algorithmic FLOPs and HBM bytes per launch of one kernel family of the FULL backbone at batch B, averaged over the launches of one
    forward (a family's launches differ: the block tail with / without the next skip_linear; attn.proj, mlp.fc2 and skip_linear share
    one row-resident or split-K kernel), the rocprofv3 name fragment for the PMC lookup, and a description."""
    D, Hd, L, N, depth, heads = mp.embed_dim, 4 * mp.embed_dim, mp.seq_len, mp.num_patches, mp.depth, mp.num_heads
    M, Mp, nskip = B * L, B * N, mp.depth // 2
    if kind == "block_tail":
        # fc1 + fc2 of every row (the extra-token rows run in the launch's hidden-split workgroups) + attn.proj of the patch rows
        fl = 2.0 * M * D * Hd * 2 + 2.0 * Mp * D * D
        # fp32 residual rows read once and written once, the bf16 attention output read once, bf16 copy for the long skip, bf16 norm1
        # output for the next block, the bf16 weights once
        by = M * D * (4 + 4 + 2 + 2 + 2) + (2 * D * Hd + D * D) * 2
        name = ("mlp_fused_kernel<%d>: attn.proj + residual + norm2 + fc1 + GELU + fc2 + residual + next norm1, M=%d D=%d hidden=%d "
                "(the small proj_rows / mlp_reduce launches of the extra-token rows are outside the event pair)" % (D, M, D, Hd))
        # depth // 2 of the depth launches per step also run the NEXT block's skip_linear on the patch rows (cat[y, skip] . Wskip^T:
        # 2 M 2D D flops; its long-skip rows replace the bf16 copy in the byte count, + Wskip): the timed launches are that mix
        ns = nskip if (D % 128 == 0 and not (dev_flags & 32)) else 0
        if ns:
            fl += ns / depth * 2.0 * Mp * 2 * D * D
            by += int(ns / depth * 2 * D * D * 2)
            name += "; %d of %d launches per step + the next block's skip_linear (flops and bytes averaged over the mix)" % (ns, depth)
        return fl, by, "mlp_fused_kernel", name, depth
    if kind == "fc1":
        return (2.0 * M * D * Hd, M * D * 2 + M * Hd * 2 + D * Hd * 2, "gemm256_kernel<1>",
                "gemm256_kernel<EPI_BIAS_GELU>: fc1 + bias + exact-erf GELU, M=%d K=%d N=%d" % (M, D, Hd), depth)
    if kind == "qkv_attention":
        fl = 2.0 * M * D * 3 * D + 4.0 * B * heads * L * L * 64
        by = M * D * 2 * 2 + 3 * D * D * 2
        return fl, by, "qkv_attention_kernel", ("qkv_attention_kernel<%d>: attn.qkv Linear + softmax(q k^T / 8) v per (image, head), norm1 rows in, "
                                                 "attention rows out (the qkv tensor is never written), B=%d L=%d heads=%d" % (D, B, L, heads)), depth
    if kind in ("rowlin", "splitk"):
        # attn.proj (K = D) and mlp.fc2 (K = 4 D) of every block + skip_linear (K = 2 D) of the out-blocks: one kernel, three shapes
        n = 2 * depth + nskip
        fl = (depth * 2.0 * M * D * D + depth * 2.0 * M * D * Hd + nskip * 2.0 * M * 2 * D * D) / n
        if kind == "rowlin":   # operand rows in (bf16), fp32 residual rows in + out (skip_linear: out only), LayerNorm rows out (bf16), mlp.fc2: bf16 copy, weights
            by = (depth * (M * D * 2 + M * D * 8 + M * D * 2 + D * D * 2) + depth * (M * Hd * 2 + M * D * 8 + M * D * 4 + D * Hd * 2)
                  + nskip * (M * 2 * D * 2 + M * D * 4 + M * D * 2 + 2 * D * D * 2)) / n
            return fl, int(by), "rowlin768_kernel", ("rowlin768_kernel: x += A . W^T + b with the rows resident in registers + the LayerNorm behind it -- attn.proj, "
                                                     "mlp.fc2 and skip_linear launches of a forward (%d + %d + %d), flops and bytes averaged over the mix, M=%d" % (depth, depth, nskip, M)), n
        by = (depth * (M * D * 2 + D * D * 2) + depth * (M * Hd * 2 + D * Hd * 2) + nskip * (M * 2 * D * 2 + 2 * D * D * 2)) / n + 2 * M * D * 4   # operand + weights in, two fp32 slabs out
        return fl, int(by), "gemm256_kernel<5>", ("gemm256_kernel<EPI_PARTIAL>: split-K halves of attn.proj / mlp.fc2 / skip_linear into fp32 slabs (%d + %d + %d launches of a "
                                                  "forward; the row pass reduce_ln_kernel behind each is outside the event pair), M=%d" % (depth, depth, nskip, M)), n
    raise ValueError(kind)


def cu_share_half_batch(kind, mp, B):
    """Share of the 256 CUs a HALF-batch launch of this family holds in the chained loop (one workgroup per CU for all of them)."""
    Bh = B // 2
    if kind == "block_tail" or kind == "rowlin":
        return min(1.0, ((Bh * mp.num_patches + 127) // 128) / 256.0)
    if kind == "qkv_attention":
        return min(1.0, Bh * mp.num_heads / 256.0)
    # persistent GEMM grids: both chains' grids are sized for half the CUs when the batch is large (capi.hip chain_gemm_cus)
    if B * mp.seq_len > 32768:
        return 0.5
    col_tiles = 4 * mp.embed_dim // 256 if kind == "fc1" else 2 * (mp.embed_dim // 256)     # fc1: N = 4 D; split-K: N = D, two k halves per tile
    return min(1.0, max(1, Bh * mp.seq_len // 256) * col_tiles / 256.0)


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return self_launch(a, argv)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         "or run bench.py --gpus N without a torchrun environment (it then launches its ranks itself)")
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    no_device = (f"bench.py rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible on this node -- one process per GPU, "
                 f"--gpus N needs N devices")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DUODIFF_DIST_BACKEND", "nccl")     # gloo: rehearsal / CPU test of the launch plumbing
        if backend == "nccl":
            if ndev <= local_rank:
                raise SystemExit(no_device)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        log(f"process group initialised: rank {rank} of world {world}, backend {backend}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path (duodiff_amd._lib.EngineUnavailable)")
    if ndev <= local_rank:
        raise SystemExit(no_device)
    torch.cuda.set_device(local_rank)

    from duodiff_amd import _lib, sampler
    from duodiff_amd.engine import sample_loop
    from duodiff_amd.uvit import UViT

    label, cfg_s, cfg_f, batch_default, cpu_sample = WORKLOADS[a.workload]
    B = a.batch or batch_default
    mp_s = ModelParams.from_dict(load_config(REPO / "configs" / f"{cfg_s}.yaml"))
    mp_f = ModelParams.from_dict(load_config(REPO / "configs" / f"{cfg_f}.yaml"))
    sd_s, sd_f = synthetic_state_dict(mp_s, 1237), synthetic_state_dict(mp_f, 1236)
    dev = f"cuda:{local_rank}"
    if a.dev_flags:
        from duodiff_amd.engine import Context
        c0 = Context.get(dev)
        c0.check(c0.lib.dd_dev_set_flags(c0.handle, a.dev_flags))
    if a.num_cus:
        from duodiff_amd.engine import Context
        Context.get(dev).set_num_cus(a.num_cus)
    shallow = UViT(**mp_s.as_dict(), precision=a.precision, max_batch=B).load_state_dict(sd_s).to(dev)
    full = UViT(**mp_f.as_dict(), precision=a.precision, max_batch=B).load_state_dict(sd_f).to(dev)
    es, ef = shallow.engine_model(B), full.engine_model(B)
    ctx = es.ctx
    build_id = _lib.load().dd_build_id().decode()

    K, W = a.steps, a.warmup
    log(f"{label}: models ready on {dev}; B={B} K={K} W={W}; library build {build_id}")
    seed = 0 + rank
    sampler.seed_everything(seed)
    x_T = torch.randn(B, mp_f.in_chans, mp_f.img_size, mp_f.img_size).to(dev).contiguous()
    y = None
    if mp_f.num_classes > 0:      # class-conditional workloads: labels as the reference CLI draws them (sampler.py:314-318)
        y = torch.randint(1, 1001, (B,)).clamp(max=mp_f.num_classes - 1).to(dev)
    stream = torch.cuda.Stream(device=dev)
    use_graph = not a.no_graph
    # K timed steps: t = 999 .. 1000-K, switch after round(t_switch * K / 1000) steps
    k_switch = a.t_switch if K == 1000 else max(1, round(a.t_switch * K / 1000.0))
    t_end = 1000 - K

    def run(x, n_steps, t_sw, t_stop):
        with torch.cuda.stream(stream):
            sample_loop(ctx, es, ef, x, t_switch=t_sw, t_start=999, t_end=t_stop, y=y, seed=seed, noise="philox",
                        use_graph=use_graph, stream=stream)

    # One timed pass = K sampling steps + the output conversion (x+1)/2 -> NHWC (library kernel, reference
    # sampler.py:145-146) + for N > 1 the single RCCL all_gather of the finished images.  Nothing else runs between
    # t0 and dt: no torch op, no allocation (every buffer below is created before the warm-up).
    # (ImageNet-256: x is the 32x32x4 latent; the reference decodes it once per run with the KL-VAE -- 2.8 ms / image on this
    # engine, 1.4 % of a 1000-step latent, DESIGN section 7 -- outside the per-step metric.)
    x = x_T.clone()  # one persistent state buffer: the captured graphs bake its address in
    imgs = torch.empty(B, mp_f.img_size, mp_f.img_size, mp_f.in_chans, device=dev)
    gathered = [torch.empty_like(imgs) for _ in range(world)] if world > 1 else None
    stream.wait_stream(torch.cuda.current_stream())   # x_T.clone() / allocations above ran on the default stream

    ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_pass(n_steps, t_sw, t_stop):
        run(x, n_steps, t_sw, t_stop)
        ctx.to_images(x, out=imgs, stream=stream)
        if dist is not None:
            with torch.cuda.stream(stream):
                if dist.get_backend() == "nccl":
                    ev_g0.record()
                    dist.all_gather(gathered, imgs)              # the single collective: final images (RCCL over xGMI)
                    ev_g1.record()
                else:                                            # gloo rehearsal: host memory
                    stream.synchronize()
                    host = [torch.empty(imgs.shape) for _ in range(world)]
                    dist.all_gather(host, imgs.cpu())

    # warm-up: W untimed steps touching both backbones (captures the graphs) followed by the EXACT tail of the timed
    # pass (output kernel, all_gather), so that no first-use cost (code-object load, graph upload, communicator
    # set-up) is left for the timed window even at K = 20
    if W > 0:
        one_pass(W, max(1, W // 2), 1000 - W)
        stream.synchronize()
        with torch.cuda.stream(stream):
            x.copy_(x_T, non_blocking=True)
        stream.synchronize()
        log("warmup done")

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one_pass(K, k_switch, t_end)
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = ctx.last_sample_timing()   # hipEvents around the K steps of this rank's dd_sample call
    chains = int(ctx.lib.dd_dev_last_sample_chains(ctx.handle))   # 2: the batch ran as two half-batch chains on two streams
    on_dev = dist is None or dist.get_backend() == "nccl"
    gather_ms = ev_g0.elapsed_time(ev_g1) if (dist is not None and dist.get_backend() == "nccl") else 0.0
    # MAX over ranks of (wall, GPU ms of the K steps, all_gather ms, -GPU ms): the last entry gives the MIN, so that a slow or
    # late rank shows in the one line rank 0 prints
    t_all = torch.tensor([dt, timing[0], gather_ms, -timing[0], -float(chains)], device=dev if on_dev else "cpu", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt, gpu_ms_max, gather_ms_max, gpu_ms_min = float(t_all[0].item()), float(t_all[1].item()), float(t_all[2].item()), -float(t_all[3].item())
    chains_min = int(round(-float(t_all[4].item())))     # a rank that fell back to one chain (odd batch, flag) shows here
    log(f"timed region done: {dt:.3f} s")
    finite = bool(torch.isfinite(imgs).all().item())

    if rank == 0:
        images = B * world
        value = images / (dt * 1000.0 / K)
        flop_img = 0.3 * mp_s.flops_per_image() * 1000 + 0.7 * mp_f.flops_per_image() * 1000
        e2e_tflops = value * flop_img / 1e12 / world
        # Dominant kernel of the full model's blocks, timed live IN CONTEXT: hipEvent pairs on the launch stream around every
        # such launch of 20 eager full-model steps run right after the timed region (same buffers, cache and clock state).
        #   D <= 512: the fused block tail (mlp_fused_kernel: attn.proj + residual + norm2 + fc1 + bias + exact-erf GELU + fc2 +
        #             bias + residual + next norm1 of the PATCH rows; the extra-token rows run in small launches outside the pair)
        #   else:     the fc1 GEMM (bias + GELU epilogue) of the two-GEMM path
        #   Which kernel: the one with the largest total time in the committed rocprofv3 kernel table of this workload's bench run
        #   (profiles/rNN/kernel_stats*.csv); without a table, the design default (the fused block tail for embed_dim <= 512, else fc1).
        D_ = mp_f.embed_dim
        fused = a.precision == "bf16" and D_ in (64, 128, 256, 512)
        shares = kernel_shares(a.workload) if a.precision == "bf16" else []
        kind = shares[0][0] if shares else ("block_tail" if fused else "fc1")
        with torch.cuda.stream(stream):
            ms, n_launch = ef.profile_steps(x, t_start=699, steps=20, y=y, stream=stream, kind=kind)
            # ... and the same launches as the timed loop runs them when it splits the batch: two half-batch chains side by side
            ms_ch, n_ch = ef.profile_steps_chained(x, t_start=699, steps=20, y=y, stream=stream, kind=kind) if chains == 2 else (None, 0)
            # the next two kernels of the table, timed the same way (5 steps each): name, share of the GPU kernel time, fraction of the roof
            top = []
            for k2, frag, sh in shares[:3]:
                m2, n2 = (ms, n_launch) if k2 == kind else ef.profile_steps(x, t_start=699, steps=5, y=y, stream=stream, kind=k2)
                f2 = family_model(k2, mp_f, B, a.dev_flags)[0]
                top.append({"name": frag, "share_of_kernel_time": sh, "ms_per_launch": m2, "launches_timed": n2, "flops_per_launch": f2,
                            "frac": (f2 / (m2 * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS) if m2 else None})
        fl, alg_bytes, pmc_kernel, kname, _ = family_model(kind, mp_f, B, a.dev_flags)
        ach = fl / (ms * 1e-3) / 1e12
        cu_share = cu_share_half_batch(kind, mp_f, B)     # CUs a half-batch launch of this kernel holds in the chained loop
        # HBM bytes / MFMA-busy fraction of that kernel: quoted from the COMMITTED rocprofv3 --pmc profile only when that profile
        # was collected on the build that is running (rocprofv3 cannot run inside this process); null + reason otherwise
        pm, why_not = committed_pmc(build_id, a.workload, pmc_kernel)
        sfx_ = "" if a.workload == "celeba" else f"_{a.workload}"
        rel = lambda f: str(f.relative_to(REPO)) if f else "profiles/"
        pmc_rel, stats_rel = rel(profile_file(f"pmc_traffic{sfx_}.json")), rel(profile_file(f"kernel_stats{sfx_}.csv"))
        traffic, mfma_busy, sclk = pm.get("traffic"), pm.get("mfma_busy"), pm.get("sclk_mhz")
        log(f"dominant kernel: {ms * 1e3:.1f} us = {ach:.0f} TFLOP/s")
        out = {
            "metric": "images/sec (whole node) DuoDiff 1000-step %s" % label,
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1000.0 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "%s DuoDiff: %s (t=999..700) + %s (t=699..0), t_switch=300, 1000-step DDPM, batch %d/GPU, "
                                   "device Philox noise" % (label, cfg_s, cfg_f, B),
                       "batch_per_gpu": B, "t_switch": a.t_switch, "hipgraph": use_graph,
                       # dd_sample runs an even batch >= 32 as two independent half-batch chains on two streams (bit-identical results:
                       # one chain's HBM-bound phases overlap the other's MFMA phases); the roofline leg below times the kernel ALONE
                       # on the chip at the full batch, one chain
                       "chains_in_timed_region": chains, "chains_min_over_ranks": chains_min,
                       "timed_steps": K, "switch_after_steps": k_switch,
                       "seconds_per_sample": (dt * 1000.0 / K) / images, "finite": finite,
                       "gpu_ms_total": timing[0], "gpu_ms_first_backbone": timing[1], "gpu_ms_late_backbone": timing[2],
                       # wall time of the timed region (MAX over ranks) minus the slowest rank's GPU time of the K steps: output kernel,
                       # the all_gather, barriers, rank skew and host launch overhead together
                       "non_step_ms": dt * 1000.0 - gpu_ms_max, "gpu_ms_total_max_over_ranks": gpu_ms_max, "gpu_ms_total_min_over_ranks": gpu_ms_min,
                       "all_gather_ms_max_over_ranks": gather_ms_max if world > 1 else None, "dev_flags": a.dev_flags,
                       "library_build_id": build_id},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": (f"bytes/launch from the committed rocprofv3 --pmc profile {pmc_rel}, "
                                            f"collected on this build ({build_id}); not measured in this run") if traffic else why_not,
                         "algorithmic_bytes": alg_bytes,
                         "kernel": kname,
                         "kernel_chosen_from": (f"largest total time in {stats_rel}" if shares else "design default (no committed kernel table for this workload / precision)"),
                         "kernels": top,
                         "ms_per_launch": ms, "ms_per_launch_source": "measured live (hipEvents on the launch stream)",
                         "launches_timed": n_launch, "flops_per_launch": fl,
                         "end_to_end_tflops_per_gpu": e2e_tflops, "end_to_end_frac": e2e_tflops / BF16_MFMA_PEAK_TFLOPS,
                         "hbm_GBps": (traffic / (ms * 1e-3) / 1e9) if traffic else None, "hbm_peak_GBps": HBM_PEAK_GBS,
                         "mfma_busy_frac_pmc": mfma_busy,
                         "mfma_busy_source": (f"SQ_VALU_MFMA_BUSY_CYCLES per SIMD / (GRBM_GUI_ACTIVE / 8) of the launch, committed profile {pmc_rel} (pmc_sq*.json) "
                                              f"of this build; not measured in this run") if mfma_busy else why_not,
                         "sclk_mhz_under_load": sclk,
                         "sclk_source": ("GRBM_GUI_ACTIVE / 8 / launch duration of the same committed profile (profiled passes clock 2-5 % below un-profiled ones)") if sclk else why_not,
                         # the timed loop's own launches of this kernel (chains_in_timed_region == 2): half the batch per launch, the other
                         # chain's kernels running beside it.  A launch of T main tiles holds T of the 256 CUs (one workgroup per CU: LDS), so
                         # its own roof is peak x T / 256; `frac` above is the conservative figure (the kernel alone, every CU in the same phase)
                         "chained": None if ms_ch is None else {
                             "ms_per_launch": ms_ch, "launches_timed": n_ch, "flops_per_launch": fl / 2, "achieved": fl / 2 / (ms_ch * 1e-3) / 1e12,
                             "cu_share": cu_share,
                             "frac_of_the_cus_it_holds": fl / 2 / (ms_ch * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS * cu_share),
                             "ms_per_launch_source": "measured live (hipEvents on both chains' streams, eager steps)"},
                         # the block tail is power-limited: constants of the committed in-kernel stamp run (a diagnostic variant build; not measured here)
                         "power_limit": None if kind != "block_tail" else {
                             "in_kernel_clock_mhz": {"all_cus_random_operands": 1680, "all_cus_zero_operands": 2380, "half_the_cus_random_operands": 2370},
                             "cycles_per_workgroup": 255000, "chunk_loop_cycles_per_chunk": 2244, "mfma_cycles_per_chunk": 2048,
                             "chunk_loop_frac_of_the_pipe_rate_in_cycles": 2048.0 / 2244.0,
                             "us_per_launch_zero_operands_same_cycles": 113.0,
                             "source": "profiles/r05/power_probe.txt, block_tail_energy_ablations.txt (tools/power_probe.py on the stamped variant of mlp_fused_kernel<512, LN, PROJ>; "
                                       "s_memtime / s_memrealtime per workgroup): the launch's cycles are worth 0.55 of the roof at the nominal clock; on random operands with "
                                       "all 256 CUs in the kernel the chip holds 1.66-1.69 GHz"},
                         "sustained_mfma_tflops_random_operands": 1910.0, "frac_of_sustained": ach / 1910.0,
                         "sustained_note": "constant, not measured in this run: register-only v_mfma_f32_32x32x16_bf16 loop, random operands, "
                                           "measured on MI355X (tools/mfma_peak.hip, profiles/r01/mfma_peak.txt); 2470 with constant operands"},
        }
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline ...")
            cb, cs = a.cpu_batch or cpu_sample[0], a.cpu_steps or cpu_sample[1]
            ycpu = y[:cb].cpu().numpy() if y is not None else None
            out["cpu_baseline"] = cpu_baseline(mp_s, mp_f, sd_s, sd_f, cb, cs, a.t_switch, ycpu)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
