"""Thin object layer over the C ABI: one Context per (process, GPU), Models built from a
reference-style state_dict.  torch tensors are containers for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .config import ModelParams

_ERR_CLASS = {
    L.DD_ERR_INVALID: ValueError,
    L.DD_ERR_NOT_FOUND: KeyError,
    L.DD_ERR_STATE: RuntimeError,
    L.DD_ERR_HIP: RuntimeError,
    L.DD_ERR_NOMEM: MemoryError,
    L.DD_ERR_UNSUPPORTED: NotImplementedError,
}

PRECISIONS = {"bf16": L.DD_PREC_BF16, "fp32": L.DD_PREC_FP32}


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """dd_ctx: bound to one device.  Creation fails loudly without a gfx950 GPU."""

    _per_device = {}

    def __init__(self, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.EngineUnavailable("no GPU visible: the DuoDiff engine runs only on MI355X (gfx950)")
        self.device = torch.cuda.current_device() if device is None else torch.device(device).index or 0
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.dd_ctx_create(int(self.device), C.byref(h))
        if rc != L.DD_OK:
            raise L.EngineUnavailable(f"dd_ctx_create(device={self.device}) failed with status {rc} "
                                      "(needs a gfx950 device and a working HIP runtime)")
        self.handle = h

    @classmethod
    def get(cls, device=None):
        idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
        if idx not in cls._per_device:
            cls._per_device[idx] = cls(idx)
        return cls._per_device[idx]

    def check(self, rc):
        if rc == L.DD_OK:
            return
        msg = self.lib.dd_last_error(self.handle)
        msg = msg.decode() if msg else f"status {rc}"
        raise _ERR_CLASS.get(rc, RuntimeError)(msg)

    def side_stream(self, device):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def sync(self, stream=None):
        self.check(self.lib.dd_sync(self.handle, _stream_ptr(stream)))

    def ddpm_step(self, x, eps, z, t, variance="beta_tilde", out=None, stream=None):
        """x' = postprocessing(eps, x, t) (reference sampler.py:47-56) on device tensors."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        var = L.DD_VAR_BETA if variance == "beta" else L.DD_VAR_BETA_TILDE
        self.check(self.lib.dd_ddpm_step(self.handle, _ptr(x), _ptr(eps.contiguous()),
                                         _ptr(z.contiguous() if z is not None else None), int(t), var,
                                         _ptr(out), x.numel(), _stream_ptr(stream)))
        return out

    def ddpm_step_coef(self, x, eps, z, c1, c2, sigma, out=None, stream=None):
        """x' = c1 * (x - c2 * eps) + sigma * z with caller-supplied scalars (any schedule; ddpm_core.py:190-193)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        self.check(self.lib.dd_ddpm_step_coef(self.handle, _ptr(x), _ptr(eps.contiguous()),
                                              _ptr(z.contiguous() if z is not None else None), float(c1), float(c2),
                                              float(sigma), _ptr(out), x.numel(), _stream_ptr(stream)))
        return out

    def to_images(self, x, out=None, stream=None):
        """(x + 1) / 2 as [B,H,W,C] (reference sampler.py:145-146), on the device."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[2] == x.shape[3]
        B, Cc, S, _ = x.shape
        out = torch.empty(B, S, S, Cc, device=x.device, dtype=torch.float32) if out is None else out
        self.check(self.lib.dd_to_images(self.handle, _ptr(x), _ptr(out), B, Cc, S, _stream_ptr(stream)))
        return out

    def early_exit_select(self, outputs, eps, classifier_outputs, threshold, stream=None):
        """eesampler.py:61-71 on device tensors -> (model_output [B,...], indices int32 [B], batch-mean errors [depth])."""
        depth, B = classifier_outputs.shape
        mo = torch.empty_like(eps)
        idx = torch.empty(B, device=eps.device, dtype=torch.int32)
        err = torch.empty(depth, device=eps.device, dtype=torch.float32)
        self.check(self.lib.dd_early_exit_select(self.handle, _ptr(outputs), _ptr(eps), _ptr(classifier_outputs),
                                                 float(threshold), depth, B, eps.numel() // max(B, 1), _ptr(mo), _ptr(idx),
                                                 _ptr(err), _stream_ptr(stream)))
        return mo, idx, err

    def affine_step(self, x, m, z, a, b, c, out=None, stream=None):
        """out = a*x + b*m + c*z on device tensors (z may be None)."""
        out = torch.empty_like(x) if out is None else out
        self.check(self.lib.dd_affine_step(self.handle, _ptr(x.contiguous()), _ptr(m.contiguous()),
                                           _ptr(z.contiguous() if z is not None else None), float(a), float(b),
                                           float(c), _ptr(out), x.numel(), _stream_ptr(stream)))
        return out

    def set_num_cus(self, n):
        """CU count this context's persistent GEMM grids are sized for (CU-masked streams)."""
        self.check(self.lib.dd_set_num_cus(self.handle, int(n)))

    def last_sample_timing(self):
        buf = (C.c_float * 3)()
        self.check(self.lib.dd_last_sample_timing(self.handle, buf))
        return tuple(buf)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.dd_ctx_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass


def schedule_tables():
    """The engine's own schedule tables (host arithmetic, usable without a GPU)."""
    lib = L.load()
    names = ["betas", "alphas", "alphas_bar", "alphas_bar_previous", "betas_tilde",
             "betas_tilde_scheduler", "c1", "c2", "sigma"]
    out = {}
    for i, n in enumerate(names):
        a = np.empty(1000, np.float32)
        rc = lib.dd_schedule_table(i, a.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != L.DD_OK:
            raise RuntimeError(f"dd_schedule_table({i}) -> {rc}")
        out[n] = a
    return out


def build_schedule(beta_init=1e-4, beta_final=0.02, beta_steps=1000):
    """NoiseScheduler.__init__ tables (ddpm_core.py:56-70) for any schedule, from the engine's host arithmetic."""
    lib = L.load()
    n = int(beta_steps)
    names = ["betas", "alphas", "alphas_bar", "alpha_bar_prev", "betas_tilde"]
    arrs = [np.empty(n, np.float32) for _ in names]
    rc = lib.dd_schedule_build(float(beta_init), float(beta_final), n, *[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
    if rc != L.DD_OK:
        raise ValueError(f"dd_schedule_build({beta_init}, {beta_final}, {beta_steps}) -> {rc}")
    return dict(zip(names, arrs))


class Model:
    """dd_model: U-ViT weights packed on the device + its activation workspace."""

    def __init__(self, ctx: Context, mp: ModelParams, max_batch: int):
        self.ctx, self.mp, self.max_batch = ctx, mp, int(max_batch)
        cfg = L.dd_config(mp.img_size, mp.patch_size, mp.in_chans, mp.embed_dim, mp.depth, mp.num_heads,
                          mp.mlp_ratio, mp.num_classes, int(mp.normalize_timesteps), self.max_batch,
                          int(mp.qkv_bias), int(mp.mlp_time_embed))
        h = C.c_void_p()
        ctx.check(ctx.lib.dd_model_create(ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        self.finalized = False
        self.precision = None

    def enable_early_exit(self, classifier_type="mlp_probe_per_layer"):
        kinds = {"mlp_probe_per_layer": L.DD_EE_MLP_PER_LAYER, "mlp_probe_per_timestep": L.DD_EE_MLP_PER_TIMESTEP,
                 "mlp_probe_per_layer_per_timestep": L.DD_EE_MLP_PER_LAYER_PER_TIMESTEP,
                 "attention_probe": L.DD_EE_ATTENTION_PROBE}
        if classifier_type not in kinds:
            raise ValueError(f"Unknown classifier type: {classifier_type}")
        self.ctx.check(self.ctx.lib.dd_model_enable_early_exit(self.handle, kinds[classifier_type]))

    def forward_early_exit(self, x, t, y=None, t_vec=None, stream=None):
        """(eps [B,C,S,S], classifier_outputs [depth,B], outputs [depth,B,C,S,S]) of EarlyExitUViT.forward."""
        B, depth = x.shape[0], self.mp.depth
        eps = torch.empty_like(x)
        cls = torch.empty(depth, B, device=x.device, dtype=torch.float32)
        outs = torch.empty((depth,) + tuple(x.shape), device=x.device, dtype=torch.float32)
        self.ctx.check(self.ctx.lib.dd_forward_early_exit(self.ctx.handle, self.handle, _ptr(x), float(t), _ptr(t_vec), _ptr(y),
                                                          _ptr(eps), _ptr(cls), _ptr(outs), B, _stream_ptr(stream)))
        return eps, cls, outs

    def set_param(self, name, tensor):
        t = tensor.detach().to("cpu", torch.float32).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        self.ctx.check(self.ctx.lib.dd_model_set_param(self.handle, name.encode(), C.c_void_p(t.data_ptr()),
                                                       shape, t.dim()))

    def finalize(self, precision="bf16"):
        with torch.cuda.device(self.ctx.device):
            self.ctx.check(self.ctx.lib.dd_model_finalize(self.handle, PRECISIONS[precision]))
        self.finalized, self.precision = True, precision

    def forward(self, x, t, y=None, out=None, t_vec=None, stream=None):
        """eps = model(x, t, y).  t: the common timestep; t_vec: optional [B] fp32 device tensor."""
        B = x.shape[0]
        out = torch.empty_like(x) if out is None else out
        self.ctx.check(self.ctx.lib.dd_forward(self.ctx.handle, self.handle, _ptr(x), float(t), _ptr(t_vec),
                                               _ptr(y), _ptr(out), B, _stream_ptr(stream)))
        return out

    def sample_step(self, x, t, y=None, z=None, noise="buffer", seed=0, variance="beta_tilde", eps_out=None,
                    stream=None):
        mode = {"none": L.DD_NOISE_NONE, "buffer": L.DD_NOISE_BUFFER, "philox": L.DD_NOISE_PHILOX}[noise]
        if mode == L.DD_NOISE_BUFFER and z is None:
            mode = L.DD_NOISE_NONE
        var = L.DD_VAR_BETA if variance == "beta" else L.DD_VAR_BETA_TILDE
        self.ctx.check(self.ctx.lib.dd_sample_step(self.ctx.handle, self.handle, _ptr(x), int(t), _ptr(y), mode,
                                                   _ptr(z), int(seed), var, _ptr(eps_out), x.shape[0],
                                                   _stream_ptr(stream)))
        return x

    PROFILE_KINDS = {"dominant": L.DD_PROF_DOMINANT, "block_tail": L.DD_PROF_BLOCK_TAIL, "fc1": L.DD_PROF_FC1, "rowlin": L.DD_PROF_ROWLIN,
                     "qkv_attention": L.DD_PROF_QKV_ATTENTION, "splitk": L.DD_PROF_SPLITK}

    def profile_steps(self, x, t_start=699, steps=10, y=None, stream=None, kind="dominant"):
        """Average ms per launch of one kernel family measured in context (event pairs around every such launch of `steps` eager steps);
        kind: "dominant" (the fused block tail where the model has one, else the fc1 GEMM) or one of PROFILE_KINDS."""
        ms, n = C.c_float(), C.c_int()
        self.ctx.check(self.ctx.lib.dd_profile_select(self.ctx.handle, self.PROFILE_KINDS[kind]))
        self.ctx.check(self.ctx.lib.dd_profile_steps(self.ctx.handle, self.handle, _ptr(x), _ptr(y), int(t_start), int(steps),
                                                     x.shape[0], _stream_ptr(stream), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_steps_chained(self, x, t_start=699, steps=10, y=None, stream=None, kind="dominant"):
        """The same with the batch split into dd_sample's two half-batch chains (the caller's stream + the context's side stream):
        average ms of a half-batch launch of the dominant kernel while the other chain runs beside it."""
        ms, n = C.c_float(), C.c_int()
        self.ctx.check(self.ctx.lib.dd_profile_select(self.ctx.handle, self.PROFILE_KINDS[kind]))
        self.ctx.check(self.ctx.lib.dd_profile_steps_chained(self.ctx.handle, self.handle, _ptr(x), _ptr(y), int(t_start), int(steps),
                                                             x.shape[0], _stream_ptr(stream), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def bench_gemm(self, B, iters=20, stream=None):
        ms, fl = C.c_float(), C.c_double()
        self.ctx.check(self.ctx.lib.dd_bench_gemm(self.ctx.handle, self.handle, int(B), int(iters),
                                                  _stream_ptr(stream), C.byref(ms), C.byref(fl)))
        return ms.value, fl.value

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.ctx.lib.dd_model_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass


def sample_loop(ctx: Context, first: Model, late, x, *, t_switch=0, t_start=999, t_end=0, y=None, seed=0,
                noise="philox", variance="beta_tilde", use_graph=True, stream=None):
    """dd_sample: the whole DDPM loop on the device (hipGraph replay per backbone), in place on x."""
    args = L.dd_sample_args()
    args.first = first.handle
    args.late = late.handle if late is not None else None
    args.t_switch = int(t_switch) if t_switch and np.isfinite(t_switch) else 0
    args.t_start, args.t_end = int(t_start), int(t_end)
    args.variance = L.DD_VAR_BETA if variance == "beta" else L.DD_VAR_BETA_TILDE
    args.noise_mode = {"none": L.DD_NOISE_NONE, "philox": L.DD_NOISE_PHILOX}[noise]
    args.use_graph = int(bool(use_graph))
    args.seed = int(seed)
    args.y_dev = y.data_ptr() if y is not None else None
    args.x_dev = x.data_ptr()
    args.B = x.shape[0]
    cur = stream if stream is not None else torch.cuda.current_stream(x.device)
    if use_graph and cur.cuda_stream == 0:
        # hipGraph capture is not permitted on the legacy default stream: run the loop on a private side stream, ordered
        # after the caller's work and before whatever the caller enqueues next
        side = ctx.side_stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ctx.check(ctx.lib.dd_sample(ctx.handle, C.byref(args), _stream_ptr(side)))
        cur.wait_stream(side)
    else:
        ctx.check(ctx.lib.dd_sample(ctx.handle, C.byref(args), _stream_ptr(cur)))
    return x


def sample_affine_loop(ctx: Context, first: Model, late, x, t, a, b, c, noise_flags, *, switch_after=None, y=None, seed=0,
                       counter_base=0, noise="philox", use_graph=True, stream=None):
    """dd_sample_affine: the table-driven loops (DDIM, predict_original / predict_previous) on the device, in place on x:
    x <- a[k] x + b[k] model(x, t[k]) + c[k] z for k = 0 .. len(t) - 1; the late model runs from step switch_after on.
    Step k draws z from Philox(key = seed, counter = counter_base + k): a loop cut into several calls passes the number
    of steps already done as counter_base and draws exactly the z of the uncut loop."""
    n = len(t)
    f32 = lambda v: np.ascontiguousarray(v, np.float32)
    tt, aa, bb, cc = f32(t), f32(a), f32(b), f32(c)
    nz = np.ascontiguousarray(noise_flags, np.int32)
    assert tt.shape == aa.shape == bb.shape == cc.shape == nz.shape == (n,)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    args = L.dd_affine_sample_args()
    args.first = first.handle
    args.late = late.handle if late is not None else None
    args.n_steps = n
    args.switch_after = n if (late is None or switch_after is None) else int(switch_after)
    args.t, args.a, args.b, args.c = (v.ctypes.data_as(fp) for v in (tt, aa, bb, cc))
    args.noise = nz.ctypes.data_as(ip)
    args.noise_mode = {"none": L.DD_NOISE_NONE, "philox": L.DD_NOISE_PHILOX}[noise]
    args.use_graph = int(bool(use_graph))
    args.seed = int(seed)
    args.counter_base = int(counter_base)
    args.y_dev = y.data_ptr() if y is not None else None
    args.x_dev = x.data_ptr()
    args.B = x.shape[0]
    cur = stream if stream is not None else torch.cuda.current_stream(x.device)
    if use_graph and cur.cuda_stream == 0:      # no capture on the legacy default stream (see sample_loop)
        side = ctx.side_stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ctx.check(ctx.lib.dd_sample_affine(ctx.handle, C.byref(args), _stream_ptr(side)))
        cur.wait_stream(side)
    else:
        ctx.check(ctx.lib.dd_sample_affine(ctx.handle, C.byref(args), _stream_ptr(cur)))
    return x


def sample_early_exit_loop(ctx: Context, model: Model, x, threshold, *, t_start=999, t_end=0, y=None, seed=0, noise="philox",
                           err=None, idx=None, use_graph=True, stream=None):
    """dd_sample_early_exit: the early-exit baseline's loop (eesampler.py:40-89) on the device, in place on x.
    err [1000, depth] fp32 / idx [1000, B] int32 device tensors (or None): rows t_start .. t_end are written."""
    args = L.dd_ee_sample_args()
    args.model = model.handle
    args.threshold = float(threshold)
    args.t_start, args.t_end = int(t_start), int(t_end)
    args.noise_mode = {"none": L.DD_NOISE_NONE, "philox": L.DD_NOISE_PHILOX}[noise]
    args.use_graph = int(bool(use_graph))
    args.B = x.shape[0]
    args.seed = int(seed)
    args.y_dev = y.data_ptr() if y is not None else None
    args.x_dev = x.data_ptr()
    if err is not None:
        assert err.is_cuda and err.dtype == torch.float32 and err.is_contiguous() and err.shape[0] == 1000
    if idx is not None:
        assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and tuple(idx.shape) == (1000, x.shape[0])
    args.err_dev = err.data_ptr() if err is not None else None
    args.idx_dev = idx.data_ptr() if idx is not None else None
    cur = stream if stream is not None else torch.cuda.current_stream(x.device)
    if use_graph and cur.cuda_stream == 0:      # no capture on the legacy default stream (see sample_loop)
        side = ctx.side_stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ctx.check(ctx.lib.dd_sample_early_exit(ctx.handle, C.byref(args), _stream_ptr(side)))
        cur.wait_stream(side)
    else:
        ctx.check(ctx.lib.dd_sample_early_exit(ctx.handle, C.byref(args), _stream_ptr(cur)))
    return x
