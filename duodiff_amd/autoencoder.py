"""Host-side mirror of the reference's frozen KL-VAE, decode side only (models/utils/autoencoder.py:452-516).

``get_autoencoder(path)`` / ``FrozenAutoencoderKL.decode(z)`` keep the reference's call surface
(``sampler.py:141-143,320-325``); the arithmetic runs in libduodiff.so (``dd_vae_*``): z/0.18215 -> post_quant_conv ->
Decoder (ResNet / attention / upsample stack) as im2col + MFMA GEMMs.  The encoder is training-only and not built.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib as L
from .engine import Context, PRECISIONS, _ptr, _stream_ptr

CH, CH_MULT, Z_CH = 128, (1, 2, 4, 4), 4   # ddconfig of reference get_autoencoder (autoencoder.py:503-516)


def vae_param_shapes() -> "OrderedDict[str, tuple]":
    """Decode-side tensors of the reference state_dict (post_quant_conv.* and decoder.*), name -> shape."""
    s = OrderedDict()

    def conv(n, co, ci, k):
        s[n + ".weight"] = (co, ci, k, k)
        s[n + ".bias"] = (co,)

    def norm(n, c):
        s[n + ".weight"] = (c,)
        s[n + ".bias"] = (c,)

    def res(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci, 3); norm(n + ".norm2", co); conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".nin_shortcut", co, ci, 1)

    conv("post_quant_conv", Z_CH, Z_CH, 1)
    top = CH * CH_MULT[-1]
    conv("decoder.conv_in", top, Z_CH, 3)
    res("decoder.mid.block_1", top, top)
    norm("decoder.mid.attn_1.norm", top)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"decoder.mid.attn_1.{n}", top, top, 1)
    res("decoder.mid.block_2", top, top)
    cin = top
    for lv in (3, 2, 1, 0):
        cout = CH * CH_MULT[lv]
        for j in range(3):
            res(f"decoder.up.{lv}.block.{j}", cin, cout)
            cin = cout
        if lv != 0:
            conv(f"decoder.up.{lv}.upsample.conv", cin, cin, 3)
    norm("decoder.norm_out", cin)
    conv("decoder.conv_out", 3, cin, 3)
    return s


def synthetic_vae_state_dict(seed: int = 4321) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 decode-side weights (no checkpoint exists offline): conv W ~ N(0, 1/fan_in), b ~ N(0, 0.02^2),
    GroupNorm gamma ~ 1 + N(0, 0.1^2), beta ~ N(0, 0.02^2)."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd = OrderedDict()
    for name, shp in vae_param_shapes().items():
        if name.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            t = torch.randn(shp, generator=g) / (fan_in ** 0.5)
        sd[name] = t.to(torch.float32).contiguous()
    return sd


class FrozenAutoencoderKL:
    """decode-only mirror; ``scale_factor`` is fixed to the reference's 0.18215 in the engine."""

    def __init__(self, state_dict=None, scale_factor=0.18215, precision="bf16", max_chunk=4, max_latent=32):
        if abs(scale_factor - 0.18215) > 1e-12:
            raise NotImplementedError("the engine's decode uses the reference scale_factor 0.18215")
        self.scale_factor, self.precision = scale_factor, precision
        self.max_chunk, self.max_latent = int(max_chunk), int(max_latent)
        self.embed_dim = 4
        self._state, self._handle, self._ctx, self._device = None, None, None, None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        want = vae_param_shapes()
        missing = [k for k in want if k not in state_dict]
        if strict and missing:
            raise RuntimeError(f"Error(s) in loading state_dict for FrozenAutoencoderKL: missing keys {missing}")
        sd = OrderedDict()
        for k, shp in want.items():
            t = torch.as_tensor(state_dict[k]).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shp)}")
            sd[k] = t.contiguous()
        self._state, self._handle = sd, None
        return [], []

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def to(self, device):
        self._device = torch.device(device)
        return self

    @property
    def device(self):
        return self._device or torch.device("cuda", torch.cuda.current_device())

    def _engine(self):
        if self._state is None:
            raise RuntimeError("FrozenAutoencoderKL has no weights: call load_state_dict first")
        if self._handle is None:
            ctx = Context.get(self.device)
            h = C.c_void_p()
            ctx.check(ctx.lib.dd_vae_create(ctx.handle, self.max_chunk, self.max_latent, C.byref(h)))
            for k, t in self._state.items():
                shape = (C.c_int64 * t.dim())(*t.shape)
                ctx.check(ctx.lib.dd_vae_set_param(h, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            with torch.cuda.device(ctx.device):
                ctx.check(ctx.lib.dd_vae_finalize(h, PRECISIONS[self.precision]))
            self._handle, self._ctx = h, ctx
        return self._ctx, self._handle

    def decode(self, z, stream=None):
        """reference autoencoder.py:486-490: [B,4,h,h] latents -> [B,3,8h,8h] images (fp32, on the device)."""
        ctx, h = self._engine()
        z = z.to(self.device, torch.float32).contiguous()
        B, c, hh, ww = z.shape
        if c != 4 or hh != ww:
            raise RuntimeError(f"expected latents [B,4,h,h], got {tuple(z.shape)}")
        out = torch.empty(B, 3, 8 * hh, 8 * ww, device=z.device, dtype=torch.float32)
        if B == 0:
            return out
        ctx.check(ctx.lib.dd_vae_decode(ctx.handle, h, _ptr(z), _ptr(out), B, hh, _stream_ptr(stream)))
        return out

    def __call__(self, inputs, fn="decode"):
        if fn != "decode":
            raise NotImplementedError("only decode is on the sampling path (encode is training-only)")
        return self.decode(inputs)

    def __del__(self):
        try:
            if self._handle is not None and self._handle.value:
                self._ctx.lib.dd_vae_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def get_autoencoder(pretrained_path, scale_factor=0.18215, precision="bf16"):
    """reference autoencoder.py:503-516: build the fixed-config KL-VAE and load its checkpoint (decode side)."""
    sd = torch.load(pretrained_path, map_location="cpu")
    return FrozenAutoencoderKL(sd, scale_factor, precision=precision)
