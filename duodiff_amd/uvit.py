"""Host-side mirror of the reference's U-ViT noise predictor (models/uvit.py:228-387).

Same constructor keywords, ``load_state_dict`` / ``eval`` / ``to`` and call signature
``model(x, timesteps, y=None) -> eps`` as the reference, so ``sampler.py``-style code reads
unchanged; all arithmetic happens in libduodiff.so on the GPU.  Unknown constructor keys
(e.g. ``classifier_type`` carried by configs/uvit_imagenet64.yaml) are ignored, not a
TypeError as in the reference.
"""
from collections import OrderedDict

import torch

from .config import ModelParams
from .engine import Context, Model
from .weights import param_shapes


class UViT:
    def __init__(self, img_size, patch_size, in_chans, embed_dim, depth, num_heads, mlp_ratio=4, qkv_bias=False,
                 num_classes=-1, normalize_timesteps=True, mlp_time_embed=False, precision="bf16", max_batch=None,
                 **ignored):
        self.params = ModelParams.from_dict(dict(
            img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
            num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, mlp_time_embed=mlp_time_embed,
            num_classes=num_classes, normalize_timesteps=normalize_timesteps))
        # attribute names of the reference module
        self.embed_dim = self.num_features = embed_dim
        self.normalize_timesteps = bool(normalize_timesteps)
        self.num_classes, self.in_chans, self.depth = num_classes, in_chans, depth
        self.num_patches = self.params.num_patches
        self.extras = self.params.extras
        self.patch_dim = self.params.patch_dim
        self.precision = precision
        self._max_batch = max_batch
        self._state = None
        self._model = None
        self._device = None
        self.calls = 0

    # ---- reference nn.Module surface ---------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        """Accepts a bare state_dict or the trainer's {"model_state_dict": ...} (checkpointer.py:64-73)."""
        if "model_state_dict" in state_dict:
            state_dict = state_dict["model_state_dict"]
        want = param_shapes(self.params)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UViT: missing keys {missing}, "
                               f"unexpected keys {unexpected}")
        sd = OrderedDict()
        for k, shp in want.items():
            if k not in state_dict:
                continue
            t = torch.as_tensor(state_dict[k]).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(t.shape)} "
                                   f"from checkpoint, the shape in current model is {tuple(shp)}")
            sd[k] = t.contiguous()
        self._state = sd
        self._model = None
        return self

    def state_dict(self):
        return OrderedDict() if self._state is None else OrderedDict(self._state)

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("duodiff_amd implements the sampling (inference) path only")
        return self

    def to(self, device):
        self._device = torch.device(device)
        return self

    @property
    def device(self):
        return self._device or torch.device("cuda", torch.cuda.current_device())

    def parameters(self):
        class _P:  # enough for `next(model.parameters()).device` (reference ddpm_core.py:137)
            pass
        p = _P()
        p.device = self.device
        yield p

    # ---- engine ------------------------------------------------------------------------
    def engine_model(self, batch_size):
        """(Re)build the device-side model when first used or when the batch outgrows the workspace."""
        if self._state is None:
            raise RuntimeError("UViT has no weights: call load_state_dict first")
        need = max(int(batch_size), int(self._max_batch or 0))
        if self._model is None or self._model.max_batch < need:
            ctx = Context.get(self.device)
            m = Model(ctx, self.params, need)
            for k, v in self._state.items():
                m.set_param(k, v)
            m.finalize(self.precision)
            self._model = m
        return self._model

    def __call__(self, x, timesteps, y=None):
        """eps = model(x, time_tensor, y) (reference models/uvit.py:351-383)."""
        self.calls += 1
        dev = self.device
        x = x.to(dev, torch.float32).contiguous()
        B = x.shape[0]
        if tuple(x.shape[1:]) != (self.in_chans, self.params.img_size, self.params.img_size):
            raise RuntimeError(f"expected input [B,{self.in_chans},{self.params.img_size},{self.params.img_size}], "
                               f"got {tuple(x.shape)}")
        if self.num_classes > 0 and y is None:
            raise RuntimeError("class-conditional UViT called without y: token count does not match pos_embed")
        m = self.engine_model(B)
        t_vec = torch.as_tensor(timesteps).to(dev, torch.float32).reshape(-1).contiguous()
        if t_vec.numel() == 1:
            t_vec = t_vec.expand(B).contiguous()
        if t_vec.numel() != B:
            raise RuntimeError("timesteps must have one entry per batch row")
        if y is not None:
            y = torch.as_tensor(y).to(dev, torch.int64).contiguous()
            if y.numel() != B:
                raise RuntimeError("y must have one label per batch row")
            if int(y.min()) < 0 or int(y.max()) >= self.num_classes:
                raise IndexError("index out of range in self")  # nn.Embedding's failure class (quirk Q4)
        return m.forward(x, 0.0, y, t_vec=t_vec)  # per-row time_tensor, exactly as the reference passes it
