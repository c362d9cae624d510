"""Host-side mirror of the reference's early-exit baseline model (models/early_exit.py:193-324).

``EarlyExitUViT(uvit, classifier_type)`` wraps a :class:`duodiff_amd.uvit.UViT`, loads the reference's state_dict
(``uvit.*`` + ``matrix.*`` probes + per-layer output heads) and returns ``(eps, classifier_outputs, outputs)`` like the
reference's forward; the per-layer OutputHead (LayerNorm + Linear + unpatchify + 3x3 conv) and the uncertainty probe --
MLPProbe (Linear(D,1) + sigmoid + mean over tokens, three table layouts) or AttentionProbe (one learned query over the
tokens, then Linear + SiLU + Linear; the reference's default ``classifier_type``) -- run in libduodiff.so next to the backbone.
"""
from collections import OrderedDict

import torch

from .engine import Context, Model
from .uvit import UViT
from .weights import EE_CLASSIFIER_TYPES, ee_param_shapes


class EarlyExitUViT:
    def __init__(self, uvit: UViT, classifier_type="attention_probe", exit_threshold=0.2):
        if classifier_type not in EE_CLASSIFIER_TYPES:
            raise ValueError(f"Unknown classifier type: {classifier_type}")
        self.uvit, self.classifier_type, self.exit_threshold = uvit, classifier_type, exit_threshold
        self._state, self._model = None, None

    def load_state_dict(self, state_dict, strict=True):
        if "model_state_dict" in state_dict:
            state_dict = state_dict["model_state_dict"]
        want = ee_param_shapes(self.uvit.params, self.classifier_type)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for EarlyExitUViT: missing keys {missing[:8]}, "
                               f"unexpected keys {unexpected[:8]}")
        sd = OrderedDict()
        for k, shp in want.items():
            t = torch.as_tensor(state_dict[k]).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(t.shape)} "
                                   f"from checkpoint, the shape in current model is {tuple(shp)}")
            sd[k] = t.contiguous()
        self._state, self._model = sd, None
        return self

    def eval(self):
        return self

    def to(self, device):
        self.uvit.to(device)
        return self

    @property
    def device(self):
        return self.uvit.device

    def engine_model(self, batch_size):
        if self._state is None:
            raise RuntimeError("EarlyExitUViT has no weights: call load_state_dict first")
        need = max(int(batch_size), int(self.uvit._max_batch or 0))
        if self._model is None or self._model.max_batch < need:
            m = Model(Context.get(self.device), self.uvit.params, need)
            m.enable_early_exit(self.classifier_type)
            for k, v in self._state.items():
                m.set_param(k[len("uvit."):] if k.startswith("uvit.") else k, v)
            m.finalize(self.uvit.precision)
            self._model = m
        return self._model

    def forward_device(self, x, timesteps, y=None):
        """(eps, classifier_outputs [depth,B], outputs [depth,B,C,S,S]) as device tensors."""
        dev, u = self.device, self.uvit
        x = x.to(dev, torch.float32).contiguous()
        B = x.shape[0]
        if u.num_classes > 0 and y is None:
            raise RuntimeError("class-conditional UViT called without y: token count does not match pos_embed")
        t_vec = torch.as_tensor(timesteps).to(dev, torch.float32).reshape(-1).contiguous()
        if t_vec.numel() == 1:
            t_vec = t_vec.expand(B).contiguous()
        if t_vec.numel() != B:
            raise RuntimeError("timesteps must have one entry per batch row")
        if y is not None:
            y = torch.as_tensor(y).to(dev, torch.int64).contiguous()
            if int(y.min()) < 0 or int(y.max()) >= u.num_classes:
                raise IndexError("index out of range in self")
        t0 = int(torch.as_tensor(timesteps).reshape(-1)[0])                 # early_exit.py:271
        return self.engine_model(B).forward_early_exit(x, float(t0), y, t_vec=t_vec)

    def __call__(self, x, timesteps, y=None):
        """reference forward: (eps, [depth tensors of shape [B]], [depth tensors of shape [B,C,S,S]])."""
        eps, cls, outs = self.forward_device(x, timesteps, y)
        return eps, list(cls.unbind(0)), list(outs.unbind(0))
