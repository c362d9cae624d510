"""Oracle (test infrastructure): the early-exit baseline (DeeDiff / AdaDiff), numpy fp32.

Follows reference models/early_exit.py: OutputHead.forward :22-28, MLPProbe.forward :36-37, AttentionProbe :40-80,
EarlyExitUViT.get_classifer :194-204 and .forward :270-320; and eesampler.py get_samples :40-89 (the "simulated"
early exit: every layer runs, the per-sample output is gathered from the first layer whose predicted error is
<= threshold).  Pinned by tests/golden/ee_*.npz generated from the reference's own modules.
"""
import numpy as np
import torch

from .schedule_oracle import sampler_schedule
from .sampling_oracle import seed_everything
from .uvit_oracle import (F32, block_forward, conv3x3, layer_norm, linear, patch_embed, timestep_embedding, unpatchify)


def _sigmoid(v):
    return (F32(1) / (F32(1) + np.exp(-v, dtype=F32))).astype(F32)


class EarlyExitOracle:
    """params: the reference EarlyExitUViT state_dict (``uvit.*``, ``matrix.*``, ``*_heads.*``) as numpy arrays."""

    def __init__(self, cfg, params, classifier_type="mlp_probe_per_layer"):
        self.cfg = dict(cfg)
        self.p = {k: np.asarray(v, F32) for k, v in params.items()}
        self.u = {k[len("uvit."):]: v for k, v in self.p.items() if k.startswith("uvit.")}
        self.classifier_type = classifier_type
        self.depth, self.num_heads = cfg["depth"], cfg["num_heads"]
        self.embed_dim, self.in_chans = cfg["embed_dim"], cfg["in_chans"]
        self.extras = 2 if cfg.get("num_classes", -1) > 0 else 1
        self.normalize_timesteps = bool(cfg.get("normalize_timesteps", True))

    def _probe_key(self, t, i):                                   # early_exit.py:194-204
        return {"mlp_probe_per_layer": f"{i}", "mlp_probe_per_timestep": f"{t}", "attention_probe": f"{i}",
                "mlp_probe_per_layer_per_timestep": f"{i}, {t}"}[self.classifier_type]

    def _attention_probe(self, h, k):                             # :60-80, num_heads = 1
        p, D = self.p, self.embed_dim
        xs = h[:, 1:, :]                                          # :72 "ignore time vector": the FIRST token, whatever it is
        kv = linear(xs, p[f"matrix.{k}.weight_kv.weight"], p[f"matrix.{k}.weight_kv.bias"])
        kk, vv = kv[..., :D], kv[..., D:]                         # "b l (k h hd) -> k b h l hd", k = 2, h = 1
        q = p[f"matrix.{k}.q"].reshape(D)
        s = (kk @ q).astype(F32) * F32(1.0 / np.sqrt(D))          # F.scaled_dot_product_attention: scale 1/sqrt(head_dim)
        s = s - s.max(axis=1, keepdims=True)
        e = np.exp(s, dtype=F32)
        pr = (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
        o = np.einsum("bl,bld->bd", pr, vv).astype(F32)
        z = linear(o, p[f"matrix.{k}.classification.0.weight"], p[f"matrix.{k}.classification.0.bias"])
        z = (z * _sigmoid(z)).astype(F32)                         # SiLU
        return linear(z, p[f"matrix.{k}.classification.2.weight"], p[f"matrix.{k}.classification.2.bias"])[:, 0]

    def _probe(self, h, t, i):                                    # :36-37: sigmoid(Linear(D,1)).mean over ALL tokens
        k = self._probe_key(t, i)
        if self.classifier_type == "attention_probe":
            return self._attention_probe(h, k)
        v = linear(h, self.p[f"matrix.{k}.classifier.0.weight"], self.p[f"matrix.{k}.classifier.0.bias"])
        return _sigmoid(v)[..., 0].mean(axis=1, dtype=F32)

    def _head(self, h, prefix):                                   # :22-28
        p = self.p
        v = layer_norm(h, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
        v = linear(v, p[prefix + "decoder_pred.weight"], p[prefix + "decoder_pred.bias"])[:, self.extras:, :]
        return conv3x3(unpatchify(v, self.in_chans), p[prefix + "final_layer.weight"], p[prefix + "final_layer.bias"])

    def __call__(self, x, timesteps, y=None):
        u = self.u
        x = np.asarray(x, F32)
        ts = np.asarray(timesteps, F32)
        t = int(ts[0])                                                                    # :271
        if self.normalize_timesteps:
            ts = (ts / F32(1000)).astype(F32)
        tok = patch_embed(x, u["patch_embed.proj.weight"], u["patch_embed.proj.bias"])
        tok = np.concatenate([timestep_embedding(ts, self.embed_dim)[:, None, :], tok], axis=1)
        if y is not None and "label_emb.weight" in u:
            tok = np.concatenate([u["label_emb.weight"][np.asarray(y)][:, None, :], tok], axis=1)
        h = (tok + u["pos_embed"]).astype(F32)
        half = self.depth // 2
        cls, outs, skips = [], [], []
        for i in range(half):                                                             # :290-297
            outs.append(self._head(h, f"in_blocks_heads.{i}."))
            cls.append(self._probe(h, t, i))
            h = block_forward(h, u, f"in_blocks.{i}.", self.num_heads)
            skips.append(h)
        outs.append(self._head(h, "mid_block_head."))                                     # :299-302
        cls.append(self._probe(h, t, half))
        h = block_forward(h, u, "mid_block.", self.num_heads)
        for i in range(half):                                                             # :304-313
            outs.append(self._head(h, f"out_blocks_heads.{i}."))
            cls.append(self._probe(h, t, half + 1 + i))
            h = block_forward(h, u, f"out_blocks.{i}.", self.num_heads, skip=skips.pop())
        h = layer_norm(h, u["norm.weight"], u["norm.bias"])                               # :315-319
        h = linear(h, u["decoder_pred.weight"], u["decoder_pred.bias"])[:, self.extras:, :]
        eps = conv3x3(unpatchify(h, self.in_chans), u["final_layer.weight"], u["final_layer.bias"])
        return eps, cls, outs


def early_exit_select(eps, cls, outs, threshold):
    """eesampler.py:61-67: stack, append the final output with a zero error, first layer with error <= threshold
    (argmax of an all-False column is 0, as in torch)."""
    outputs = np.stack(list(outs) + [eps])
    c = np.stack(list(cls) + [np.zeros_like(cls[0])])
    idx = np.argmax((c <= F32(threshold)).astype(np.int32), axis=0)
    B = eps.shape[0]
    return outputs[idx, np.arange(B)], idx, c


def ee_get_samples(model, batch_size, seed, num_channels, sample_height, sample_width, threshold, depth, y=None,
                   autoencoder=None, num_steps=1000):
    """eesampler.py:40-89 (z from the torch CPU stream, which is what randn_like draws from on device=cpu)."""
    tb = sampler_schedule()
    seed_everything(seed)
    x = torch.randn(batch_size, num_channels, sample_height, sample_width).numpy()
    err = np.zeros((1000, depth), F32)
    ind = np.zeros((1000, batch_size), F32)
    for t in range(999, 999 - num_steps, -1):
        eps, cls, outs = model(x, (F32(t) * np.ones(batch_size, F32)).astype(F32), y)
        mo, idx, c = early_exit_select(eps, cls, outs, threshold)
        err[t] = c.mean(axis=1, dtype=F32)[:depth]
        ind[t, :] = idx
        a, ab, bt = tb["alphas"][t], tb["alphas_bar"][t], tb["betas_tilde"][t]
        z = torch.randn(x.shape).numpy() if t > 0 else F32(0)
        x = (np.sqrt(F32(1) / a) * (x - (F32(1) - a) / np.sqrt(F32(1) - ab) * mo) + np.sqrt(bt) * z).astype(F32)
    if autoencoder is not None:
        x = autoencoder(x)
    samples = ((x + F32(1)) / F32(2)).astype(F32).transpose(0, 2, 3, 1)
    return np.ascontiguousarray(samples), err, ind
