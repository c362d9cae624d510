"""Oracle (test infrastructure): KL-VAE decode, restated over torch-CPU functional ops.

Follows reference models/utils/autoencoder.py: nonlinearity :33-35, Normalize :38-41, Upsample :56-59,
ResnetBlock.forward :121-136, AttnBlock.forward :155-185, Decoder.forward :403-449, FrozenAutoencoderKL.decode
:486-490, for the fixed ddconfig of get_autoencoder :503-516.  Parameters: a dict keyed by the reference state_dict
names.  torch is used functionally (conv2d / group_norm / bmm are the ATen kernels the reference itself dispatches
to); pinned by tests/golden/vae_*.npz generated from the reference's own modules.
"""
import torch
import torch.nn.functional as F

CH, CH_MULT = 128, (1, 2, 4, 4)


def _swish(x):
    return x * torch.sigmoid(x)


def _gn(x, p, n):
    return F.group_norm(x, 32, p[n + ".weight"], p[n + ".bias"], eps=1e-6)


def _conv(x, p, n, pad):
    return F.conv2d(x, p[n + ".weight"], p[n + ".bias"], padding=pad)


def _resnet(x, p, n):
    h = _conv(_swish(_gn(x, p, n + ".norm1")), p, n + ".conv1", 1)
    h = _conv(_swish(_gn(h, p, n + ".norm2")), p, n + ".conv2", 1)
    if (n + ".nin_shortcut.weight") in p:
        x = _conv(x, p, n + ".nin_shortcut", 0)
    return x + h


def _attn(x, p, n):
    h_ = _gn(x, p, n + ".norm")
    q, k, v = (_conv(h_, p, f"{n}.{t}", 0) for t in ("q", "k", "v"))
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    h_ = torch.bmm(v.reshape(b, c, h * w), w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(h_, p, n + ".proj_out", 0)


@torch.no_grad()
def vae_decode(z, params, scale_factor=0.18215):
    p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in params.items()}
    z = torch.as_tensor(z, dtype=torch.float32)
    h = _conv((1.0 / scale_factor) * z, p, "post_quant_conv", 0)
    h = _conv(h, p, "decoder.conv_in", 1)
    h = _resnet(h, p, "decoder.mid.block_1")
    h = _attn(h, p, "decoder.mid.attn_1")
    h = _resnet(h, p, "decoder.mid.block_2")
    for lv in (3, 2, 1, 0):
        for j in range(3):
            h = _resnet(h, p, f"decoder.up.{lv}.block.{j}")
        if lv != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, p, f"decoder.up.{lv}.upsample.conv", 1)
    h = _conv(_swish(_gn(h, p, "decoder.norm_out")), p, "decoder.conv_out", 1)
    return h.numpy()
