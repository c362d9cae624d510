"""Dev tool for rocprofv3: a few full-model forwards at the headline shapes (CelebA, B=128, bf16)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.uvit import UViT
from duodiff_amd.weights import synthetic_state_dict
cfg = load_config("/root/repo/configs/uvit_celeba.yaml")
mp = ModelParams.from_dict(cfg)
m = UViT(**mp.as_dict(), precision="bf16", max_batch=128)
m.load_state_dict(synthetic_state_dict(mp, 1))
m.to("cuda")
x = torch.randn(128, 3, 64, 64)
for _ in range(3):
    e = m(x, torch.full((128,), 500.0))
torch.cuda.synchronize()
print("ok", float(e.abs().mean()))
