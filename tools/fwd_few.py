"""Dev tool for rocprofv3: a few full-model forwards of one BASELINE workload's full backbone (bf16, its BASELINE batch).

    python tools/fwd_few.py [celeba|imagenet64|imagenet256] [dev_flags]
"""
import sys, torch
sys.path.insert(0, "/root/repo")
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.uvit import UViT
from duodiff_amd.weights import synthetic_state_dict
W = {"celeba": ("uvit_celeba", 128), "imagenet64": ("uvit_imagenet64", 256), "imagenet256": ("uvit_imagenet256", 32)}
name, B = W[sys.argv[1] if len(sys.argv) > 1 else "celeba"]
cfg = load_config(f"/root/repo/configs/{name}.yaml")
mp = ModelParams.from_dict(cfg)
if len(sys.argv) > 2 and int(sys.argv[2]):
    from duodiff_amd.engine import Context
    c0 = Context.get("cuda:0")
    c0.check(c0.lib.dd_dev_set_flags(c0.handle, int(sys.argv[2])))
m = UViT(**mp.as_dict(), precision="bf16", max_batch=B)
m.load_state_dict(synthetic_state_dict(mp, 1))
m.to("cuda")
x = torch.randn(B, mp.in_chans, mp.img_size, mp.img_size)
y = torch.randint(1, 1001, (B,)).clamp(max=mp.num_classes - 1) if mp.num_classes > 0 else None
for _ in range(3):
    e = m(x, torch.full((B,), 500.0), y)
torch.cuda.synchronize()
print("ok", float(e.abs().mean()))
