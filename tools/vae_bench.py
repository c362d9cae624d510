"""Time dd_vae_decode on synthetic latents (ImageNet-256 latent config: 32x32x4 -> 256x256x3)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from duodiff_amd.autoencoder import FrozenAutoencoderKL, synthetic_vae_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ae = FrozenAutoencoderKL(synthetic_vae_state_dict(), precision=prec, max_chunk=chunk).to("cuda:0")
z = torch.randn(B, 4, 32, 32, device="cuda")
y = ae.decode(z); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    y = ae.decode(z)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
print(f"vae decode {prec} B={B} chunk={chunk}: {dt*1e3:.1f} ms  = {dt/B*1e3:.2f} ms/image, out std {float(y.std()):.3f}")
