"""Dev tool: eesampler at CelebA size and the latent ImageNet-256 sampler (DDIM + KL-VAE decode), seeded synthetic checkpoints."""
import subprocess, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, "/root/repo")
import torch, yaml
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict, synthetic_ee_state_dict
from duodiff_amd.autoencoder import synthetic_vae_state_dict
R = Path("/root/repo")
tmp = Path(tempfile.mkdtemp())

def run(cmd):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=str(R), capture_output=True, text=True)
    print(" ".join(cmd[2:4]), "rc", r.returncode, "wall", round(time.time() - t0, 1), "s", r.stdout.strip().splitlines()[-1:] , r.stderr[-600:] if r.returncode else "")
    if r.returncode: sys.exit(1)

# early-exit baseline, CelebA size
cfg = load_config(R / "configs/deediff_celeba.yaml")
mp = ModelParams.from_dict(cfg)
torch.save(dict(synthetic_ee_state_dict(mp, 5, "mlp_probe_per_layer")), tmp / "ee.pth")
run([sys.executable, "-m", "duodiff_amd.eesampler", "--threshold", "0.45", "--checkpoint_path", str(tmp / "ee.pth"), "--batch_size", "64",
     "--output_folder", str(tmp / "ee_out"), "--config_path", str(R / "configs/deediff_celeba.yaml"), "--no_png", "--noise", "device"])
ind = torch.load(tmp / "ee_out/indices_by_timestep.pt")
print("exit histogram", torch.bincount(ind.flatten().long(), minlength=14).tolist())

# latent ImageNet-256: DDIM 50 steps, class-conditional, VAE decode
c = load_config(R / "configs/uvit_imagenet256.yaml")
mp = ModelParams.from_dict(c)
torch.save(dict(synthetic_state_dict(mp, 6)), tmp / "i256.pth")
torch.save(dict(synthetic_vae_state_dict()), tmp / "ae.pth")
run([sys.executable, "-m", "duodiff_amd.sampler", "--checkpoint_path", str(tmp / "i256.pth"), "--autoencoder_checkpoint_path", str(tmp / "ae.pth"),
     "--batch_size", "32", "--parametrization", "predict_noise", "--output_folder", str(tmp / "lat_out"),
     "--config_path", str(R / "configs/uvit_imagenet256.yaml"), "--class_id", "1", "--use_ddim", "--ddim_steps", "50", "--no_png"])
import numpy as np
s = np.load(tmp / "lat_out/samples.npy")
print("latent samples", s.shape, "finite", bool(np.isfinite(s).all()))
