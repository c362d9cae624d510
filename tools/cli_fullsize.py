"""Dev tool: the sampler CLI end to end at the headline size (CelebA DuoDiff, B=128, PNG output) on seeded synthetic checkpoints."""
import subprocess, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, "/root/repo")
import torch
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.weights import synthetic_state_dict
R = Path("/root/repo")
tmp = Path(tempfile.mkdtemp())
for name, seed in (("uvit_celeba_3", 1), ("uvit_celeba", 2)):
    mp = ModelParams.from_dict(load_config(R / "configs" / f"{name}.yaml"))
    torch.save({"model_state_dict": synthetic_state_dict(mp, seed)}, tmp / f"{name}.pth")
out = tmp / "out"
cmd = [sys.executable, "-m", "duodiff_amd.sampler", "--seed", "0", "--batch_size", "128", "--parametrization", "predict_noise",
       "--config_path", str(R / "configs/uvit_celeba_3.yaml"), "--checkpoint_path", str(tmp / "uvit_celeba_3.pth"),
       "--config_path_late", str(R / "configs/uvit_celeba.yaml"), "--checkpoint_path_late", str(tmp / "uvit_celeba.pth"),
       "--t_switch", "300", "--output_folder", str(out), "--noise", "device", "--timesteps_save", "500"]
t0 = time.time()
r = subprocess.run(cmd, cwd=str(R), capture_output=True, text=True)
print(r.stdout[-400:], r.stderr[-800:])
print("rc", r.returncode); sys.stdout.flush()
if r.returncode: sys.exit(1)
print("wall", round(time.time() - t0, 1), "s; files:", len(list(out.glob("*.png"))), (out / "statistics.txt").read_text().strip())
