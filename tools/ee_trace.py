"""Dev tool for rocprofv3: a few device-resident early-exit sampling steps (deediff_celeba, B = 128): python tools/ee_trace.py [steps]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd import eesampler
from duodiff_amd.config import ModelParams, load_config
from duodiff_amd.early_exit import EarlyExitUViT
from duodiff_amd.uvit import UViT
from duodiff_amd.weights import synthetic_ee_state_dict
B = 128
cfg = dict(load_config(REPO / "configs" / "deediff_celeba.yaml")["model_params"])
ctype = cfg.pop("classifier_type")
mp = ModelParams.from_dict(cfg)
ee = EarlyExitUViT(UViT(**mp.as_dict(), max_batch=B), ctype).load_state_dict(synthetic_ee_state_dict(mp, 77, ctype)).eval().to("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eesampler.get_samples(ee, B, 0, 3, 64, 64, 0.1, mp.depth, noise="device", num_steps=K)
torch.cuda.synchronize()
print("ok")
