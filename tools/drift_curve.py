#!/usr/bin/env python3
"""Free-running drift of the bf16 engine (parity reporting (3)).

CelebA DuoDiff pair (uvit_celeba_3 for t = 999..700, uvit_celeba for t = 699..0, t_switch = 300), B = 2, seeded synthetic
weights.  Three trajectories start from the same x_T and receive the SAME z at every step (torch CPU stream, as the
reference draws it): the bf16 engine, the fp32 engine (exact f32 MFMA) and the CPU oracle (fp32, torch-functional
restatement of the reference).  After K in {1, 10, 100, 300, 1000} steps: max-abs and rms differences, absolute and
normalised by rms(x) of the oracle trajectory (random weights are not a denoiser: |x| grows to O(100) by t = 0).

    python tools/drift_curve.py [--steps 1000] [--out gpurun_out/drift_curve.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import oracle  # noqa: E402  (the checker; this tool is test infrastructure, not product)
from duodiff_amd.config import ModelParams, load_config  # noqa: E402
from duodiff_amd.uvit import UViT  # noqa: E402
from duodiff_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "drift_curve.json"))
    ap.add_argument("--no_oracle", action="store_true")
    a = ap.parse_args()
    B, t_switch = 2, 300
    mp_s = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba_3.yaml"))
    mp_f = ModelParams.from_dict(load_config(REPO / "configs" / "uvit_celeba.yaml"))
    sd_s, sd_f = synthetic_state_dict(mp_s, 1237), synthetic_state_dict(mp_f, 1236)
    eng = {}
    for prec in ("bf16", "fp32"):
        eng[prec] = (UViT(**mp_s.as_dict(), precision=prec, max_batch=B).load_state_dict(sd_s).to("cuda").engine_model(B),
                     UViT(**mp_f.as_dict(), precision=prec, max_batch=B).load_state_dict(sd_f).to("cuda").engine_model(B))
    orc = None if a.no_oracle else (oracle.UViTTorchOracle(mp_s.as_dict(), sd_s), oracle.UViTTorchOracle(mp_f.as_dict(), sd_f))
    tables = oracle.sampler_schedule()
    torch.manual_seed(0)
    x_T = torch.randn(B, 3, 64, 64)
    xs = {p: x_T.cuda().contiguous() for p in eng}
    xo = x_T.numpy().copy()
    marks = [k for k in (1, 10, 100, 300, 1000) if k <= a.steps]
    rows = []
    t0 = time.time()
    for i in range(a.steps):
        t = 999 - i
        late = t < 1000 - t_switch                     # the late model takes over AFTER the step at t == 1000 - t_switch
        z = torch.randn(x_T.shape) if t > 0 else None
        for p, (es, ef) in eng.items():
            (ef if late else es).sample_step(xs[p], t, z=(z.cuda() if z is not None else None), noise="buffer")
        if orc is not None:
            eps = (orc[1] if late else orc[0])(xo, np.full((B,), t, np.float32))
            xo = oracle.ddpm_step(xo, eps, z.numpy() if z is not None else None, t, tables)
        if i + 1 in marks:
            torch.cuda.synchronize()
            xb, xf = xs["bf16"].cpu().numpy(), xs["fp32"].cpu().numpy()
            ref = xo if orc is not None else xf
            scale = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
            row = {"K": i + 1, "t_after": t, "rms_x": scale}
            pairs = [("bf16_vs_fp32", xb, xf)]
            if orc is not None:
                pairs += [("bf16_vs_oracle", xb, xo), ("fp32_vs_oracle", xf, xo)]
            for name, p, q in pairs:
                d = np.abs(p.astype(np.float64) - q)
                row[name] = {"max_abs": float(d.max()), "rms": float(np.sqrt((d ** 2).mean())),
                             "max_abs_over_rms_x": float(d.max() / scale), "rms_over_rms_x": float(np.sqrt((d ** 2).mean()) / scale)}
            rows.append(row)
            print(json.dumps(row), flush=True)
        if (i + 1) % 100 == 0:
            print(f"[drift {time.time() - t0:6.1f}s] step {i + 1}/{a.steps}", file=sys.stderr, flush=True)
    meets = {r["K"]: {k: v["max_abs"] <= 1e-3 for k, v in r.items() if isinstance(v, dict)} for r in rows}
    out = {"workload": "CelebA DuoDiff pair, B=2, t_switch=300, synthetic weights (seeds 1237 / 1236), x_T and z from torch CPU seed 0",
           "rows": rows, "max_abs_within_1e-3": meets}
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1))
    print("written", a.out, file=sys.stderr)


if __name__ == "__main__":
    main()
