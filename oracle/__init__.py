"""CPU oracle for the DuoDiff sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy/fp32 restatement of the
reference algorithm (razvanmatisan/duodiff: sampler.py, ddpm_core.py,
models/uvit.py) used as the checker for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``duodiff_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  The reference's own tests hold no numeric vectors
(SURVEY.md section 8c), so the oracle is pinned against outputs of the reference
itself, imported on CPU in the build container by ``oracle/gen_golden.py`` and
committed as ``tests/golden/*.npz`` (data only, no reference source).
"""
from .schedule_oracle import (  # noqa: F401
    sampler_schedule,
    scheduler_schedule,
    ddpm_step,
    predict_original_step,
    predict_previous_step,
    ddim_step,
)
from .uvit_oracle import (  # noqa: F401
    UViTOracle,
    timestep_embedding,
    layer_norm,
    gelu_erf,
    attention,
    block_forward,
    patch_embed,
    unpatchify,
    conv3x3,
)
from .uvit_oracle_torch import UViTTorchOracle  # noqa: F401
from .vae_oracle import vae_decode  # noqa: F401
from .early_exit_oracle import EarlyExitOracle, early_exit_select, ee_get_samples  # noqa: F401
from .sampling_oracle import get_samples, get_samples_ddim, scheduler_sample, seed_everything  # noqa: F401
