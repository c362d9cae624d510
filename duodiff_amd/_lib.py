"""ctypes binding of libduodiff.so (include/duodiff.h).  No CPU fallback: a missing
library is an ImportError-class failure raised at first use, loudly."""
import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(os.environ.get("DUODIFF_LIB") or Path(__file__).resolve().parent / "libduodiff.so")

DD_OK = 0
DD_ERR_INVALID, DD_ERR_NOT_FOUND, DD_ERR_STATE, DD_ERR_HIP, DD_ERR_NOMEM, DD_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
DD_PREC_BF16, DD_PREC_FP32 = 0, 1
DD_VAR_BETA_TILDE, DD_VAR_BETA = 0, 1
DD_NOISE_NONE, DD_NOISE_BUFFER, DD_NOISE_PHILOX = 0, 1, 2
DD_EE_MLP_PER_LAYER, DD_EE_MLP_PER_TIMESTEP, DD_EE_MLP_PER_LAYER_PER_TIMESTEP, DD_EE_ATTENTION_PROBE = 0, 1, 2, 3
ABI_VERSION = 5
DD_DEV_NO_FUSED_MLP, DD_DEV_NO_FUSED_PROJ, DD_DEV_NO_FUSED_HEAD, DD_DEV_GENERIC_EMBED, DD_DEV_MLP_EXTRAS_ONLY = 1, 2, 4, 8, 16
DD_DEV_NO_FUSED_SKIP, DD_DEV_NO_FUSED_QKV, DD_DEV_NO_FUSED_QA = 32, 64, 128
DD_PROF_DOMINANT, DD_PROF_BLOCK_TAIL, DD_PROF_FC1, DD_PROF_ROWLIN, DD_PROF_QKV_ATTENTION, DD_PROF_SPLITK = 0, 1, 2, 3, 4, 5
DD_DEV_NO_CHAINS, DD_DEV_FORCE_CHAINS, DD_DEV_NO_ROWLIN, DD_DEV_NO_ROWLIN_PROJ, DD_DEV_NO_EMBED_LN, DD_DEV_NO_SPLITK, DD_DEV_NO_ROWLIN_SKIP = 256, 512, 1024, 2048, 4096, 8192, 16384
DD_DEV_NO_SPLIT_HEADS = 32768


class dd_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "img_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads", "mlp_ratio",
        "num_classes", "normalize_timesteps", "max_batch", "qkv_bias", "mlp_time_embed")]


class dd_sample_args(C.Structure):
    _fields_ = [("first", C.c_void_p), ("late", C.c_void_p), ("t_switch", C.c_int32),
                ("t_start", C.c_int32), ("t_end", C.c_int32), ("variance", C.c_int32),
                ("noise_mode", C.c_int32), ("use_graph", C.c_int32), ("seed", C.c_uint64),
                ("y_dev", C.c_void_p), ("x_dev", C.c_void_p), ("B", C.c_int32), ("reserved", C.c_int32)]


class dd_affine_sample_args(C.Structure):
    _fields_ = [("first", C.c_void_p), ("late", C.c_void_p), ("n_steps", C.c_int32), ("switch_after", C.c_int32),
                ("t", C.POINTER(C.c_float)), ("a", C.POINTER(C.c_float)), ("b", C.POINTER(C.c_float)),
                ("c", C.POINTER(C.c_float)), ("noise", C.POINTER(C.c_int32)), ("noise_mode", C.c_int32),
                ("use_graph", C.c_int32), ("seed", C.c_uint64), ("y_dev", C.c_void_p), ("x_dev", C.c_void_p),
                ("B", C.c_int32), ("counter_base", C.c_int32)]


class dd_ee_sample_args(C.Structure):
    _fields_ = [("model", C.c_void_p), ("threshold", C.c_float), ("t_start", C.c_int32), ("t_end", C.c_int32),
                ("noise_mode", C.c_int32), ("use_graph", C.c_int32), ("B", C.c_int32), ("seed", C.c_uint64),
                ("y_dev", C.c_void_p), ("x_dev", C.c_void_p), ("err_dev", C.c_void_p), ("idx_dev", C.c_void_p)]


# every symbol include/duodiff.h (and include/duodiff_dev.h: dd_dev_*) declares: name -> (restype, argtypes)
SIGNATURES = {
    "dd_abi_version": (C.c_int, []),
    "dd_build_id": (C.c_char_p, []),
    "dd_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "dd_ctx_destroy": (None, [C.c_void_p]),
    "dd_last_error": (C.c_char_p, [C.c_void_p]),
    "dd_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dd_schedule_table": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "dd_schedule_build": (C.c_int, [C.c_float, C.c_float, C.c_int] + [C.POINTER(C.c_float)] * 5),
    "dd_model_create": (C.c_int, [C.c_void_p, C.POINTER(dd_config), C.POINTER(C.c_void_p)]),
    "dd_model_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "dd_model_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "dd_model_num_params": (C.c_int64, [C.c_void_p]),
    "dd_model_destroy": (None, [C.c_void_p]),
    "dd_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dd_model_enable_early_exit": (C.c_int, [C.c_void_p, C.c_int]),
    "dd_forward_early_exit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dd_early_exit_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dd_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dd_ddpm_step_coef": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "dd_affine_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "dd_sample_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dd_to_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dd_sample": (C.c_int, [C.c_void_p, C.POINTER(dd_sample_args), C.c_void_p]),
    "dd_sample_affine": (C.c_int, [C.c_void_p, C.POINTER(dd_affine_sample_args), C.c_void_p]),
    "dd_sample_early_exit": (C.c_int, [C.c_void_p, C.POINTER(dd_ee_sample_args), C.c_void_p]),
    "dd_bench_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "dd_vae_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dd_vae_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "dd_vae_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "dd_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dd_vae_destroy": (None, [C.c_void_p]),
    "dd_profile_steps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dd_profile_steps_chained": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dd_profile_select": (C.c_int, [C.c_void_p, C.c_int]),
    "dd_dev_qkv_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "dd_dev_mlp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_void_p, C.POINTER(C.c_float)] + [C.c_void_p] * 8),
    "dd_dev_head_dec": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "dd_dev_poison_workspaces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dd_dev_graph_captures": (C.c_longlong, [C.c_void_p]),
    "dd_dev_last_sample_chains": (C.c_int, [C.c_void_p]),
    "dd_dev_set_flags": (C.c_int, [C.c_void_p, C.c_uint]),
    "dd_set_num_cus": (C.c_int, [C.c_void_p, C.c_int]),
    "dd_plan_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dd_last_sample_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None


class EngineUnavailable(RuntimeError):
    """libduodiff.so is missing or unusable.  There is no CPU path to fall back to."""


def load():
    """dlopen the in-tree library and attach prototypes.  Raises EngineUnavailable if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineUnavailable(
            f"{LIB_PATH} not found: build it with `python -m duodiff_amd.build` "
            "(hipcc --offload-arch=gfx950).  duodiff_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:  # e.g. libamdhip64 missing
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    lib.dd_abi_version.restype = C.c_int
    if lib.dd_abi_version() != ABI_VERSION:
        raise EngineUnavailable(f"{LIB_PATH}: ABI version {lib.dd_abi_version()}, this package binds version {ABI_VERSION}; rebuild")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # the production surface (include/duodiff.h) must be complete; a library built without the development entry
            # points (include/duodiff_dev.h: dd_dev_*), e.g. an A/B build of an older tree, still loads
            if name.startswith("dd_dev_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
