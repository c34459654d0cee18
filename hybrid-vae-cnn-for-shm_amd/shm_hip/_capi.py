"""ctypes binding of include/shm_hip.h (libshm_hip.so, built in-tree by csrc/Makefile).

There is no CPU fallback: importing this module without the built library raises ImportError, and every
entry point raises RuntimeError with shm_last_error() on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# SHM_HIP_LIB: another build of the same library (A/B measurements on one box); the default is the in-tree build
_LIB_PATH = Path(os.environ.get("SHM_HIP_LIB") or Path(__file__).resolve().parent.parent / "csrc" / "libshm_hip.so")

SHM_PREC_F32, SHM_PREC_BF16, SHM_PREC_SPLIT = 0, 1, 2
PRECISIONS = {"fp32": SHM_PREC_F32, "bf16": SHM_PREC_BF16, "split": SHM_PREC_SPLIT}


class VaeConfig(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("latent_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("num_layers", C.c_int32), ("layer_norm", C.c_int32), ("precision", C.c_int32),
                ("dropout", C.c_float)]


class CnnConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("num_classes", C.c_int32), ("seq_len", C.c_int32),
                ("num_features", C.c_int32), ("dropout", C.c_float)]


class OlCnnConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("num_classes", C.c_int32), ("seq_len", C.c_int32), ("num_features", C.c_int32)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("offset", C.c_int64), ("ndim", C.c_int32), ("shape", C.c_int32 * 4)]


def _load():
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {_LIB_PATH.parent}`); this package has no CPU fallback")
    # PyTorch ships its own libamdhip64 (SONAME libamdhip64.so.7, found through torch's RPATH).  libshm_hip.so asks for the same
    # SONAME, so it binds to whichever copy is already in the process: import torch FIRST, or the system copy is loaded beside
    # torch's and the process ends up with two HIP runtimes, the second of which finds no device ("no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(str(_LIB_PATH))
    vp, f32p = C.c_void_p, C.c_void_p
    cfgp = C.POINTER(VaeConfig)
    ccfgp = C.POINTER(CnnConfig)
    sig = {
        "shm_last_error": (C.c_char_p, []),
        "shm_version": (C.c_int, []),
        "shm_vae_param_count": (C.c_int64, [cfgp]),
        "shm_vae_param_layout": (C.c_int, [cfgp, C.POINTER(TensorDesc), C.c_int]),
        "shm_vae_plan_create": (C.c_int, [cfgp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "shm_vae_plan_destroy": (None, [vp]),
        "shm_vae_plan_workspace_bytes": (C.c_size_t, [vp]),
        "shm_vae_pack_weights": (C.c_int, [vp, f32p, vp, vp]),
        "shm_vae_forward": (C.c_int, [vp, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int32, C.c_uint64, vp, vp]),
        "shm_vae_forward_sample": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint64, f32p, f32p, f32p, f32p, C.c_int32, C.c_uint64, vp, vp]),
        "shm_eps_normal": (None, [C.c_uint64, C.c_uint64, C.c_int64, vp]),
        "shm_vae_encode": (C.c_int, [vp, f32p, f32p, f32p, C.c_int32, C.c_uint64, vp, vp]),
        "shm_vae_decode": (C.c_int, [vp, f32p, f32p, C.c_int32, C.c_uint64, vp, vp]),
        "shm_olcnn_param_count": (C.c_int64, [C.POINTER(OlCnnConfig)]),
        "shm_olcnn_param_layout": (C.c_int, [C.POINTER(OlCnnConfig), C.POINTER(TensorDesc), C.c_int]),
        "shm_olcnn_workspace_bytes": (C.c_size_t, [C.POINTER(OlCnnConfig), C.c_int32]),
        "shm_olcnn_forward": (C.c_int, [C.POINTER(OlCnnConfig), vp, vp, vp, C.c_int32, vp, vp]),
        "shm_sim_smooth_force": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]),
        "shm_sim_newmark": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                      vp, C.c_int32, vp]),
        "shm_sim_sensor_fault": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                           vp, vp, C.c_int32, vp]),
        "shm_peak_mfma": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, vp, C.POINTER(C.c_double), vp]),
        "shm_peak_hbm": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp, C.POINTER(C.c_double), vp]),
        "shm_window_gather": (C.c_int, [vp, C.c_int64, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, C.c_float, vp, vp]),
        "shm_cnn_eval_loss": (C.c_int, [vp, vp, C.c_int32, C.c_int32, vp, vp]),
        "shm_vae_eval_loss": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, vp]),
        "shm_vae_plan_status": (C.c_int, [vp, vp, vp]),
        "shm_vae_plan_status_word": (C.c_int, [vp, vp, vp, vp]),
        "shm_vae_plan_status_offset": (C.c_int64, [vp]),
        "shm_vae_plan_reset_status": (C.c_int, [vp, vp, vp]),
        "shm_set_pair_launch": (C.c_int, [C.c_int]),
        "shm_vae_loss": (C.c_int, [vp, f32p, C.c_float, f32p, vp, vp]),
        "shm_vae_backward": (C.c_int, [vp, f32p, f32p, f32p, f32p, f32p, vp, vp]),
        "shm_adam_scratch_bytes": (C.c_size_t, [C.c_int64]),
        "shm_adam_step": (C.c_int, [C.c_int64, f32p, f32p, f32p, f32p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_float, vp, f32p, vp]),
        "shm_adam_step_status": (C.c_int, [C.c_int64, f32p, f32p, f32p, f32p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.c_float, C.c_float, C.c_float, vp, f32p, vp, vp, vp]),
        "shm_cnn_param_count": (C.c_int64, [ccfgp]),
        "shm_cnn_param_layout": (C.c_int, [ccfgp, C.POINTER(TensorDesc), C.c_int]),
        "shm_cnn_plan_create": (C.c_int, [ccfgp, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "shm_cnn_plan_destroy": (None, [vp]),
        "shm_cnn_plan_workspace_bytes": (C.c_size_t, [vp]),
        "shm_cnn_forward": (C.c_int, [vp, f32p, f32p, f32p, f32p, C.c_int32, C.c_uint64, vp, vp]),
        "shm_cnn_loss": (C.c_int, [vp, vp, f32p, vp, vp]),
        "shm_cnn_backward": (C.c_int, [vp, f32p, f32p, f32p, f32p, vp, vp]),
        "shm_cnn_build_input": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, f32p, f32p, f32p, vp]),
        "shm_cnn_predict": (C.c_int, [C.c_int32, C.c_int32, f32p, vp, f32p, vp]),
        "shm_profile_begin": (C.c_int, []),
        "shm_profile_end": (C.c_int, [C.c_char_p, C.c_size_t]),
        "shm_dropout_keep_mask": (None, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp]),
        "shm_dropout_keep_mask_tiled": (None, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    return lib, tuple(sig)


lib, EXPORTS = _load()


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib.shm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")


def make_config(input_dim, latent_dim, hidden_dim, num_layers, layer_norm, dropout, precision=SHM_PREC_F32) -> VaeConfig:
    return VaeConfig(int(input_dim), int(latent_dim), int(hidden_dim), int(num_layers), int(bool(layer_norm)),
                     int(precision), float(dropout))


def param_layout(cfg: VaeConfig):
    """[(name, offset, shape)] in state_dict order, and the flat buffer length."""
    n = lib.shm_vae_param_layout(C.byref(cfg), None, 0)
    if n < 0:
        check(n, "shm_vae_param_layout")
    arr = (TensorDesc * n)()
    lib.shm_vae_param_layout(C.byref(cfg), arr, n)
    total = lib.shm_vae_param_count(C.byref(cfg))
    out = [(d.name.decode(), int(d.offset), tuple(int(d.shape[i]) for i in range(d.ndim))) for d in arr]
    return out, int(total)


def cnn_param_layout(cfg: CnnConfig):
    n = lib.shm_cnn_param_layout(C.byref(cfg), None, 0)
    if n < 0:
        check(n, "shm_cnn_param_layout")
    arr = (TensorDesc * n)()
    lib.shm_cnn_param_layout(C.byref(cfg), arr, n)
    total = lib.shm_cnn_param_count(C.byref(cfg))
    return [(d.name.decode(), int(d.offset), tuple(int(d.shape[i]) for i in range(d.ndim))) for d in arr], int(total)
