"""ctypes binding of librpde_hip.so (the C ABI declared in include/rpde.h).

There is no CPU fallback: if the library is missing or a tensor is not a
contiguous fp32 HIP tensor the call raises.  PyTorch only provides device
memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPDE_LIB: load another build of the same library (the timestamp-instrumented debug variant)
LIB_PATH = os.environ.get("RPDE_LIB") or os.path.join(_HERE, "lib", "librpde_hip.so")

ACT = {"identity": 0, "gelu": 1, "relu": 2}
NORM = {"backward": 0, "ortho": 1, "forward": 2}
MODE = {"full": 0, "low-pass": 1}

ERR_ARG, ERR_HIP, ERR_WORKSPACE, ERR_MODES = -1, -2, -3, -4


class RpdeError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("batch", C.c_int), ("zdiv", C.c_int),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64),
        ("ksplit", C.c_int), ("sCk", C.c_int64),
        ("alpha", C.c_float), ("accumulate", C.c_int),
        ("bias", C.c_void_p), ("bias_mode", C.c_int),
        ("act_a", C.c_int), ("act_b", C.c_int),
        ("epi_dact", C.c_int),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_ld", C.c_int64),
        ("write_act", C.c_int), ("drop_where", C.c_int),
        ("colsum", C.c_void_p), ("aux_out", C.c_void_p), ("b_split", C.c_void_p), ("a_split", C.c_void_p), ("acc_src", C.c_void_p),
        ("drop_epoch", C.c_void_p),
    ]


class FFParams(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int), ("dim", C.c_int), ("factor", C.c_int),
        ("layer_norm", C.c_int), ("ln_eps", C.c_float),
        ("dropout_p", C.c_float), ("seed", C.c_uint64),
        ("post_act", C.c_int),
        ("weights", C.POINTER(C.c_void_p)), ("biases", C.POINTER(C.c_void_p)),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("seed_epoch", C.c_void_p),
    ]


_P, _I, _L, _Z, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_double
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/rpde.h one to one
_SIGNATURES = {
    "rpde_last_error": (C.c_char_p, []),
    "rpde_version": (_I, []),
    "rpde_plan_cache_count": (_I, []),
    "rpde_plan_create": (_I, [C.POINTER(_P), _I, _I, _I, _P]),
    "rpde_plan_destroy": (_I, [_P]),
    "rpde_plan_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "rpde_plan_tables": (_I, [_P, _P, _P]),
    "rpde_gemm_f32": (_I, [C.POINTER(GemmDesc), _P]),
    "rpde_split_weights_bytes": (C.c_size_t, [_I, _I]),
    "rpde_split_weights": (_I, [_P, _I, C.c_int64, _I, _I, _P, _P]),
    "rpde_fspectral1d_ws_bytes": (_Z, [_I, _I, _I, _I]),
    "rpde_fspectral1d_spec_elems": (_Z, [_I, _I, _I, _I]),
    "rpde_fspectral1d_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fspectral1d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fspectral2d_ws_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "rpde_fspectral2d_spec_elems": (_Z, [_I, _I, _I, _I, _I, _I]),
    "rpde_fspectral2d_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fspectral2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_spectral1d_ws_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "rpde_spectral1d_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_spectral1d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_spectral2d_ws_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "rpde_spectral2d_spec_elems": (_Z, [_I, _I, _I, _I, _I, _I]),
    "rpde_spectral2d_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_spectral2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_feedforward_ws_bytes": (_Z, [_L, _I, _I, _I]),
    "rpde_feedforward_fwd_ws_bytes": (_Z, [_I, _I, _I]),
    "rpde_feedforward_is_fused": (_I, [_I, _I, _I, _L]),
    "rpde_feedforward_fwd": (_I, [C.POINTER(FFParams), _P, _P, _PP, _PP, _P, _P, _L, _P, _Z, _P]),
    "rpde_feedforward_bwd": (_I, [C.POINTER(FFParams), _P, _PP, _PP, _P, _P, _P, _PP, _PP, _P, _P, _L, _P, _Z, _P]),
    "rpde_feedforward_prepare": (_I, [C.POINTER(FFParams), _P, _Z, _P]),
    "rpde_feedforward_fwd_prepared": (_I, [C.POINTER(FFParams), _P, _P, _P, _L, _P, _Z, _P]),
    "rpde_fspectral2d_prep_bytes": (_Z, [_I, _I, _I, _I]),
    "rpde_fspectral2d_eval_ws_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "rpde_fspectral2d_prepare": (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fspectral2d_fwd_prepared": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fno2d_lift_block_eval_ws_bytes": (_Z, [_I] * 7),
    "rpde_fno2d_lift_block_eval_ok": (_I, [_I] * 7),
    "rpde_fno2d_lift_block_eval_fwd": (_I, [_P] * 10 + [_I] * 8 + [_P, _Z, _P]),
    "rpde_linear_ws_bytes": (_Z, [_L, _I, _I]),
    "rpde_weight_norm_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "rpde_weight_norm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "rpde_linear_fwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "rpde_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _Z, _P]),
    "rpde_conv1x1_ws_bytes": (_Z, [_I, _I, _I, _L]),
    "rpde_conv1x1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P]),
    "rpde_conv1x1_act_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _I, _P]),
    "rpde_fnoblock2d_eval_ws_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "rpde_fnoblock2d_eval_ok": (_I, [_I, _I, _I, _I, _I]),
    "rpde_fnoblock2d_eval_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_fnoblock2d_proj_eval_ok": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "rpde_fnoblock2d_proj_eval_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_conv_mlp_ok": (_I, [_I, _I, _I, _L]),
    "rpde_conv_mlp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _P]),
    "rpde_conv1x1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P, _Z, _P]),
    "rpde_resize1d_ws_bytes": (_Z, [_L, _I, _I]),
    "rpde_resize1d": (_I, [_P, _P, _L, _I, _I, _P, _Z, _P]),
    "rpde_resize2d_ws_bytes": (_Z, [_L, _I, _I, _I, _I]),
    "rpde_resize2d": (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _Z, _P]),
    "rpde_concat_grid": (_I, [_P, _P, _I, _I, _I, _I, _I, _D, _D, _I, _P, _P, _P]),
    "rpde_transpose_cs": (_I, [_P, _P, _I, _L, _I, _I, _P]),
    "rpde_act_fwd": (_I, [_P, _P, _L, _I, _P]),
    "rpde_act_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "rpde_rel_l2_stats_elems": (_L, [_I]),
    "rpde_rel_l2_fwd": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _P]),
    "rpde_rel_l2_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _P]),
    "rpde_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P]),
    "rpde_adamw_step_dev": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P]),
    "rpde_adamw_set_hyper_dev": (_I, [_P, _F, _F, _P]),
    "rpde_adamw_apply_dev": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (no device needed) and bind every signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RpdeError(
            f"{LIB_PATH} is missing: build it with `python resolution-pde_amd/rpde/build.py` "
            "(hipcc, gfx950).  There is no CPU fallback for the spectral hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status == 0:
        return
    msg = (load().rpde_last_error() or b"").decode(errors="replace")
    text = f"rpde {what} failed ({status}): {msg}"
    if status == ERR_ARG and "not recognized" in msg:
        raise ValueError(msg)
    raise RpdeError(text)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """device pointer of a contiguous fp32 (or complex64, as floats) HIP tensor"""
    if t is None:
        return None
    if not t.is_cuda:
        raise RpdeError("the spectral hot path runs on the GPU only (got a CPU tensor); "
                        "the CPU oracle lives under oracle/ and is test infrastructure")
    if t.dtype not in (torch.float32, torch.complex64):
        raise RpdeError(f"expected float32/complex64, got {t.dtype}")
    if not t.is_contiguous():
        raise RpdeError("expected a contiguous tensor")
    return t.data_ptr()


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = ptr(t)
    return arr
