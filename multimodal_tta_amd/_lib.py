"""ctypes binding of libmmtta.so (the C ABI declared in include/mmtta.h).

There is deliberately no fallback: if the library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmmtta.so")

F32, BF16 = 0, 1
CONV_FWD, CONV_DGRAD, CONVT_FWD, CONVT_DGRAD = 0, 1, 2, 3
NORM_INSTANCE, NORM_BATCH, NORM_GROUP = 0, 1, 2


class MmttaError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p),
        ("n", C.c_int32), ("c", C.c_int32), ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("sn", C.c_int64), ("sc", C.c_int64), ("sd", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64),
        ("dtype", C.c_int32), ("flags", C.c_int32),
    ]


class NormOnLoad(C.Structure):
    _fields_ = [
        ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("relu", C.c_int32), ("_pad", C.c_int32), ("scale", C.c_void_p), ("shift", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("op", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
        ("dtype", C.c_int32),
    ]


class PackItem(C.Structure):
    _fields_ = [("desc", ConvDesc), ("w_master", C.c_void_p), ("packed", C.c_void_p)]


class ParamSets(C.Structure):          # mmtta_param_sets
    _fields_ = [("items_per_set", C.c_int32), ("inner", C.c_int32), ("packed_outer", C.c_int64), ("packed_inner", C.c_int64),
                ("weight_outer", C.c_int64), ("weight_inner", C.c_int64), ("bias_outer", C.c_int64), ("bias_inner", C.c_int64)]


class ConvPlan(C.Structure):
    _fields_ = [
        ("tiles", C.c_int32), ("launches", C.c_int32), ("ksplit", C.c_int32), ("stats_rows", C.c_int32),
        ("config", C.c_int32), ("_pad", C.c_int32), ("workspace_bytes", C.c_int64),
    ]


class OptimDesc(C.Structure):          # mmtta_optim_desc
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("momentum", C.c_float), ("dampening", C.c_float), ("nesterov", C.c_int32)]


OPTIM_ADAM, OPTIM_ADAMW, OPTIM_SGD = 0, 1, 2


class ConvEpilogue(C.Structure):
    _fields_ = [("add", C.POINTER(Tensor)), ("add_norm", NormOnLoad)]


class IntensityRule(C.Structure):      # mmtta_intensity_rule
    _fields_ = [("clip", C.c_int32), ("zscore", C.c_int32), ("masked", C.c_int32), ("min_count", C.c_int32),
                ("lo", C.c_float), ("hi", C.c_float), ("mask_gt", C.c_float), ("eps", C.c_float),
                ("mean", C.c_float), ("std", C.c_float), ("legacy", C.c_int32), ("_pad", C.c_int32)]


_P = C.POINTER
_SIGNATURES = {
    "mmtta_last_error": (C.c_char_p, []),
    "mmtta_abi_version": (C.c_int, []),
    "mmtta_set_option": (C.c_int, [C.c_int, C.c_int]),
    "mmtta_intensity_scratch_bytes": (C.c_int64, [C.c_int]),
    "mmtta_intensity_normalize": (C.c_int, [_P(Tensor), _P(IntensityRule), _P(Tensor), C.c_void_p, C.c_void_p]),
    "mmtta_copy_strided": (C.c_int, [_P(Tensor), _P(Tensor), C.c_void_p]),
    "mmtta_conv_packed_bytes": (C.c_int64, [_P(ConvDesc)]),
    "mmtta_conv_pack_weights": (C.c_int, [_P(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmtta_conv_pack_table_bytes": (C.c_int64, [C.c_int]),
    "mmtta_conv_pack_table_build": (C.c_int, [_P(PackItem), C.c_int, C.c_void_p, _P(C.c_int64)]),
    "mmtta_conv_pack_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "mmtta_conv_plan": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(Tensor), _P(ConvPlan)]),
    "mmtta_conv_run": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(NormOnLoad), C.c_void_p, C.c_void_p, _P(ConvEpilogue),
                                 _P(Tensor), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mmtta_conv_run_sets": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(NormOnLoad), C.c_void_p, C.c_void_p, _P(ConvEpilogue),
                                      _P(Tensor), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, _P(ParamSets), C.c_void_p]),
    "mmtta_conv_wgrad_workspace_bytes": (C.c_int64, [_P(ConvDesc), _P(Tensor), _P(Tensor)]),
    "mmtta_conv_wgrad_workspace_bytes_sets": (C.c_int64, [_P(ConvDesc), _P(Tensor), _P(Tensor), _P(ParamSets)]),
    "mmtta_conv_wgrad_sets": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(NormOnLoad), _P(Tensor), C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int64, _P(ParamSets), C.c_void_p]),
    "mmtta_conv_wgrad_kernel": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(Tensor)]),
    "mmtta_conv_wgrad": (C.c_int, [_P(ConvDesc), _P(Tensor), _P(NormOnLoad), _P(Tensor), C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mmtta_norm_stats_finalize": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                            C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "mmtta_reduce_rows_per_n": (C.c_int, [_P(Tensor)]),
    "mmtta_channel_stats": (C.c_int, [_P(Tensor), C.c_void_p, C.c_void_p]),
    "mmtta_combine": (C.c_int, [_P(Tensor), _P(NormOnLoad), _P(Tensor), _P(NormOnLoad), _P(Tensor), C.c_void_p]),
    "mmtta_norm_bwd_reduce": (C.c_int, [_P(Tensor), _P(Tensor), _P(NormOnLoad), C.c_void_p, C.c_void_p]),
    "mmtta_norm_bwd_finalize": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p]),
    "mmtta_norm_bwd_apply": (C.c_int, [_P(Tensor), _P(Tensor), _P(NormOnLoad), C.c_void_p, C.c_void_p, _P(Tensor),
                                       C.c_void_p]),
    "mmtta_norm_bwd_small_ok": (C.c_int, [_P(Tensor), _P(Tensor), _P(NormOnLoad), _P(Tensor)]),
    "mmtta_norm_bwd_small": (C.c_int, [_P(Tensor), _P(Tensor), _P(NormOnLoad), C.c_int64, _P(Tensor), C.c_void_p]),
    "mmtta_upsample2x_fwd": (C.c_int, [_P(Tensor), _P(Tensor), C.c_void_p]),
    "mmtta_upsample2x_bwd": (C.c_int, [_P(Tensor), _P(Tensor), C.c_int, C.c_void_p]),
    "mmtta_lincomb": (C.c_int, [C.c_int, _P(_P(Tensor)), _P(C.c_float), _P(Tensor), C.c_int, C.c_void_p]),
    "mmtta_entropy_partials": (C.c_int64, [_P(Tensor)]),
    "mmtta_entropy_loss": (C.c_int, [_P(Tensor), C.c_int, _P(Tensor), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmtta_entropy_partials_items": (C.c_int64, [_P(Tensor)]),
    "mmtta_entropy_loss_items": (C.c_int, [_P(Tensor), C.c_int, _P(Tensor), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmtta_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "mmtta_optim_step": (C.c_int, [_P(OptimDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_void_p]),
    "mmtta_optim_step_sets": (C.c_int, [_P(OptimDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "mmtta_mask_dice_counts": (C.c_int, [_P(Tensor), _P(Tensor), C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmtta_dice_ce_scratch_bytes": (C.c_int64, [_P(Tensor)]),
    "mmtta_dice_ce_sums": (C.c_int, [_P(Tensor), _P(Tensor), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmtta_dice_ce_grad": (C.c_int, [_P(Tensor), _P(Tensor), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p, _P(Tensor), C.c_void_p]),
    "mmtta_surface_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "mmtta_surface_distances": (C.c_int, [C.c_void_p, _P(Tensor), _P(C.c_double), C.c_double, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmmtta.so and type every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmttaError(
            f"{LIB_PATH} is missing: build it with `python -m multimodal_tta_amd.build` (hipcc, gfx950). "
            "There is no CPU or PyTorch fallback for the adaptation path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the binding drifted apart
        fn.restype = res
        fn.argtypes = args
    if lib.mmtta_abi_version() != 2:
        raise MmttaError("libmmtta.so ABI version mismatch")
    if os.environ.get("MMTTA_NO_PIPE", "0") == "1":      # A/B aid: MMTTA_OPT_IGEMM_PIPELINE off (same results bit for bit)
        lib.mmtta_set_option(6, 0)
    if os.environ.get("MMTTA_NO_EPIVEC", "0") == "1":    # A/B aid: MMTTA_OPT_EPILOGUE_VEC16 off
        lib.mmtta_set_option(9, 0)
    if "MMTTA_WGVEC" in os.environ:                      # A/B aid: MMTTA_OPT_WGRAD_VECTOR_STAGING
        lib.mmtta_set_option(11, int(os.environ["MMTTA_WGVEC"]))
    if "MMTTA_THINMFMA" in os.environ:                   # A/B aid: MMTTA_OPT_THIN_MFMA
        lib.mmtta_set_option(13, int(os.environ["MMTTA_THINMFMA"]))
    if "MMTTA_CLSFUSE" in os.environ:                    # A/B aid: MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS
        lib.mmtta_set_option(12, int(os.environ["MMTTA_CLSFUSE"]))
    if "MMTTA_LEAN" in os.environ:                       # A/B aid: MMTTA_OPT_IGEMM_LEAN
        lib.mmtta_set_option(10, int(os.environ["MMTTA_LEAN"]))
    _lib = lib
    return lib


def exported_names() -> Sequence[str]:
    return list(_SIGNATURES)


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().mmtta_last_error().decode("utf-8", "replace")
        raise MmttaError(f"{what or 'libmmtta'} failed with status {status}: {msg}")


def stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


TENSOR_OWNS_PAD = 1   # include/mmtta.h MMTTA_TENSOR_OWNS_PAD


def desc_cl(t: torch.Tensor) -> Tensor:
    """Descriptor of a channels-last activation view: torch shape [N, D, H, W, C], stride(C) == 1.  fp32, or bf16 for the
    forward activations of `bf16` precision (storage type travels in the descriptor; strides are in elements)."""
    if t.dim() != 5 or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda:
        raise MmttaError(f"expected a CUDA float32 / bfloat16 [N,D,H,W,C] view, got {tuple(t.shape)} {t.dtype} {t.device}")
    n, d, h, w, c = t.shape
    sn, sd, sh, sw, sc = t.stride()
    if sc != 1 and c != 1:
        raise MmttaError("activation views must have unit stride along C")
    flags = TENSOR_OWNS_PAD if getattr(t, "_mmtta_owns_pad", False) else 0
    return Tensor(t.data_ptr(), n, c, d, h, w, sn, 1, sd, sh, sw, BF16 if t.dtype == torch.bfloat16 else F32, flags)


def desc_ncdhw(t: torch.Tensor) -> Tensor:
    """Descriptor of a boundary tensor: torch shape [N, C, D, H, W], any strides."""
    if t.dim() != 5 or t.dtype != torch.float32 or not t.is_cuda:
        raise MmttaError(f"expected a CUDA float32 [N,C,D,H,W] tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    n, c, d, h, w = t.shape
    sn, sc, sd, sh, sw = t.stride()
    return Tensor(t.data_ptr(), n, c, d, h, w, sn, sc, sd, sh, sw, F32, 0)


def norm_on_load(mean=None, rstd=None, gamma=None, beta=None, relu=False, scale=None, shift=None) -> NormOnLoad:
    return NormOnLoad(ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), 1 if relu else 0, 0, ptr(scale), ptr(shift))
