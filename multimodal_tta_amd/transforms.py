"""Input normalisation pre-pass on the GPU: the reference's ``_normalize_img``
(reference src/datasets/transforms.py:129-223), configured by the same keys
(``training.data.transforms.{normalize,intensity_policy,mean,std}``, reference
configs/_global_patches/hecktor21.yaml:22-50, brats.yaml:39-40), computed by ``mmtta_intensity_normalize``.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import _lib
from .ops import check, ptr, stream_ptr, MmttaError

_NEG_INF = float("-inf")


def build_rules(channels: int, intensity_policy: Optional[Dict[str, Any]] = None, mean: Optional[Sequence[float]] = None,
                std: Optional[Sequence[float]] = None, channel_names: Optional[Sequence[str]] = None) -> List[_lib.IntensityRule]:
    """Per-channel rules with the reference's precedence: an enabled intensity policy wins over mean/std."""
    ip = dict(intensity_policy or {})
    rules = []
    if bool(ip.get("enabled", False)):
        chans = ip.get("channels", {}) if isinstance(ip.get("channels", {}), dict) else {}
        if channel_names is None:
            cn = ip.get("channel_names", None)
            if isinstance(cn, (list, tuple)) and len(cn) > 0:
                channel_names = [str(x) for x in cn]
        if channel_names is None:
            names = [str(i) for i in range(channels)]
        else:
            if len(channel_names) != channels:
                raise RuntimeError(f"[3DTransforms] len(channel_names)={len(channel_names)} != C={channels}. "
                                   "Please set dataset.modality_order (or transforms.channel_names) to match channels.")
            names = [str(x) for x in channel_names]
        for name in names:
            rule = chans.get(name, {})
            if not isinstance(rule, dict):
                rule = {}
            r = _lib.IntensityRule(0, 0, 0, 16, 0.0, 0.0, _NEG_INF, 1e-6, 0.0, 1.0, 0, 0)
            clip = rule.get("clip", None)
            if isinstance(clip, (list, tuple)) and len(clip) == 2:
                r.clip, r.lo, r.hi = 1, float(clip[0]), float(clip[1])
            zc = rule.get("zscore", None)
            if isinstance(zc, dict):
                r.zscore = 1
                r.masked = 1 if bool(zc.get("masked", True)) else 0
                r.mask_gt = float(zc.get("mask_gt", _NEG_INF))
                r.eps = float(zc.get("eps", 1.0e-6))
                r.min_count = int(zc.get("min_count", 16))
            rules.append(r)
        return rules

    def per_channel(v, default):
        if v is None:
            return [default] * channels
        v = [float(t) for t in (v if isinstance(v, (list, tuple)) else [v])]
        if len(v) == 1:
            v = v * channels
        if len(v) != channels:
            raise RuntimeError(f"[3DTransforms] len(mean/std)={len(v)} != C={channels}")
        return v
    m, s = per_channel(mean, 0.0), per_channel(std, 1.0)
    return [_lib.IntensityRule(0, 0, 0, 16, 0.0, 0.0, _NEG_INF, 1e-6, m[c], s[c], 1, 0) for c in range(channels)]


def normalize_image(img: torch.Tensor, normalize: bool = True, intensity_policy: Optional[Dict[str, Any]] = None,
                    mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
                    channel_names: Optional[Sequence[str]] = None) -> torch.Tensor:
    """img: [C,D,H,W] fp32 on the GPU -> normalised copy (same shape, contiguous)."""
    if not normalize:
        return img
    if img.ndim != 4:
        raise ValueError(f"[3DTransforms] expect image [C,D,H,W], got {tuple(img.shape)}")
    if not img.is_cuda or img.dtype != torch.float32:
        raise MmttaError("normalize_image computes on an MI355X through libmmtta.so: pass a CUDA float32 tensor")
    c = int(img.shape[0])
    rules = build_rules(c, intensity_policy, mean, std, channel_names)
    lib = _lib.load()
    x = img.contiguous().unsqueeze(0)
    y = torch.empty_like(x)
    nbytes = int(lib.mmtta_intensity_scratch_bytes(c))
    if nbytes < 0:
        raise MmttaError(f"normalize_image: {c} channels unsupported")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    arr = (_lib.IntensityRule * c)(*rules)
    tx, ty = _lib.desc_ncdhw(x), _lib.desc_ncdhw(y)
    check(lib.mmtta_intensity_normalize(C.byref(tx), arr, C.byref(ty), ptr(scratch), stream_ptr()), "intensity_normalize")
    return y[0]
