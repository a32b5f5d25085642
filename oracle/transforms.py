"""Oracle of the input normalisation pre-pass.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates ``_normalize_img`` of the reference (src/datasets/transforms.py:129-223) on torch CPU tensors.
PARITY UNPINNED: the reference holds no fixtures for it and its module does not import here (omegaconf/monai).
"""
from typing import Any, Dict, Optional, Sequence

import torch


def normalize_image(img: torch.Tensor, normalize: bool = True, intensity_policy: Optional[Dict[str, Any]] = None,
                    mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
                    channel_names: Optional[Sequence[str]] = None) -> torch.Tensor:
    """img: [C,D,H,W].  (A) intensity policy (:147-198)  (B) legacy mean/std (:200-223)."""
    if not normalize:                                          # :137-140
        return img
    if img.ndim != 4:                                          # :142-143
        raise ValueError(f"[3DTransforms] expect image [C,D,H,W], got {tuple(img.shape)}")
    c = int(img.shape[0])
    ip = dict(intensity_policy or {})
    if bool(ip.get("enabled", False)):                         # :147
        chans = ip.get("channels", {}) if isinstance(ip.get("channels", {}), dict) else {}
        if channel_names is None:
            cn = ip.get("channel_names", None)                 # :126-129
            if isinstance(cn, (list, tuple)) and len(cn) > 0:
                channel_names = [str(x) for x in cn]
        if channel_names is None:
            names = [str(i) for i in range(c)]                 # :150-152
        else:
            if len(channel_names) != c:                        # :154-158
                raise RuntimeError(f"[3DTransforms] len(channel_names)={len(channel_names)} != C={c}.")
            names = [str(x) for x in channel_names]
        out = img.clone()
        for ci, name in enumerate(names):
            rule = chans.get(name, {})
            if not isinstance(rule, dict):
                rule = {}
            x = out[ci]
            clip = rule.get("clip", None)                      # :171-175
            if isinstance(clip, (list, tuple)) and len(clip) == 2:
                x = torch.clamp(x, min=float(clip[0]), max=float(clip[1]))
            zc = rule.get("zscore", None)                      # :178-196
            if isinstance(zc, dict):
                masked = bool(zc.get("masked", True))
                mask_gt = float(zc.get("mask_gt", float("-inf")))
                eps = float(zc.get("eps", 1.0e-6))
                min_count = int(zc.get("min_count", 16))
                if masked:
                    m = x > mask_gt
                    vals = x[m] if int(m.sum().item()) >= min_count else x.reshape(-1)
                else:
                    vals = x.reshape(-1)
                mu = vals.mean()
                sd = vals.std(unbiased=False).clamp_min(eps)
                x = (x - mu) / sd
            out[ci] = x
        return out
    mean_t = torch.zeros(c, dtype=img.dtype) if mean is None else torch.as_tensor(mean, dtype=img.dtype)   # :202-211
    if mean_t.numel() == 1:
        mean_t = mean_t.repeat(c)
    if mean_t.numel() != c:
        raise RuntimeError(f"[3DTransforms] len(mean)={mean_t.numel()} != C={c}")
    std_t = torch.ones(c, dtype=img.dtype) if std is None else torch.as_tensor(std, dtype=img.dtype)       # :213-220
    if std_t.numel() == 1:
        std_t = std_t.repeat(c)
    if std_t.numel() != c:
        raise RuntimeError(f"[3DTransforms] len(std)={std_t.numel()} != C={c}")
    shape = (c,) + (1,) * (img.ndim - 1)
    return (img - mean_t.view(shape)) / std_t.view(shape)                                                   # :222-223
