"""Oracle for the optional surface metrics of the evaluator (HD95 / ASD).

Test infrastructure.  The reference computes these through MONAI (src/evaluation/seg_eval.py:226-236:
``HausdorffDistanceMetric(include_background=True, reduction="none", percentile=95, directed=False)`` and
``compute_average_surface_distance``; called at :327 and :334-341 with ``spacing=evaluation.seg.spacing``), then
applies its own penalty / sanitising (:342-355).  MONAI is neither vendored nor installed (PARITY UNPINNED, see
``oracle/__init__.py``); what follows restates the published algorithm of ``monai.metrics.utils`` (>= 1.3):

* ``get_mask_edges``: edges = mask XOR ``scipy.ndimage.binary_erosion(mask)`` (default cross-shaped structure,
  border_value 0).  MONAI first crops both masks to their joint bounding box with a margin of 1; the crop only
  shortens scipy's work, erosion and distances are unchanged by it, so it is not reproduced.
* ``get_surface_distance(A, B)``: ``distance_transform_edt(~B, sampling=spacing)`` read at the voxels of A, as
  float32; all-inf when B has no edge; when A has no edge but B has, an inf array with one entry per voxel of B
  (a MONAI quirk that only matters for the mean).
* HD: per direction ``torch.quantile(d, percentile / 100)`` (float32, linear interpolation; NaN for an empty set),
  then the maximum over both directions.  ASD: mean of d(pred -> gt), joined with d(gt -> pred) when symmetric; NaN when
  there are no distances.
* evaluator post-processing (reference code, not MONAI): GT non-empty and prediction empty -> volume diagonal in mm;
  non-finite values of valid entries -> diagonal.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np
import torch
from scipy import ndimage


def mask_edges(mask: np.ndarray) -> np.ndarray:
    mask = np.asarray(mask, dtype=bool)
    if not mask.any():
        return np.zeros_like(mask)
    return ndimage.binary_erosion(mask) ^ mask


def surface_distance(edges_a: np.ndarray, edges_b: np.ndarray, spacing: Sequence[float]) -> np.ndarray:
    if not edges_b.any():
        return np.full(int(edges_a.sum()), np.inf, dtype=np.float32)
    if not edges_a.any():
        return np.full(int(edges_b.sum()), np.inf, dtype=np.float32)
    dis = ndimage.distance_transform_edt(~edges_b, sampling=list(spacing))
    return dis.astype(np.float32)[edges_a]


def percentile_distance(d: np.ndarray, percentile: float) -> float:
    if d.shape == (0,):
        return float("nan")
    return float(torch.quantile(torch.from_numpy(np.ascontiguousarray(d)), percentile / 100.0).item())


def hd_asd(pred: np.ndarray, gt: np.ndarray, spacing: Sequence[float] = (1.0, 1.0, 1.0), percentile: float = 95.0,
           asd_symmetric: bool = False) -> Tuple[float, float]:
    """One (volume, region): boolean masks [D,H,W] -> (hd, asd) as MONAI returns them (before the evaluator's fix-ups)."""
    ep, eg = mask_edges(pred), mask_edges(gt)
    d_pg = surface_distance(ep, eg, spacing)
    d_gp = surface_distance(eg, ep, spacing)
    q = torch.tensor([percentile_distance(d_pg, percentile), percentile_distance(d_gp, percentile)])
    hd = float(torch.max(q).item())
    d = np.concatenate([d_pg, d_gp]) if asd_symmetric else d_pg
    asd = float("nan") if d.shape == (0,) else float(torch.from_numpy(d).mean().item())
    return hd, asd


def diag_mm(d: int, h: int, w: int, spacing: Sequence[float]) -> float:
    """reference src/evaluation/seg_eval.py:89-103"""
    dd, hh, ww = max(d - 1, 0) * spacing[0], max(h - 1, 0) * spacing[1], max(w - 1, 0) * spacing[2]
    return float(math.sqrt(dd * dd + hh * hh + ww * ww))


def evaluator_surface(pred: torch.Tensor, gt: torch.Tensor, spacing: Sequence[float], asd_symmetric: bool = False):
    """pred, gt: uint8 [B,R,D,H,W] -> hd95 [B,R], asd [B,R] after the reference's penalty and sanitising (:342-355)."""
    B, R = pred.shape[:2]
    D, H, W = (int(v) for v in pred.shape[2:])
    dm = diag_mm(D, H, W, spacing)
    hd = torch.empty(B, R)
    asd = torch.empty(B, R)
    for b in range(B):
        for r in range(R):
            hv, av = hd_asd(pred[b, r].numpy() > 0, gt[b, r].numpy() > 0, spacing, 95.0, asd_symmetric)
            hd[b, r], asd[b, r] = hv, av
    valid = gt.reshape(B, R, -1).sum(-1) > 0
    pred_empty = pred.reshape(B, R, -1).sum(-1) == 0
    pen = valid & pred_empty
    hd[pen] = dm
    asd[pen] = dm
    hd[(~torch.isfinite(hd)) & valid] = dm
    asd[(~torch.isfinite(asd)) & valid] = dm
    return hd, asd
