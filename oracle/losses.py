"""Oracle losses (test infrastructure; parity unpinned).

* Entropy objectives of the adaptation step: BUILD-DEFINED - the
  reference has no TTA loss.  Sigmoid/multilabel heads (the shipped configs,
  reference configs/_global_patches/brats.yaml:55 ``sigmoid: true``) use the per-channel
  Bernoulli entropy; ``training.criterion.softmax`` heads (reference
  src/core/trainers/seg_trainer.py:41-54) use the categorical entropy.
* ``DiceCELoss``: restatement of ``monai.losses.DiceCELoss`` as the reference constructs it at
  src/core/trainers/seg_trainer.py:59-79 and src/evaluation/seg_eval.py:209-220
  (SURVEY.md Appendix A.5).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


def bernoulli_entropy_loss(logits: torch.Tensor) -> torch.Tensor:
    """mean over (b, r, voxel) of H(sigmoid(z)) = softplus(z) - z*sigmoid(z)."""
    z = logits.float()
    return (F.softplus(z) - z * torch.sigmoid(z)).mean()


def categorical_entropy_loss(logits: torch.Tensor) -> torch.Tensor:
    """mean over (b, voxel) of H(softmax_r(z)) = logsumexp_r(z) - sum_r p_r z_r."""
    z = logits.float()
    logp = F.log_softmax(z, dim=1)
    return -(logp.exp() * logp).sum(dim=1).mean()


def entropy_loss(logits: torch.Tensor, softmax: bool = False) -> torch.Tensor:
    return categorical_entropy_loss(logits) if softmax else bernoulli_entropy_loss(logits)


class DiceCELoss(nn.Module):
    """lambda_dice * Dice + lambda_ce * (soft-label softmax CE if R != 1 else BCE-with-logits)."""

    def __init__(
        self,
        include_background: bool = True,
        to_onehot_y: bool = False,
        sigmoid: bool = False,
        softmax: bool = False,
        squared_pred: bool = False,
        jaccard: bool = False,
        reduction: str = "mean",
        smooth_nr: float = 1e-5,
        smooth_dr: float = 1e-5,
        weight: Optional[torch.Tensor] = None,
        lambda_dice: float = 1.0,
        lambda_ce: float = 1.0,
    ):
        super().__init__()
        if reduction != "mean":
            raise ValueError("the reference only uses reduction='mean'")
        if to_onehot_y:
            raise ValueError("to_onehot_y is not on the adaptation path (multilabel region masks)")
        self.include_background, self.sigmoid, self.softmax = include_background, sigmoid, softmax
        self.squared_pred, self.jaccard = squared_pred, jaccard
        self.smooth_nr, self.smooth_dr = smooth_nr, smooth_dr
        self.lambda_dice, self.lambda_ce = lambda_dice, lambda_ce
        w = None if weight is None else torch.as_tensor(weight, dtype=torch.float32)
        self.register_buffer("weight", w)
        dice_w = w[1:] if (w is not None and not include_background) else w
        self.register_buffer("dice_weight", dice_w)

    def dice(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        p = logits
        if self.sigmoid:
            p = torch.sigmoid(p)
        n_ch = p.shape[1]
        if self.softmax and n_ch > 1:
            p = torch.softmax(p, dim=1)
        t = target
        if not self.include_background and n_ch > 1:
            p, t = p[:, 1:], t[:, 1:]
        axes = list(range(2, p.ndim))
        inter = torch.sum(t * p, dim=axes)
        if self.squared_pred:
            g, q = torch.sum(t ** 2, dim=axes), torch.sum(p ** 2, dim=axes)
        else:
            g, q = torch.sum(t, dim=axes), torch.sum(p, dim=axes)
        den = g + q
        if self.jaccard:
            den = 2.0 * (den - inter)
        f = 1.0 - (2.0 * inter + self.smooth_nr) / (den + self.smooth_dr)
        if self.dice_weight is not None and t.shape[1] != 1 and self.dice_weight.numel() == t.shape[1]:
            f = f * self.dice_weight.to(f)
        return f.mean()

    def ce(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if logits.shape[1] != 1:
            return F.cross_entropy(logits, target.to(logits.dtype), weight=self.weight, reduction="mean")
        return F.binary_cross_entropy_with_logits(logits, target.to(logits.dtype), pos_weight=self.weight,
                                                  reduction="mean")

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if logits.shape != target.shape:
            raise ValueError(f"shape mismatch: logits {tuple(logits.shape)} vs target {tuple(target.shape)}")
        return self.lambda_dice * self.dice(logits, target) + self.lambda_ce * self.ce(logits, target)
