"""Oracle for the evaluation tail: threshold -> masks -> Dice/IoU -> aggregation.

Test infrastructure.  These functions follow files that ARE in the reference (no third-party
code involved): src/evaluation/seg_eval.py:41-68 (``_binary_dice_iou``), :304-306 (sigmoid,
``>= threshold``, GT ``> 0.5``), :250-270 + :363-378 (float64 accumulation overall and per
domain, empty-GT regions skipped) and :402-460 (finalisation and metric key names).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Sequence, Tuple

import torch


def masks_from_logits(logits: torch.Tensor, label: torch.Tensor, threshold: float) -> Tuple[torch.Tensor, torch.Tensor]:
    prob = torch.sigmoid(logits)
    return (prob >= threshold).to(torch.uint8), (label.float() > 0.5).to(torch.uint8)


def binary_dice_iou(pred: torch.Tensor, gt: torch.Tensor, eps: float = 1e-7):
    """pred, gt: [B,R,D,H,W] in {0,1} -> dice [B,R], iou [B,R], valid [B,R] (all fp32 arithmetic)."""
    assert pred.shape == gt.shape, f"pred {pred.shape} != gt {gt.shape}"
    B, R = pred.shape[:2]
    pf = pred.reshape(B, R, -1).float()
    gf = gt.reshape(B, R, -1).float()
    inter = (pf * gf).sum(-1)
    ps, gs = pf.sum(-1), gf.sum(-1)
    valid = gs > 0
    dice = (2.0 * inter + eps) / (ps + gs + eps)
    iou = (inter + eps) / (ps + gs - inter + eps)
    return dice, iou, valid


class RegionAccumulator:
    """float64 sums/counts per region, overall and per domain string; then the metric dict."""

    def __init__(self, region_order: Sequence[str]):
        self.regions = list(region_order)
        R = len(self.regions)
        z = lambda: torch.zeros(R, dtype=torch.float64)  # noqa: E731
        self.sum_d, self.cnt_d, self.sum_i, self.cnt_i = z(), z(), z(), z()
        self.dom = defaultdict(lambda: [z(), z(), z(), z()])
        self.total_loss, self.n_samples = 0.0, 0

    def add(self, dice: torch.Tensor, iou: torch.Tensor, valid: torch.Tensor, domains: Sequence[str]) -> None:
        B, R = dice.shape
        for i in range(B):
            d = self.dom[domains[i]]
            for c in range(R):
                if bool(valid[i, c].item()):
                    dv, iv = float(dice[i, c].item()), float(iou[i, c].item())
                    self.sum_d[c] += dv
                    self.cnt_d[c] += 1.0
                    self.sum_i[c] += iv
                    self.cnt_i[c] += 1.0
                    d[0][c] += dv
                    d[1][c] += 1.0
                    d[2][c] += iv
                    d[3][c] += 1.0

    def add_loss(self, loss: float, batch: int) -> None:
        self.total_loss += float(loss) * batch
        self.n_samples += batch

    @staticmethod
    def _finalize(s: torch.Tensor, c: torch.Tensor) -> List[float]:
        return [float((s[k] / c[k]).item()) if c[k] > 0 else 0.0 for k in range(len(s))]

    @staticmethod
    def _avg(means: List[float], cnt: torch.Tensor) -> float:
        ok = [k for k in range(len(means)) if cnt[k] > 0]
        return float(sum(means[k] for k in ok) / max(1, len(ok)))

    def metrics(self, report_loss: bool = False) -> Dict[str, float]:
        md, mi = self._finalize(self.sum_d, self.cnt_d), self._finalize(self.sum_i, self.cnt_i)
        out: Dict[str, float] = {}
        for name, v in zip(self.regions, md):
            out[f"{name.lower()}_dc"] = v
        out["avg_dc"] = self._avg(md, self.cnt_d)
        out["miou"] = self._avg(mi, self.cnt_i)
        out["jc"] = out["miou"]
        out["loss"] = float(self.total_loss / max(1, self.n_samples)) if report_loss else 0.0
        for dom in sorted(self.dom.keys()):
            sd, cd, si, ci = self.dom[dom]
            safe = dom if dom != "" else "unknown"
            dm, dim_ = self._finalize(sd, cd), self._finalize(si, ci)
            for name, v in zip(self.regions, dm):
                out[f"dom/{safe}/{name.lower()}_dc"] = v
            out[f"dom/{safe}/avg_dc"] = self._avg(dm, cd)
            out[f"dom/{safe}/miou"] = self._avg(dim_, ci)
        return out
