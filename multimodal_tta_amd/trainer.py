"""``seg_supervised_step``: the reference's supervised step (reference src/core/trainers/seg_trainer.py:97-145:
``zero_grad -> model(x) -> DiceCELoss -> backward -> optimizer.step``) on the native engine, registered in the
PLUGINS table.

The loss and its gradient come from two HIP kernels (``mmtta_dice_ce_sums`` + ``mmtta_dice_ce_grad``), the
backward and the optimizer are the ones the adaptation loop uses; no autograd graph is built.  The criterion keys
are the reference TRAINER's (``training.criterion.{lambda_dice,lambda_ce,include_background,squared_pred,jaccard,
ce_weight}`` with ``include_background`` defaulting to False, reference seg_trainer.py:33-48,59-79; the evaluator
reads ``weight`` and defaults ``include_background`` to True instead, reference seg_eval.py:80,201 - that asymmetry
is the reference's and is mirrored).  Both heads of the reference's switch (:41-54): ``sigmoid: true`` (multilabel
region masks, every shipped config) and ``softmax: true`` (Dice on softmax probabilities; labels are one-hot /
probability maps [B,R,...], or class indices [B,1,...] with ``to_onehot_y``, the reference's default for softmax heads).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from . import ops
from .config import as_cfg, get_config
from .models.base import HipSegModel
from .ops import MmttaError
from .registry import register_plugin
from .tta import EntropyMinimizationTTA, select_params


@register_plugin("seg_supervised_step")
class SupervisedSegStep(EntropyMinimizationTTA):
    """Shares construction (optimizer hyper-parameters, parameter groups, precision) with the adaptation plugin;
    only the objective differs."""

    def __init__(self, config: Any = None):
        super().__init__(config)
        self.group = 1          # a training batch shares ONE weight set (method.group is the adaptation plugin's: one replica per volume)
        crit = get_config(as_cfg(config), "training.criterion", {}) or {}
        # mode switches exactly as the reference reads them (seg_trainer.py:41-54)
        self.softmax = bool(get_config(crit, "softmax", False))
        self.sigmoid = bool(get_config(crit, "sigmoid", not self.softmax))
        self.to_onehot_y = bool(get_config(crit, "to_onehot_y", self.softmax))
        if self.softmax and self.sigmoid:
            raise ValueError("[SegTrainer] Invalid config: softmax=True and sigmoid=True cannot both be True.")
        if (not self.softmax) and (not self.sigmoid):
            raise ValueError("[SegTrainer] Invalid config: both softmax and sigmoid are False. Please set one True.")
        self.include_background = bool(get_config(crit, "include_background", False))       # seg_trainer.py:33
        self.squared_pred = bool(get_config(crit, "squared_pred", False))
        self.jaccard = bool(get_config(crit, "jaccard", False))
        self.lambda_dice = float(get_config(crit, "lambda_dice", 1.0))
        self.lambda_ce = float(get_config(crit, "lambda_ce", 1.0))
        w = get_config(crit, "ce_weight", None)                                              # seg_trainer.py:48
        self.ce_weight = [float(v) for v in list(w)] if w is not None and len(list(w)) > 0 else None
        self._w_dev: Optional[torch.Tensor] = None

    def loss_value(self, sums: torch.Tensor, B: int, R: int, nvox: int) -> float:
        """Host arithmetic from the sums (the reference syncs once per step as well: ``loss.item()``, :145)."""
        s = sums.cpu().view(B, R * 3 + 1)
        per = s[:, :R * 3].view(B, R, 3).to(torch.float32)
        inter, ps, gs = per[..., 0], per[..., 1], per[..., 2]
        if not self.include_background and R > 1:
            inter, ps, gs = inter[:, 1:], ps[:, 1:], gs[:, 1:]
        den = gs + ps
        if self.jaccard:
            den = 2.0 * (den - inter)
        f = 1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)
        if self.ce_weight is not None and f.shape[1] != 1:
            dw = torch.tensor(self.ce_weight[1:] if not self.include_background else self.ce_weight, dtype=torch.float32)
            if dw.numel() == f.shape[1]:
                f = f * dw
        ce = float((s[:, R * 3].sum() / (B * nvox)).item())
        return self.lambda_dice * float(f.mean().item()) + self.lambda_ce * ce

    @torch.no_grad()
    def run_step(self, batch: Dict[str, Any]) -> Dict[str, float]:
        """One supervised step on ``batch = {"image": [B,C,D,H,W], "label": [B,R,D,H,W]}`` (reference :97-145)."""
        if self.rt is None:
            raise MmttaError("call setup(model, device) first")
        rt, ar = self.rt, self.rt.arena
        dev = rt.device
        x = batch["image"].to(dev).float()
        y = batch["label"].to(dev).float().contiguous()
        if y.ndim == 4:
            y = y.unsqueeze(0).expand(x.size(0), -1, -1, -1, -1).contiguous()
        if self.to_onehot_y and y.shape[1] == 1 and self.rt.out_channels > 1:
            # class-index labels [B,1,...] -> one-hot [B,R,...] (monai to_onehot_y).  With class weights monai's CE term then
            # takes the index-target form, whose 'mean' divides by the summed weights of the targets - not built
            if self.ce_weight is not None:
                raise NotImplementedError("training.criterion.to_onehot_y with ce_weight (index-target weighted mean)")
            y = torch.nn.functional.one_hot(y[:, 0].long(), self.rt.out_channels).permute(0, 4, 1, 2, 3).float().contiguous()
        ops.Workspace.lane = self.lane
        rt.training = True
        rt.pack_all()
        logits = rt.forward_cl(rt.stage_input(x))
        n, d, h, w, r = logits.shape
        if tuple(y.shape) != (n, r, d, h, w):
            raise ValueError(f"[SegTrainer] model logits must be [B,{y.shape[1]},D,H,W], got {(n, r, d, h, w)}")
        if self.ce_weight is not None and self._w_dev is None:
            if len(self.ce_weight) != r:
                raise ValueError(f"criterion.ce_weight has {len(self.ce_weight)} entries for {r} channels")
            self._w_dev = torch.tensor(self.ce_weight, dtype=torch.float32, device=dev)
        sums = rt.pool.flat("dce_sums", n * (r * 3 + 1), dtype=torch.float64)
        ops.dice_ce_sums(logits, y, self._w_dev, self.squared_pred, sums, logits_channels_last=True, softmax=self.softmax)
        dlogits = rt.pool.cl("dlogits", n, d, h, w, r, ldc=(r + 3) // 4 * 4)
        ops.dice_ce_grad(logits, y, self._w_dev, self.squared_pred, self.jaccard, self.include_background,
                         self.lambda_dice, self.lambda_ce, sums, dlogits, logits_channels_last=True, softmax=self.softmax)
        if ar.n_train > 0:
            rt.run_backward(dlogits)
            self.optimizer_step()
        return {"loss": self.loss_value(sums, n, r, d * h * w)}
