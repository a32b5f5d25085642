"""nn.Module facade shared by the HIP-backed segmentation models.

Keeps the reference's module contract (``forward(x[B,C,D,H,W]) ->
logits[B,R,D,H,W]``, ``.train()/.eval()``, ``named_parameters()``, ``state_dict()`` with MONAI
key names, ``.to(device)``) while every FLOP runs in libmmtta.so.  Parameters are adopted into
the engine's flat arena on first use on a device; ``module.to()`` or a fresh
``load_state_dict`` simply trigger a re-adoption.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Set

import torch
import torch.nn as nn

from .. import ops
from ..engine import Runtime
from ..ops import MmttaError

DEFAULT_NO_DECAY_KEYS = ("bias", "bn", "norm", "LayerNorm")


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are engine launch sequences."""

    @staticmethod
    def forward(ctx, x, model, *params):
        rt = model.runtime(x.device)
        logits_cl = rt.run_forward(x)
        ctx.model = model
        ctx.rt = rt
        ctx.nparams = len(params)
        return ops.from_cl(logits_cl)

    @staticmethod
    def backward(ctx, dlogits):
        rt = ctx.rt
        n, c, d, h, w = dlogits.shape
        g_cl = rt.pool.cl("dlogits_in", n, d, h, w, c, ldc=(c + 3) // 4 * 4)
        ops.to_cl(dlogits.contiguous().float(), out=g_cl)
        rt.run_backward(g_cl)
        grads = [r.grad for r in rt.refs if r.trainable]
        assert len(grads) == ctx.nparams
        return (None, None, *grads)


class HipSegModel(nn.Module):
    runtime_cls = None  # set by subclasses

    def __init__(self):
        super().__init__()
        self._rt: Optional[Runtime] = None
        self._trainable: Optional[Set[str]] = None
        self._no_decay_keys: Sequence[str] = DEFAULT_NO_DECAY_KEYS
        self._treat_1d = True
        self.conv_dtype = ops.F32
        self.group = 1

    def set_group(self, group: int) -> None:
        """``group`` volumes adapt side by side through one launch sequence, each with its own parameter replica
        (``method.group``; takes effect at the next runtime build)."""
        group = max(1, int(group))
        if group != self.group:
            self.group = group
            self._rt = None

    # ---- engine binding
    def set_precision(self, precision: str, storage: Optional[str] = None, grad_storage: Optional[str] = None) -> None:
        """'fp32': exact fp32 MFMA everywhere, fp32 storage (the parity path).  'bf16': forward, input-gradient and
        27-tap weight-gradient convolutions round their operands to bf16 and accumulate in fp32; norms, loss, the
        optimizer and the master weights stay fp32.  ``storage`` (bf16 precision only): 'bf16' (default) keeps the wide
        forward activations in HBM as bf16 - what torch autocast does - where the model's runtime supports it;
        'fp32' keeps every tensor fp32 (round-1 behaviour)."""
        if precision not in ops.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(ops.PRECISIONS)}, got {precision!r}")
        storage = "bf16" if storage is None else str(storage).lower()
        if storage not in ("bf16", "fp32"):
            raise ValueError(f"storage must be 'bf16' or 'fp32', got {storage!r}")
        grad_storage = "bf16" if grad_storage is None else str(grad_storage).lower()
        if grad_storage not in ("bf16", "fp32"):
            raise ValueError(f"grad_storage must be 'bf16' or 'fp32', got {grad_storage!r}")
        dt = ops.PRECISIONS[precision]
        if (dt != self.conv_dtype or storage != getattr(self, "act_storage", "bf16")
                or grad_storage != getattr(self, "grad_storage", "bf16")):
            self.conv_dtype = dt
            self.act_storage = storage
            self.grad_storage = grad_storage      # bf16: the wide activation gradients too (where activations are bf16-stored)
            self._rt = None

    def configure_training(self, trainable: Optional[Set[str]] = None,
                           no_decay_keys: Sequence[str] = DEFAULT_NO_DECAY_KEYS, treat_1d: bool = True) -> None:
        """Choose which parameters adapt (None = all) and the decay / no-decay split
        (reference src/core/experiment_manager.py:214-228).  Takes effect at the next runtime build."""
        self._trainable = None if trainable is None else set(trainable)
        self._no_decay_keys = tuple(no_decay_keys)
        self._treat_1d = bool(treat_1d)
        self._rt = None

    def runtime(self, device: torch.device) -> Runtime:
        device = torch.device(device)
        if device.type != "cuda":
            raise MmttaError("this model computes on an MI355X through libmmtta.so; move inputs to a cuda device "
                             "(no CPU / PyTorch fallback exists)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        rt = self._rt
        if rt is not None and rt.device == device and rt.arena is not None and rt.arena.owns():
            return rt
        rt = self.runtime_cls(self, device)
        rt.assign_groups(self._trainable, self._no_decay_keys, self._treat_1d)
        rt.build_arena()
        self._rt = rt
        return rt

    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        if x.dim() != 5:
            raise ValueError(f"expected input [B,C,D,H,W], got {tuple(x.shape)}")
        x = x.float()
        rt = self.runtime(x.device)
        rt.training = self.training
        params = [r.param for r in rt.refs if r.trainable]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _ModelFn.apply(x, self, *params)
        return ops.from_cl(rt.run_forward(x))
