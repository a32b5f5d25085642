"""Oracle for the optimizer on the adaptation path (test infrastructure).

Follows reference src/core/experiment_manager.py:199-237 (``_build_optimizer_for``: two
parameter groups - decay / no-decay - where a parameter is no-decay when its *name* contains
any of ``training.param_groups.no_decay_keys`` or, with ``treat_1d_as_no_decay``, when it is
1-D) with the defaults of reference configs/training/default.yaml:30-56
(Adam lr 1e-5, weight_decay 5e-4 coupled L2, betas (0.9, 0.9999), eps 1e-8).
``adam_reference_step`` spells out torch.optim.Adam's update for
per-kernel parity tests.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Iterable, List, Sequence, Tuple

import torch

from .unet import cfg_get

DEFAULT_NO_DECAY_KEYS = ("bias", "bn", "norm", "LayerNorm")


def split_param_groups(named_params: Iterable[Tuple[str, torch.nn.Parameter]], weight_decay: float,
                       no_decay_keys: Sequence[str] = DEFAULT_NO_DECAY_KEYS, treat_1d: bool = True):
    decay, no_decay = [], []
    for n, p in named_params:
        if not p.requires_grad:
            continue
        nd = any(k in n for k in no_decay_keys) or (treat_1d and p.ndim == 1)
        (no_decay if nd else decay).append(p)
    return [
        {"params": decay, "weight_decay": float(weight_decay)},
        {"params": no_decay, "weight_decay": 0.0},
    ]


def build_adam(named_params, train_cfg: Dict[str, Any]) -> torch.optim.Adam:
    """``train_cfg`` is the ``training`` node of the composed config."""
    named_params = list(named_params)
    opt_cfg = cfg_get(train_cfg, "optimizers.adam", {}) or {}
    rules = cfg_get(train_cfg, "param_groups", {}) or {}
    wd = float(cfg_get(opt_cfg, "weight_decay", cfg_get(train_cfg, "weight_decay", 0.0)))
    groups = split_param_groups(
        named_params, wd,
        no_decay_keys=list(cfg_get(rules, "no_decay_keys", [])),
        treat_1d=bool(cfg_get(rules, "treat_1d_as_no_decay", True)),
    )
    groups = [g for g in groups if len(g["params"]) > 0]
    betas = cfg_get(opt_cfg, "betas", [0.9, 0.999])
    return torch.optim.Adam(
        groups,
        lr=float(cfg_get(opt_cfg, "lr", cfg_get(train_cfg, "learning_rate", 1e-3))),
        betas=(float(betas[0]), float(betas[1])),
        eps=float(cfg_get(opt_cfg, "eps", 1e-8)),
        amsgrad=bool(cfg_get(opt_cfg, "amsgrad", False)),
    )


def build_optimizer(named_params, train_cfg: Dict[str, Any]) -> torch.optim.Optimizer:
    """The reference's optimizer factory (reference src/core/experiment_manager.py:199-237): ``training.optimizer`` in
    {sgd, adam, adamw} (default "sgd"), keyword arguments from ``training.optimizers.<name>`` restricted to the
    class's allowed set, ``lr`` / ``momentum`` / ``weight_decay`` fall back to ``training.*``; the decay / no-decay
    groups as in ``build_adam``."""
    named_params = list(named_params)
    name = str(cfg_get(train_cfg, "optimizer", "sgd")).lower()
    space = {
        "sgd": (torch.optim.SGD, {"lr", "momentum", "weight_decay", "dampening", "nesterov"}),
        "adam": (torch.optim.Adam, {"lr", "betas", "eps", "weight_decay", "amsgrad"}),
        "adamw": (torch.optim.AdamW, {"lr", "betas", "eps", "weight_decay", "amsgrad"}),
    }
    if name not in space:
        raise ValueError(f"Unsupported optimizer: {name}")
    cls, allowed = space[name]
    opt_cfg = cfg_get(train_cfg, f"optimizers.{name}", {}) or {}
    rules = cfg_get(train_cfg, "param_groups", {}) or {}
    wd = float(cfg_get(opt_cfg, "weight_decay", cfg_get(train_cfg, "weight_decay", 0.0)))
    groups = split_param_groups(named_params, wd, no_decay_keys=list(cfg_get(rules, "no_decay_keys", [])),
                                treat_1d=bool(cfg_get(rules, "treat_1d_as_no_decay", True)))
    groups = [g for g in groups if len(g["params"]) > 0]
    kwargs = {k: opt_cfg[k] for k in allowed if k in opt_cfg}
    kwargs.pop("weight_decay", None)          # the groups carry it (reference: group values override the default)
    if "betas" in kwargs:
        kwargs["betas"] = (float(kwargs["betas"][0]), float(kwargs["betas"][1]))
    for k in ("lr", "eps", "momentum", "dampening"):
        if k in kwargs:
            kwargs[k] = float(kwargs[k])
    if "lr" not in kwargs:
        kwargs["lr"] = float(cfg_get(opt_cfg, "lr", cfg_get(train_cfg, "learning_rate", 1e-3)))
    if name == "sgd" and "momentum" not in kwargs:
        kwargs["momentum"] = float(cfg_get(opt_cfg, "momentum", cfg_get(train_cfg, "momentum", 0.0)))
    return cls(groups, **kwargs)


def adam_reference_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float, wd: float):
    """One torch.optim.Adam (amsgrad=False, coupled L2) update on fp32 tensors; returns (p, m, v)."""
    g = g + wd * p if wd != 0.0 else g
    m = m + (1.0 - beta1) * (g - m)                      # == lerp(m, g, 1-beta1)
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
