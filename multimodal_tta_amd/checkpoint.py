"""Source-weight interchange with the reference's checkpoints.

The reference's ``CheckpointHook`` (reference src/core/hooks.py:53-93) writes
``{"epoch", "model_state_dict", "optimizer_state_dict", "best_metrics"[, "scheduler_state_dict"]}`` with
``torch.save``; models wrapped in ``nn.DataParallel`` (reference src/core/experiment_manager.py:95-96) carry a
``module.`` prefix on every key.  The HIP-backed models of this package keep MONAI's ``state_dict`` key names
(tests/golden/state_dict_keys.json), so a reference-trained U-Net loads as is.

Files are read with ``weights_only=True`` only: nothing in a checkpoint is executed.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch

_PREFIX = "module."


def strip_data_parallel_prefix(state_dict: Dict[str, Any]) -> Dict[str, Any]:
    return {(k[len(_PREFIX):] if k.startswith(_PREFIX) else k): v for k, v in state_dict.items()}


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """The model ``state_dict`` of a CheckpointHook file, or of a bare ``state_dict`` file."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"checkpoint not found: {path}")
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt.get("model_state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: neither a CheckpointHook file nor a state_dict")
    return strip_data_parallel_prefix(sd)


def load_source_weights(model: torch.nn.Module, path: str, strict: bool = True) -> Dict[str, Any]:
    """Load source-model weights into a model of this package; returns the checkpoint's bookkeeping fields."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True) if os.path.exists(path) else None
    sd = read_state_dict(path)
    model.load_state_dict(sd, strict=strict)
    meta = {}
    if isinstance(ckpt, dict) and "model_state_dict" in ckpt:
        meta = {"epoch": ckpt.get("epoch", 0), "best_metrics": ckpt.get("best_metrics", {})}
    return meta


def save_checkpoint(path: str, model: torch.nn.Module, epoch: int = 0, best_metrics: Optional[Dict[str, float]] = None,
                    optimizer_state_dict: Optional[Dict[str, Any]] = None) -> None:
    """Write a file the reference's ``CheckpointHook.load_checkpoint`` (reference hooks.py:72-93) accepts."""
    state = {
        "epoch": int(epoch),
        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer_state_dict": optimizer_state_dict if optimizer_state_dict is not None else {"state": {}, "param_groups": []},
        "best_metrics": dict(best_metrics or {}),
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
