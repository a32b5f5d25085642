"""Oracle of the per-volume adaptation loop (test infrastructure; BUILD-DEFINED semantics).

The reference has no adaptation loop.  The step skeleton is the reference's
supervised step, src/core/trainers/seg_trainer.py:97-145 (zero_grad -> model(x) -> loss ->
backward -> optimizer.step -> loss.item()), with the loss swapped for the entropy objective,
followed by the reference's evaluation tail, src/evaluation/seg_eval.py:300-308.  The exact
definition is SURVEY.md Appendix C; this function is its executable specification.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from .adam import build_optimizer
from .losses import entropy_loss


def select_params(model: torch.nn.Module, spec: Union[str, Sequence[str]] = "all") -> List[Tuple[str, torch.nn.Parameter]]:
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if spec == "all":
        return named
    if spec == "norm_affine":
        return [(n, p) for n, p in named if ".adn.N." in n]
    pats = [spec] if isinstance(spec, str) else list(spec)
    return [(n, p) for n, p in named if any(s in n for s in pats)]


def modality_mask(num_modalities: int, missing: Sequence[int], p_drop: float, gen: Optional[torch.Generator]) -> List[bool]:
    """present[m] for one step: never-present ``missing`` channels plus a seeded Bernoulli(p) drop of
    each present one; at least one modality always survives (the first present one)."""
    present = [m not in set(int(i) for i in missing) for m in range(num_modalities)]
    if p_drop > 0.0 and gen is not None:
        u = torch.rand(num_modalities, generator=gen)
        dropped = [present[m] and bool(u[m] < p_drop) for m in range(num_modalities)]
        if all(dropped[m] or not present[m] for m in range(num_modalities)):
            first = next(m for m in range(num_modalities) if present[m])
            dropped[first] = False
        present = [present[m] and not dropped[m] for m in range(num_modalities)]
    return present


def apply_modality_mask(x: torch.Tensor, present: Sequence[bool]) -> torch.Tensor:
    if all(present):
        return x
    keep = torch.tensor([1.0 if p else 0.0 for p in present], dtype=x.dtype, device=x.device)
    return x * keep.view(1, -1, 1, 1, 1)


def adapt_volume(
    model: torch.nn.Module,
    x: torch.Tensor,
    train_cfg: Dict[str, Any],
    steps: int = 10,
    params: Union[str, Sequence[str]] = "all",
    softmax: bool = False,
    episodic: bool = True,
    missing: Sequence[int] = (),
    moddrop_p: float = 0.0,
    moddrop_seed: int = 0,
    masked_means: bool = False,
) -> Dict[str, Any]:
    """Adapt ``model`` on one volume ``x`` [1,C,D,H,W]; returns final logits and per-step losses.

    ``masked_means`` forwards the per-step ``present`` mask to models that accept it (deep fusion).
    """
    source = copy.deepcopy(model.state_dict()) if episodic else None
    named = select_params(model, params)
    chosen = {id(p) for _, p in named}
    frozen = []
    for p in model.parameters():
        if id(p) not in chosen and p.requires_grad:
            p.requires_grad_(False)
            frozen.append(p)
    # the reference's factory, unconditionally: `training.optimizer` selects the class and a config that names none gets the
    # factory's own default (sgd, reference src/core/experiment_manager.py:199-210) - the same default the adaptation plugin
    # applies, so oracle and product can never adapt with different optimizers
    opt = build_optimizer(named, train_cfg) if named else None
    gen = torch.Generator().manual_seed(int(moddrop_seed)) if moddrop_p > 0.0 else None
    C = x.shape[1]
    losses: List[float] = []
    model.train()
    for _ in range(int(steps)):
        present = modality_mask(C, missing, moddrop_p, gen)
        xin = apply_modality_mask(x, present)
        if opt is not None:
            opt.zero_grad()
        z = model(xin, present=present) if masked_means else model(xin)
        loss = entropy_loss(z, softmax=softmax)
        if opt is not None:
            loss.backward()
            if masked_means:
                # a branch that is absent this step contributes a ZERO gradient (its Adam moments and the
                # coupled weight decay still advance) - torch would otherwise skip parameters whose grad is None
                for _, p in named:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
            opt.step()
        losses.append(float(loss.item()))
    model.eval()
    with torch.no_grad():
        present = modality_mask(C, missing, 0.0, None)
        xin = apply_modality_mask(x, present)
        logits = model(xin, present=present) if masked_means else model(xin)
    adapted = copy.deepcopy(model.state_dict())
    for p in frozen:
        p.requires_grad_(True)
    if episodic:
        model.load_state_dict(source)
    return {"logits": logits, "losses": losses, "adapted_state": adapted}
