"""``entmin_tta``: per-test-volume entropy-minimisation adaptation, registered in the reference's
free PLUGINS slot (reference src/registry.py:66,94-96; selected through the existing ``method``
config group, reference configs/config.yaml:7).

The reference ships no adaptation code; the semantics are this repo's (SURVEY.md
Appendix C, executable form: oracle/tta.py).  The step skeleton is the reference's supervised step
(src/core/trainers/seg_trainer.py:105-145): forward -> loss -> backward -> optimizer.step, with
the entropy objective as the loss, Adam built with the reference's decay / no-decay split
(src/core/experiment_manager.py:199-237, configs/training/default.yaml:30-56), then the
reference's evaluation tail (src/evaluation/seg_eval.py:300-308).

MI355X shape of the loop: the whole step (weight repack, forward, fused loss+gradient, backward,
fused Adam over the flat arena) is ONE captured hipGraph replayed ``steps`` times; nothing
returns to the host inside a volume except the final per-region counts (one small copy).
"""
from __future__ import annotations

import warnings
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .config import as_cfg, get_config
from .models.base import DEFAULT_NO_DECAY_KEYS, HipSegModel
from .ops import MmttaError
from .registry import register_plugin


def select_params(model: torch.nn.Module, spec) -> List[str]:
    """Names of the parameters that adapt: 'all' | 'norm_affine' | substring or list of substrings."""
    names = [n for n, _ in model.named_parameters()]
    if spec is None or spec == "all":
        return names
    if spec == "norm_affine":
        return [n for n in names if ".adn.N." in n]
    pats = [spec] if isinstance(spec, str) else list(spec)
    return [n for n in names if any(s in n for s in pats)]


def modality_mask(num_modalities: int, missing: Sequence[int], p_drop: float,
                  gen: Optional[torch.Generator]) -> List[bool]:
    """present[m] for one step: permanently missing channels plus a seeded Bernoulli(p) drop of each
    present one; at least one modality always survives."""
    miss = set(int(i) for i in missing)
    present = [m not in miss for m in range(num_modalities)]
    if p_drop > 0.0 and gen is not None:
        u = torch.rand(num_modalities, generator=gen)
        dropped = [present[m] and bool(u[m] < p_drop) for m in range(num_modalities)]
        if all(dropped[m] or not present[m] for m in range(num_modalities)):
            first = next(m for m in range(num_modalities) if present[m])
            dropped[first] = False
        present = [present[m] and not dropped[m] for m in range(num_modalities)]
    return present


def drop_modality(x: torch.Tensor, present: Sequence[bool]) -> torch.Tensor:
    """Zero the absent channels (0 is the background value of the data: reference src/datasets/brats.py:7)."""
    if all(present):
        return x
    keep = torch.tensor([1.0 if p else 0.0 for p in present], dtype=x.dtype, device=x.device)
    return x * keep.view(1, -1, 1, 1, 1)


@register_plugin("entmin_tta")
class EntropyMinimizationTTA:
    """Construct with the ROOT config (like the reference's evaluation strategies,
    src/core/experiment_manager.py:369-370), then ``setup(model, device)`` once and
    ``adapt_volume(x)`` per test volume."""

    def __init__(self, config: Any = None):
        cfg = as_cfg(config)
        self.cfg = cfg
        m = get_config(cfg, "method", {}) or {}
        self.steps = int(get_config(m, "steps", 10))
        self.episodic = bool(get_config(m, "episodic", True))
        self.params_spec = get_config(m, "params", "all")
        self.precision = str(get_config(m, "precision", "fp32")).lower()
        self.storage = str(get_config(m, "storage", "bf16")).lower()      # activation storage of bf16 precision
        self.grad_storage = str(get_config(m, "grad_storage", "bf16")).lower()      # activation-gradient storage, likewise
        self.missing = [int(i) for i in (get_config(m, "missing_modalities", []) or [])]
        md = get_config(m, "moddrop", {}) or {}
        self.moddrop_p = float(get_config(md, "p", 0.0)) if bool(get_config(md, "enabled", False)) else 0.0
        self.moddrop_seed = int(get_config(md, "seed", 0))
        self.use_graph = bool(get_config(m, "use_graph", True))
        # volumes adapted side by side as the batch items of ONE launch sequence, each with its own replica of the weights
        # and optimizer state (mmtta_param_sets): a volume alone fills a quarter of the chip or less at the lower levels
        self.group = max(1, int(get_config(m, "group", 1)))
        self.lanes = max(1, int(get_config(m, "lanes", 1)))      # the evaluator's lanes: volumes in flight = lanes x group
        # launch geometry (split-K, weight-gradient slabs) is chosen for this many volumes in flight; `auto` = lanes x group.
        # Two runs agree BIT FOR BIT when they use the same figure (the geometry fixes the summation order), whatever their
        # lanes / group are - which is how the grouped arrangement is checked against one volume at a time
        tv = get_config(m, "tune_volumes", "auto")
        self.tune_volumes = None if tv in (None, "auto") else max(1, int(tv))
        self.side_streams = int(get_config(m, "side_streams", 0))   # 0: weight gradients stay on the main stream
        tr = get_config(cfg, "training", {}) or {}
        # the reference's factory (src/core/experiment_manager.py:199-237): `training.optimizer` names the class
        # (default "sgd" there, "adam" in the shipped configs/training/default.yaml:11), `training.optimizers.<name>` holds
        # its arguments, `training.{learning_rate,weight_decay,momentum}` are the fall-backs
        opt_name = str(get_config(tr, "optimizer", "sgd")).lower()
        if opt_name not in ops.OPTIMIZERS:
            raise ValueError(f"Unsupported optimizer: {opt_name}")
        oc = get_config(tr, f"optimizers.{opt_name}", {}) or {}
        if bool(get_config(oc, "amsgrad", False)):
            raise NotImplementedError("amsgrad")
        if bool(get_config(oc, "maximize", False)):
            raise NotImplementedError("maximize")
        betas = get_config(oc, "betas", [0.9, 0.999])
        self.optim = ops.OptimSpec(
            name=opt_name,
            lr=float(get_config(oc, "lr", get_config(tr, "learning_rate", 1e-3))),
            beta1=float(betas[0]), beta2=float(betas[1]),
            eps=float(get_config(oc, "eps", 1e-8)),
            # the reference's parameter groups always carry a weight decay (`optimizers.<name>.weight_decay`, else
            # `training.weight_decay`, else 0), so torch's own AdamW default of 1e-2 never applies
            weight_decay=float(get_config(oc, "weight_decay", get_config(tr, "weight_decay", 0.0))),
            momentum=float(get_config(oc, "momentum", get_config(tr, "momentum", 0.0))) if opt_name == "sgd" else 0.0,
            dampening=float(get_config(oc, "dampening", 0.0)) if opt_name == "sgd" else 0.0,
            nesterov=bool(get_config(oc, "nesterov", False)) if opt_name == "sgd" else False)
        if self.optim.nesterov and (self.optim.momentum <= 0.0 or self.optim.dampening != 0.0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")      # torch.optim.SGD's message
        rules = get_config(tr, "param_groups", {}) or {}
        self.no_decay_keys = list(get_config(rules, "no_decay_keys", []))
        self.treat_1d = bool(get_config(rules, "treat_1d_as_no_decay", True))
        crit = get_config(tr, "criterion", {}) or {}
        self.softmax = bool(get_config(crit, "softmax", False))
        if self.precision not in ops.PRECISIONS:
            raise ValueError(f"method.precision={self.precision}: expected one of {sorted(ops.PRECISIONS)}")
        self.model: Optional[HipSegModel] = None
        self.rt = None
        self._graphs: Dict[Tuple, torch.cuda.CUDAGraph] = {}
        self._gen: Optional[torch.Generator] = None
        self.lane = 0      # plugins that adapt different volumes concurrently (own stream each) take distinct lanes

    # ------------------------------------------------------------------ setup
    def setup(self, model: HipSegModel, device) -> "EntropyMinimizationTTA":
        if not isinstance(model, HipSegModel):
            raise TypeError("entmin_tta drives the HIP-backed models of this package (registered as 'unet', "
                            "'unet_multimodal_deepfusion', 'unet_multimodal_midfusion')")
        device = torch.device(device)
        self.model = model
        ops.tune_for_volumes_in_flight(self.tune_volumes or self.lanes * self.group)      # launch geometry for that many volumes
        names = select_params(model, self.params_spec)
        model.set_precision(self.precision, self.storage, self.grad_storage)
        model.set_group(self.group)
        model.configure_training(set(names), self.no_decay_keys, self.treat_1d)
        model.to(device)
        try:
            self.rt = model.runtime(device)
        except NotImplementedError as exc:
            if self.group == 1:
                raise
            # norm layers with parameters (BatchNorm / affine norms) are per-model, not per-volume: such a model adapts one
            # volume per launch sequence; the lanes still keep several volumes in flight
            warnings.warn(f"method.group = {self.group} -> 1: {exc}")
            self.group = 1
            model.set_group(1)
            self.rt = model.runtime(device)
        self.group = int(self.rt.group)          # what the runtime supports (a runtime without per-volume sets reports 1)
        self.rt.overlap_wgrad = self.side_streams > 0
        self.rt.n_side = max(1, self.side_streams)
        self.rt.arena.snapshot_source()
        self.rt.snapshot_buffers()
        self._graphs.clear()
        return self

    # ------------------------------------------------------------------ one step
    def _step_launches(self, x_cl: torch.Tensor, present: Optional[Sequence[bool]]) -> None:
        rt, ar = self.rt, self.rt.arena
        ops.Workspace.lane = self.lane
        rt.training = True
        rt.use_sets = rt.group > 1          # batch item g reads / writes parameter replica g
        try:
            rt.pack_all()
            logits = rt.forward_cl(x_cl) if present is None else rt.forward_cl(x_cl, present=present)
            n, d, h, w, r = logits.shape
            # (the categorical objective writes fp32 only)
            gdt = rt.thin_grad_dtype() if (not self.softmax and r <= 4) else torch.float32
            dlogits = rt.pool.cl("dlogits", n, d, h, w, r, ldc=(r + 3) // 4 * 4, dtype=gdt)
            if rt.group > 1:
                # every volume of the group is its own objective (own mean, own gradient scale): loss [group]
                partial = rt.pool.flat("ent_partial", ops.entropy_partials_items(logits), dtype=torch.float64)
                loss = rt.pool.flat("ent_loss", rt.group)
                ops.entropy_loss_items(logits, dlogits, partial, loss, softmax=self.softmax)
            else:
                partial = rt.pool.flat("ent_partial", ops.entropy_partials(logits), dtype=torch.float64)
                loss = rt.pool.flat("ent_loss", 1)
                ops.entropy_loss(logits, dlogits, partial, loss, softmax=self.softmax)
            if ar.n_train > 0:
                rt.run_backward(dlogits)
                self.optimizer_step(n)
        finally:
            rt.use_sets = False

    def optimizer_step(self, volumes: int = 1) -> None:
        """The arena optimizer: ONE launch over [decay | no-decay] of every replica in use (+ the device step counter)."""
        ar = self.rt.arena
        if ar.replicas > 1:
            ops.optim_step_sets(self.optim, ar.params_all, ar.grads_all, ar.exp_avg_all, ar.exp_avg_sq_all, ar.n_train, ar.n_decay,
                                min(volumes, ar.replicas), ar.step)
            return
        ops.optim_step(self.optim, ar.params[:ar.n_train], ar.grads[:ar.n_train], ar.exp_avg[:ar.n_train],
                       ar.exp_avg_sq[:ar.n_train], ar.n_decay, ar.step)

    # hyper-parameters under their old attribute names (tests, scripts)
    lr = property(lambda self: self.optim.lr)
    beta1 = property(lambda self: self.optim.beta1)
    beta2 = property(lambda self: self.optim.beta2)
    eps = property(lambda self: self.optim.eps)
    weight_decay = property(lambda self: self.optim.weight_decay)

    def _step(self, x_cl: torch.Tensor, present: Optional[Sequence[bool]]) -> None:
        if not self.use_graph:
            self._step_launches(x_cl, present)
            return
        key = (tuple(x_cl.shape), x_cl.data_ptr(), None if present is None else tuple(present))
        g = self._graphs.get(key)
        if g is None:
            # eager warm-up (allocates every buffer, sizes the workspace), then capture.  Both run on the CALLER's stream
            # when that is not the default stream: a lane keeps its one stream (= its hardware queue, ops.lane_streams)
            # for everything, no helper streams are created per capture
            cur = torch.cuda.current_stream(x_cl.device)
            on_default = cur == torch.cuda.default_stream(x_cl.device)
            work = ops.helper_stream(x_cl.device) if on_default else cur
            if on_default:
                work.wait_stream(cur)
            with torch.cuda.stream(work):
                self._step_launches(x_cl, present)
            if on_default:
                cur.wait_stream(work)
            torch.cuda.synchronize()
            try:
                ops.Workspace.frozen = True
                ops.Workspace.captured.add((x_cl.device.index if x_cl.device.index is not None else torch.cuda.current_device(),
                                            self.lane))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=work):
                    self._step_launches(x_cl, present)
            except Exception as exc:  # capture is an optimisation: the eager launches are the same kernels
                warnings.warn(f"hipGraph capture failed ({exc}); running the step eagerly")
                self.use_graph = False
                g = None
            finally:
                ops.Workspace.frozen = False
            self._graphs[key] = g
            return  # the warm-up already performed this step
        g.replay()

    # ------------------------------------------------------------------ per volume
    @torch.no_grad()
    def adapt_volume(self, x: torch.Tensor, steps: Optional[int] = None) -> Dict[str, Any]:
        """x: [1,C,D,H,W] fp32 on the plugin's device - or, with ``method.group`` = G > 1, up to G volumes [B,C,D,H,W]
        that adapt independently (volume b on parameter replica b).  Returns the final logits (channels-last view
        [B,D,H,W,R]) and the per-step losses ([steps], or [steps, B] for a group)."""
        if self.rt is None:
            raise MmttaError("call setup(model, device) first")
        rt, ar = self.rt, self.rt.arena
        steps = self.steps if steps is None else int(steps)
        B = int(x.shape[0])
        if rt.group > 1 and B > rt.group:
            raise ValueError(f"method.group = {rt.group}: at most {rt.group} volumes per call, got {B}")
        if self.episodic:
            ar.restore_source()
            rt.restore_buffers()
        C = x.shape[1]
        masked = bool(self.missing) or self.moddrop_p > 0.0
        gen = torch.Generator().manual_seed(self.moddrop_seed) if self.moddrop_p > 0.0 else None
        base_present = modality_mask(C, self.missing, 0.0, None)
        x = x.float()
        wants_present = masked and getattr(rt, "supports_present", False)
        # a runtime that applies the mask while its first level stages the input (models/unet.py) reads the volume as it is;
        # the others get the masked volume re-staged whenever the mask changes
        restage = masked and not getattr(rt, "input_mask_on_load", False)
        x_cl = rt.stage_input(drop_modality(x, base_present) if restage else x)
        grouped = rt.group > 1
        loss_hist = rt.pool.flat("loss_hist", max(steps, 1) * (B if grouped else 1))
        loss_buf = rt.pool.flat("ent_loss", rt.group if grouped else 1)
        if grouped:
            loss_hist = loss_hist.view(max(steps, 1), B)
        for t in range(steps):
            present = None
            if masked:
                p = modality_mask(C, self.missing, self.moddrop_p, gen)
                if self.moddrop_p > 0.0 and restage:
                    rt.stage_input(drop_modality(x, p))
                present = p if wants_present else None
            self._step(x_cl, present)
            if grouped:
                loss_hist[t].copy_(loss_buf[:B])
            else:
                loss_hist[t:t + 1].copy_(loss_buf)
        if masked and self.moddrop_p > 0.0 and restage:
            rt.stage_input(drop_modality(x, base_present))
        rt.training = False
        ops.Workspace.lane = self.lane
        rt.use_sets = grouped
        try:
            rt.pack_all()
            logits_cl = (rt.forward_cl(x_cl, present=base_present) if wants_present else rt.forward_cl(x_cl))
        finally:
            rt.use_sets = False
        losses = loss_hist[:steps]
        return {"logits_cl": logits_cl, "losses": losses[:, 0] if (grouped and B == 1) else losses}

    def logits(self, result: Dict[str, Any]) -> torch.Tensor:
        return ops.from_cl(result["logits_cl"])
