"""Evaluation plugins for the adaptation path.

``seg_eval`` is the reference's strategy under its own name (reference
src/evaluation/seg_eval.py:151, contract SURVEY.md section 8b): same constructor (root config),
same ``evaluate_epoch(model, data_loader, device) -> Dict[str, float]``, same metric keys and
float64 aggregation (:250-270, :363-378, :402-460), same errors (:279-302).  What changed is where
the voxel work runs: threshold + the three reductions of ``_binary_dice_iou`` (:41-68, :304-306)
are one HIP kernel returning exact integer counts, so a batch costs ONE device->host copy instead
of the reference's 3*B*R ``.item()`` syncs (:366-368).

``seg_tta_eval`` adapts every volume with the ``entmin_tta`` plugin before scoring it, shards
volumes across ranks (one process per GPU) and merges the per-volume table with a single
``all_gather`` - RCCL over xGMI on an MI355X node, gloo in the CPU tests.
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import ops
from .config import as_cfg, get_config
from .registry import get_plugin, register_evaluation_strategy

EPS = 1e-7


# ----------------------------------------------------------------------------- small host-side pieces
def as_list_str(x: Any, batch_size: int) -> List[str]:
    """batch['domain'] in any collated form -> list[str] of length B (reference seg_eval.py:19-38)."""
    if x is None:
        return [""] * batch_size
    if isinstance(x, (list, tuple)):
        return [str(v) for v in x]
    if isinstance(x, str):
        return [x] * batch_size
    if torch.is_tensor(x):
        if x.ndim == 0:
            return [str(int(x.item()))] * batch_size
        if x.numel() == batch_size:
            return [str(int(v.item())) for v in x.view(-1)]
    return [str(x)] * batch_size


def dice_iou_from_counts(counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """counts int64 [B,R,3] = (inter, pred, gt) -> dice, iou, valid with the reference's fp32 formulas
    (seg_eval.py:55-66).  The counts are exact (< 2**24 per the shipped shapes), so the fp32 values equal
    the reference's sums of {0,1} floats bit for bit."""
    c = counts.to(torch.float32)
    inter, ps, gs = c[..., 0], c[..., 1], c[..., 2]
    valid = gs > 0
    dice = (2.0 * inter + EPS) / (ps + gs + EPS)
    iou = (inter + EPS) / (ps + gs - inter + EPS)
    return dice, iou, valid


class RegionAccumulator:
    """float64 sums / counts per region, overall and per domain (reference seg_eval.py:250-270,363-378)."""

    def __init__(self, region_order: Sequence[str], surface: bool = False):
        self.regions = list(region_order)
        self.surface = bool(surface)
        R = len(self.regions)
        self._z = lambda: torch.zeros(R, dtype=torch.float64)
        # sum_dice, cnt_dice, sum_iou, cnt_iou, then (surface) sum_hd95, cnt_hd95, sum_asd, cnt_asd
        self.tot = [self._z() for _ in range(8)]
        self.dom: Dict[str, List[torch.Tensor]] = defaultdict(lambda: [self._z() for _ in range(8)])
        self.total_loss, self.n_samples = 0.0, 0

    def add_row(self, dice: Sequence[float], iou: Sequence[float], valid: Sequence[bool], domain: str,
                hd95: Optional[Sequence[float]] = None, asd: Optional[Sequence[float]] = None) -> None:
        d = self.dom[domain]
        for c in range(len(self.regions)):
            if bool(valid[c]):
                dv, iv = float(dice[c]), float(iou[c])
                for acc in (self.tot, d):
                    acc[0][c] += dv
                    acc[1][c] += 1.0
                    acc[2][c] += iv
                    acc[3][c] += 1.0
                if self.surface:
                    hv, av = float(hd95[c]), float(asd[c])
                    for acc in (self.tot, d):
                        acc[4][c] += hv
                        acc[5][c] += 1.0
                        acc[6][c] += av
                        acc[7][c] += 1.0

    def add_loss(self, loss: float, batch: int) -> None:
        self.total_loss += float(loss) * batch
        self.n_samples += batch

    @staticmethod
    def _fin(s: torch.Tensor, c: torch.Tensor) -> List[float]:
        return [float((s[k] / c[k]).item()) if c[k] > 0 else 0.0 for k in range(len(s))]

    @staticmethod
    def _avg(means: List[float], cnt: torch.Tensor) -> float:
        ok = [k for k in range(len(means)) if cnt[k] > 0]
        return float(sum(means[k] for k in ok) / max(1, len(ok)))

    def metrics(self, report_loss: bool) -> Dict[str, float]:
        md, mi = self._fin(self.tot[0], self.tot[1]), self._fin(self.tot[2], self.tot[3])
        out: Dict[str, float] = {}
        for name, v in zip(self.regions, md):
            out[f"{name.lower()}_dc"] = v
        out["avg_dc"] = self._avg(md, self.tot[1])
        out["miou"] = self._avg(mi, self.tot[3])
        out["jc"] = out["miou"]
        out["loss"] = float(self.total_loss / max(1, self.n_samples)) if report_loss else 0.0
        if self.surface:        # reference seg_eval.py:424-440
            self._surface_keys(out, "", self.tot)
        for dom in sorted(self.dom.keys()):
            sd, cd, si, ci = self.dom[dom][:4]
            safe = dom if dom != "" else "unknown"
            dm, dim_ = self._fin(sd, cd), self._fin(si, ci)
            for name, v in zip(self.regions, dm):
                out[f"dom/{safe}/{name.lower()}_dc"] = v
            out[f"dom/{safe}/avg_dc"] = self._avg(dm, cd)
            out[f"dom/{safe}/miou"] = self._avg(dim_, ci)
            if self.surface:    # reference seg_eval.py:459-476
                self._surface_keys(out, f"dom/{safe}/", self.dom[dom])
        return out

    def _surface_keys(self, out: Dict[str, float], prefix: str, acc: List[torch.Tensor]) -> None:
        mh, ma = self._fin(acc[4], acc[5]), self._fin(acc[6], acc[7])
        for name, v in zip(self.regions, mh):
            out[f"{prefix}{name.lower()}_hd95"] = v
        out[f"{prefix}avg_hd95"] = self._avg(mh, acc[5])
        for name, v in zip(self.regions, ma):
            out[f"{prefix}{name.lower()}_asd"] = v
        out[f"{prefix}avg_asd"] = self._avg(ma, acc[7])


class DiceCEReport:
    """monai DiceCELoss(sigmoid=True, ...) value for ``report_loss`` from the HIP sums kernel
    (reference seg_eval.py:198-220,395-400)."""

    def __init__(self, crit_cfg: Any):
        crit_cfg = as_cfg(crit_cfg)
        self.include_background = bool(get_config(crit_cfg, "include_background", True))
        self.squared_pred = bool(get_config(crit_cfg, "squared_pred", False))
        self.jaccard = bool(get_config(crit_cfg, "jaccard", False))
        self.lambda_dice = float(get_config(crit_cfg, "lambda_dice", 1.0))
        self.lambda_ce = float(get_config(crit_cfg, "lambda_ce", 1.0))
        w = get_config(crit_cfg, "weight", None)
        self.weight = [float(x) for x in list(w)] if w is not None and len(list(w)) > 0 else None
        self._w_dev: Optional[torch.Tensor] = None

    def launch(self, logits: torch.Tensor, label: torch.Tensor, channels_last: bool = False):
        """Queue the sums kernel on the current stream; ``value`` turns the result into the loss later (one host sync)."""
        B, R = (logits.shape[0], logits.shape[-1]) if channels_last else logits.shape[:2]
        nvox = logits.numel() // (B * R)
        if self.weight is not None and self._w_dev is None:
            if len(self.weight) != R:
                raise ValueError(f"criterion.weight has {len(self.weight)} entries for {R} channels")
            self._w_dev = torch.tensor(self.weight, dtype=torch.float32, device=logits.device)
        out = torch.empty(B * (R * 3 + 1), dtype=torch.float64, device=logits.device)
        ops.dice_ce_sums(logits, label, self._w_dev, self.squared_pred, out, logits_channels_last=channels_last)
        return out, B, R, nvox

    def value(self, pending) -> float:
        out, B, R, nvox = pending
        s = out.cpu().view(B, R * 3 + 1)
        per = s[:, :R * 3].view(B, R, 3).to(torch.float32)
        inter, ps, gs = per[..., 0], per[..., 1], per[..., 2]
        if not self.include_background and R > 1:
            inter, ps, gs = inter[:, 1:], ps[:, 1:], gs[:, 1:]
        den = gs + ps
        if self.jaccard:
            den = 2.0 * (den - inter)
        f = 1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)
        if self.weight is not None and f.shape[1] != 1:
            dw = torch.tensor(self.weight[1:] if not self.include_background else self.weight, dtype=torch.float32)
            if dw.numel() == f.shape[1]:
                f = f * dw
        dice = float(f.mean().item())
        ce = float((s[:, R * 3].sum() / (B * nvox * (R if R == 1 else 1))).item())
        return self.lambda_dice * dice + self.lambda_ce * ce

    def values_per_volume(self, pending) -> List[float]:
        """The same loss for every volume of the batch on its own (reduction 'mean' makes the batch value their mean):
        what a rank contributes per row to the gathered table."""
        out, B, R, nvox = pending
        return [self.value((out.view(B, R * 3 + 1)[b:b + 1].reshape(-1), 1, R, nvox)) for b in range(B)]

    def __call__(self, logits: torch.Tensor, label: torch.Tensor, channels_last: bool = False) -> float:
        return self.value(self.launch(logits, label, channels_last))


# ----------------------------------------------------------------------------- seg_eval
@register_evaluation_strategy("seg_eval")
class SegmentationEvaluationStrategy:
    def __init__(self, config: Any = None):
        self.config = as_cfg(config)
        seg = get_config(self.config, "evaluation.seg", {}) or {}
        self.threshold = float(get_config(seg, "threshold", 0.5))
        self.region_order = list(get_config(seg, "region_order", ["ET", "TC", "WT"]))
        sp = get_config(seg, "spacing", [1.0, 1.0, 1.0])
        sp = list(sp) if sp is not None else [1.0, 1.0, 1.0]
        if len(sp) != 3:
            raise ValueError(f"[BratsSegEval] evaluation.seg.spacing must have length 3, got {sp}")
        self.spacing = tuple(float(v) for v in sp)
        self.report_loss = bool(get_config(self.config, "evaluation.loss.report_loss", False))
        # HD95 / ASD, off by default like the reference (src/evaluation/seg_eval.py:193-196)
        surf = get_config(self.config, "evaluation.surface", {}) or {}
        self.enable_surface = bool(get_config(surf, "enable", False))
        self.asd_symmetric = bool(get_config(surf, "asd_symmetric", False))
        self.loss_fn = DiceCEReport(get_config(self.config, "training.criterion", {}) or {})
        # input pre-pass on the GPU (raw volumes in, the reference's `_normalize_img` applied here instead of in the
        # dataset worker; reference src/datasets/transforms.py:129-223).  The NIfTI datasets of this package hand over
        # raw intensities, so the pre-pass defaults to on for them and to off for the synthetic source (already
        # normalised); `training.data.transforms.normalize_on_device` overrides either way.
        tcfg = get_config(self.config, "training.data.transforms", {}) or {}
        synthetic = bool(get_config(self.config, "dataset.synthetic.enabled", True))
        nod = get_config(tcfg, "normalize_on_device", None)
        self.normalize_on_device = (not synthetic) if nod is None else bool(nod)
        self._tcfg = tcfg

    def prepare_image(self, x: torch.Tensor) -> torch.Tensor:
        if not self.normalize_on_device:
            return x
        from .transforms import normalize_image
        t = self._tcfg
        names = get_config(self.config, "dataset.modality_order", None)
        return torch.stack([normalize_image(x[b].float(), bool(get_config(t, "normalize", True)),
                                            get_config(t, "intensity_policy", None), get_config(t, "mean", None),
                                            get_config(t, "std", None), list(names) if names else None)
                            for b in range(x.size(0))], 0)

    # -- per batch
    def check_batch(self, batch: Dict[str, Any], device) -> Tuple[torch.Tensor, torch.Tensor]:
        x = self.prepare_image(batch["image"].to(device))
        B = x.size(0)
        if "label" not in batch:
            raise KeyError("[BratsSegEval] batch must contain 'label' for region-based eval.")
        y = batch["label"]
        y = y.to(device) if torch.is_tensor(y) else torch.as_tensor(y, device=device)
        if y.ndim == 4:
            y = y.unsqueeze(0).expand(B, -1, -1, -1, -1)
        if y.ndim != 5:
            raise ValueError(f"[BratsSegEval] label must be 5D, got {tuple(y.shape)}")
        R = int(y.size(1))
        if R != len(self.region_order):
            raise ValueError(f"[BratsSegEval] label channels={R} but region_order={len(self.region_order)}")
        return x, y.float()

    def score(self, logits: torch.Tensor, y: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
        """logits [B,R,D,H,W] (or channels-last view) + labels -> exact counts int64 [B,R,3] on the host.
        With ``evaluation.surface.enable`` the prediction mask is kept for :meth:`surface`."""
        R = y.shape[1]
        shape_ok = (logits.ndim == 5 and (logits.shape[-1] if channels_last else logits.shape[1]) == R)
        if not shape_ok:
            raise ValueError(f"[BratsSegEval] model logits must be [B,{R},D,H,W], got {tuple(logits.shape)}")
        counts = torch.empty((y.shape[0], R, 3), dtype=torch.int64, device=y.device)
        self._mask = torch.empty(tuple(y.shape), dtype=torch.uint8, device=y.device) if self.enable_surface else None
        ops.mask_dice_counts(logits, y, self.threshold, counts, self._mask, logits_channels_last=channels_last)
        return counts.cpu()

    def surface_launch(self, mask: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Queue HD95 / ASD of (mask, y) on the current stream -> device tensors [B,R] as MONAI would return them."""
        return ops.surface_distances(mask, y, self.spacing, 95.0, self.asd_symmetric)

    def surface_fix(self, hd: torch.Tensor, asd: torch.Tensor, counts: torch.Tensor, shape) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reference's penalty (GT non-empty, prediction empty -> volume diagonal in mm) and sanitising (non-finite ->
        diagonal) on the valid entries (reference src/evaluation/seg_eval.py:342-355); host tensors out."""
        D, H, W = (int(v) for v in shape)
        sd, sh, sw = self.spacing
        dd, hh, ww = max(D - 1, 0) * sd, max(H - 1, 0) * sh, max(W - 1, 0) * sw
        diag_mm = float(math.sqrt(dd * dd + hh * hh + ww * ww))
        hd, asd = hd.cpu(), asd.cpu()
        valid = counts[..., 2] > 0
        pred_empty = counts[..., 1] == 0
        pen = valid & pred_empty
        hd[pen] = diag_mm
        asd[pen] = diag_mm
        hd[(~torch.isfinite(hd)) & valid] = diag_mm
        asd[(~torch.isfinite(asd)) & valid] = diag_mm
        return hd, asd

    def surface(self, y: torch.Tensor, counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """HD95 / ASD [B,R] (host, fp32) of the masks of the last :meth:`score` call (reference seg_eval.py:312-355)."""
        hd, asd = self.surface_launch(self._mask, y)
        return self.surface_fix(hd, asd, counts, y.shape[2:])

    @torch.no_grad()
    def evaluate_epoch(self, model: torch.nn.Module, data_loader: Iterable, device) -> Dict[str, float]:
        """Single process: the reference's loop and aggregation (seg_eval.py:276-460).  Under an initialised
        ``torch.distributed`` group with more than one rank ``data_loader`` is this rank's shard: every volume becomes a
        row of the per-volume table, the tables are merged (``merge_rank_tables``) and every rank reports the metrics of
        the WHOLE split - a rank never prints the metrics of its shard as if they were the test set."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        model.eval()
        model.to(device)
        acc = RegionAccumulator(self.region_order, self.enable_surface)
        rows: List[torch.Tensor] = []
        domain_names: List[str] = []
        n_local = 0
        for batch in data_loader:
            x, y = self.check_batch(batch, device)
            logits = model(x)
            counts = self.score(logits.float(), y)
            dice, iou, valid = dice_iou_from_counts(counts)
            hd, asd = self.surface(y, counts) if self.enable_surface else (None, None)
            domains = as_list_str(batch.get("domain", None), batch_size=x.size(0))
            if world == 1:
                for i in range(x.size(0)):
                    acc.add_row(dice[i].tolist(), iou[i].tolist(), valid[i].tolist(), domains[i],
                                hd[i].tolist() if hd is not None else None, asd[i].tolist() if asd is not None else None)
                if self.report_loss:
                    acc.add_loss(self.loss_fn(logits.float(), y), x.size(0))
                continue
            losses = (self.loss_fn.values_per_volume(self.loss_fn.launch(logits.float(), y)) if self.report_loss
                      else [0.0] * x.size(0))
            idx = batch.get("index", None)
            if idx is None:
                raise KeyError("[seg_eval] a sharded evaluation needs batch['index'] (the volume's position in the whole split): "
                               "rank-local counters collide across ranks")
            for i in range(x.size(0)):
                if domains[i] not in domain_names:
                    domain_names.append(domains[i])
                parts = [torch.tensor([int(idx[i]), domain_names.index(domains[i]),
                                       losses[i]], dtype=torch.float64),
                         dice[i].double(), iou[i].double(), valid[i].double()]
                if self.enable_surface:
                    parts += [hd[i].double(), asd[i].double()]
                rows.append(torch.cat(parts))
                n_local += 1
        if world == 1:
            return acc.metrics(self.report_loss)
        R = len(self.region_order)
        table = torch.stack(rows) if rows else torch.empty((0, table_width(R, self.enable_surface)), dtype=torch.float64)
        table, domain_names = merge_rank_tables(table, domain_names, device)
        self.last_table = table
        return metrics_from_table(table, self.region_order, domain_names, self.report_loss, self.enable_surface)


# ----------------------------------------------------------------------------- sharding
def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin, deterministic: volume i -> rank i mod W."""
    return list(range(rank, n_items, world))


def table_width(R: int, surface: bool = False) -> int:
    return 3 + (5 if surface else 3) * R  # index, domain_id, loss, then dice[R], iou[R], valid[R] (, hd95[R], asd[R])


def gather_table(rows: torch.Tensor, n_items: int, world: int, group=None) -> torch.Tensor:
    """all_gather the fixed-shape per-volume table [ceil(N/W), width] (float64; unused rows have index -1)
    and return the rows sorted by volume index.  The only collective of the path."""
    import torch.distributed as dist

    per = (n_items + world - 1) // world
    pad = torch.full((per, rows.shape[1]), -1.0, dtype=torch.float64, device=rows.device)
    pad[:rows.shape[0]] = rows
    if not (dist.is_available() and dist.is_initialized()):
        allrows = pad
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        allrows = torch.cat(bufs, dim=0)
    allrows = allrows.cpu()
    allrows = allrows[allrows[:, 0] >= 0]
    order = torch.argsort(allrows[:, 0], stable=True)
    allrows = allrows[order]
    if allrows.shape[0] > 1:          # a sampler that pads the last shard repeats volumes: keep the first row per index
        keep = torch.ones(allrows.shape[0], dtype=torch.bool)
        keep[1:] = allrows[1:, 0] != allrows[:-1, 0]
        allrows = allrows[keep]
    return allrows


def gather_masks(local: Sequence[Tuple[int, torch.Tensor]], device, group=None) -> Dict[int, torch.Tensor]:
    """The optional second collective of a sharded evaluation (north star: "all-gather final
    Dice/logits"): every rank's thresholded masks uint8 [R,D,H,W], keyed by volume index, gathered to every rank.
    One ``all_reduce(MAX)`` agrees on the pad shape (ranks may hold unequal shares and ragged extents), one ``all_gather``
    moves the index / extent rows, one the padded masks (6 MB per 128^3 BraTS volume).  Single process: the local dict."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return {int(i): m for i, m in local}
    world = dist.get_world_size()
    dev = device if dist.get_backend() == "nccl" else "cpu"
    ext = torch.zeros(5, dtype=torch.int64)                   # rows, R, D, H, W: the maxima over ranks
    ext[0] = len(local)
    for _, m in local:
        ext[1:] = torch.maximum(ext[1:], torch.tensor(list(m.shape), dtype=torch.int64))
    ext = ext.to(dev)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=group)
    per, R, D, H, W = (int(v) for v in ext.tolist())
    meta = torch.full((max(per, 1), 5), -1, dtype=torch.int64)
    buf = torch.zeros((max(per, 1), R, D, H, W), dtype=torch.uint8)
    for k, (i, m) in enumerate(local):
        meta[k] = torch.tensor([int(i), *m.shape], dtype=torch.int64)
        buf[k, :m.shape[0], :m.shape[1], :m.shape[2], :m.shape[3]] = m
    meta, buf = meta.to(dev), buf.to(dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    dist.all_gather(bufs, buf, group=group)
    out: Dict[int, torch.Tensor] = {}
    for mt, bf in zip(metas, bufs):
        mt, bf = mt.cpu(), bf.cpu()
        for k in range(mt.shape[0]):
            i, r, d, h, w = (int(v) for v in mt[k].tolist())
            if i >= 0 and i not in out:
                out[i] = bf[k, :r, :d, :h, :w].clone()
    return out


def metrics_from_table(table: torch.Tensor, region_order: Sequence[str], domain_names: Sequence[str],
                       report_loss: bool, surface: bool = False) -> Dict[str, float]:
    """Replay the reference aggregation over gathered rows in volume-index order: the result is
    identical to a single-process run (float64 sums, order fixed by index)."""
    R = len(region_order)
    acc = RegionAccumulator(region_order, surface)
    for row in table:
        dom = domain_names[int(row[1].item())] if 0 <= int(row[1].item()) < len(domain_names) else ""
        dice = row[3:3 + R].to(torch.float32).tolist()
        iou = row[3 + R:3 + 2 * R].to(torch.float32).tolist()
        valid = (row[3 + 2 * R:3 + 3 * R] > 0.5).tolist()
        hd = row[3 + 3 * R:3 + 4 * R].to(torch.float32).tolist() if surface else None
        asd = row[3 + 4 * R:3 + 5 * R].to(torch.float32).tolist() if surface else None
        acc.add_row(dice, iou, valid, dom, hd, asd)
        if report_loss:
            acc.add_loss(float(row[2].item()), 1)
    return acc.metrics(report_loss)


def merge_rank_tables(table: torch.Tensor, domain_names: List[str], device) -> Tuple[torch.Tensor, List[str]]:
    """Merge the ranks' per-volume tables into the table of the whole split, identical on every rank.

    THREE collectives, all outside the data path and all tiny: ``all_gather_object`` of the domain-name lists (domain
    ids must mean the same on every rank), ``all_reduce`` of the row counts (ranks may hold unequal shares), then the
    ``all_gather`` of the fixed-shape table (``gather_table``) - RCCL over xGMI with the ``nccl`` backend, gloo in the
    CPU tests.  Rows come back sorted by volume index, so the aggregation that follows is order-identical to a
    single-process run."""
    import torch.distributed as dist

    world = dist.get_world_size()
    names: List[Optional[List[str]]] = [None] * world
    dist.all_gather_object(names, list(domain_names))
    merged = sorted({n for lst in names for n in (lst or [])})
    remap = {i: merged.index(n) for i, n in enumerate(domain_names)}
    table = table.clone()
    for r in range(table.shape[0]):
        table[r, 1] = remap[int(table[r, 1].item())]
    dev = device if dist.get_backend() == "nccl" else "cpu"
    counts_t = torch.tensor([table.shape[0]], dtype=torch.int64, device=dev)
    per_max = counts_t.clone()
    dist.all_reduce(counts_t)
    dist.all_reduce(per_max, op=dist.ReduceOp.MAX)
    # gather_table pads every rank to ceil(N/W) rows; with an arbitrary (not round-robin) shard the largest share can
    # exceed that, so size the pad by the largest share
    n_items = max(int(counts_t.item()), int(per_max.item()) * world, 1)
    table = gather_table(table.to(dev), n_items, world)
    return table, merged


@register_evaluation_strategy("seg_tta_eval")
class TTASegmentationEvaluationStrategy(SegmentationEvaluationStrategy):
    """Adapt each volume (plugin ``method.name``, default ``entmin_tta``), then score it."""

    def __init__(self, config: Any = None):
        super().__init__(config)
        self.plugin_name = str(get_config(self.config, "method.name", "entmin_tta"))
        # volumes adapted concurrently on this GPU, each with its own weights, buffers, graph and stream: episodic
        # adaptation has no cross-volume state, and one volume alone leaves most CUs waiting at the lower U-Net levels
        # (measured, unet 4x128^3, S = 10: 29.6 / 39.7 / 35.7 volumes/s for 1 / 2 / 3 lanes)
        self.lanes = max(1, int(get_config(self.config, "method.lanes", 1)))
        # volumes that adapt TOGETHER as the batch items of one launch sequence, each on its own replica of the weights
        # (the plugin's `method.group`); lanes x group volumes are in flight per GPU
        self.group = max(1, int(get_config(self.config, "method.group", 1)))
        # north star: "RCCL ... used only to all-gather final Dice/logits": with `evaluation.gather_masks` the thresholded
        # uint8 masks [R,D,H,W] of every volume are gathered to every rank as well (a second all_gather, off by default)
        self.gather_masks = bool(get_config(self.config, "evaluation.gather_masks", False))
        self.local_masks: List[Tuple[int, torch.Tensor]] = []
        self.last_masks: Dict[int, torch.Tensor] = {}
        self.plugin = None
        self.plugins: List[Any] = []
        self.streams: List[Any] = []

    def _setup_lanes(self, model: torch.nn.Module, device) -> None:
        from .registry import get_model
        self.plugin = get_plugin(self.plugin_name)(self.config).setup(model, device)
        self.group = int(getattr(self.plugin, "group", 1))       # what the plugin settled on (1 for models whose norms carry parameters)
        self.plugins, self.streams = [self.plugin], [None]
        if self.lanes > 1:
            pool = ops.lane_streams(self.lanes, device)          # one hardware queue per lane
            self.streams = [pool[0]]
            mcfg = get_config(self.config, "model", None)
            for lane in range(1, self.lanes):
                twin = get_model(str(get_config(mcfg, "name", "unet")))(mcfg)
                twin.load_state_dict(model.state_dict())
                p = get_plugin(self.plugin_name)(self.config)
                p.lane = lane
                self.plugins.append(p.setup(twin, device))
                self.streams.append(pool[lane])

    def _submit(self, lane: int, xb: torch.Tensor, yb: torch.Tensor) -> Dict[str, Any]:
        """Queue adaptation + scoring of one volume - or of one GROUP of volumes (``method.group``, batch items that adapt
        independently) - on the lane's stream; nothing here waits for the GPU."""
        B, R = yb.shape[0], yb.shape[1]
        res = self.plugins[lane].adapt_volume(xb)
        counts = torch.empty((B, R, 3), dtype=torch.int64, device=yb.device)
        mask = torch.empty(tuple(yb.shape), dtype=torch.uint8, device=yb.device) if (self.enable_surface or self.gather_masks) else None
        ops.mask_dice_counts(res["logits_cl"], yb, self.threshold, counts, mask, logits_channels_last=True)
        job: Dict[str, Any] = {"counts": counts, "shape": tuple(yb.shape[2:]), "keep": (xb, yb, mask, res), "mask": mask}
        if self.report_loss:
            job["loss"] = self.loss_fn.launch(res["logits_cl"], yb, channels_last=True)
        if self.enable_surface:
            job["surface"] = self.surface_launch(mask, yb)
        return job

    def _finish(self, job: Dict[str, Any]) -> List[torch.Tensor]:
        """First host read of a queued group -> the table rows of its volumes."""
        counts = job["counts"].cpu()
        dice, iou, valid = dice_iou_from_counts(counts)
        B = counts.shape[0]
        losses = self.loss_fn.values_per_volume(job["loss"]) if self.report_loss else [0.0] * B
        if self.enable_surface:
            hd, asd = self.surface_fix(job["surface"][0], job["surface"][1], counts, job["shape"])
        rows = []
        for b in range(B):
            parts = [torch.tensor([job["index"][b], job["domain_id"][b], losses[b]], dtype=torch.float64),
                     dice[b].double(), iou[b].double(), valid[b].double()]
            if self.enable_surface:
                parts += [hd[b].double(), asd[b].double()]
            rows.append(torch.cat(parts))
        if self.gather_masks:
            mk = job["mask"].cpu()
            for b in range(B):
                self.local_masks.append((int(job["index"][b]), mk[b]))
        return rows

    @torch.no_grad()
    def evaluate_epoch(self, model: torch.nn.Module, data_loader: Iterable, device) -> Dict[str, float]:
        """``data_loader`` yields this rank's shard (any batch size).  ``method.group`` volumes adapt together in one launch
        sequence (each on its own replica of the weights), ``method.lanes`` such groups are in flight on their own streams;
        results are read back when a lane is needed again."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.plugin is None:
            self._setup_lanes(model, device)
        R = len(self.region_order)
        done: List[Tuple[int, torch.Tensor]] = []          # (submission order, row)
        pending: List[Optional[Dict[str, Any]]] = [None] * self.lanes
        domain_names: List[str] = []
        self.local_masks = []
        n_local = 0
        n_groups = 0
        group: List[Tuple[torch.Tensor, torch.Tensor, int, int, int]] = []      # (x1, y1, order, index, domain id)
        keep_alive: List[torch.Tensor] = []

        def flush(lane: int) -> None:
            job = pending[lane]
            if job is not None:
                if self.streams[lane] is not None:
                    self.streams[lane].synchronize()         # host reads below are issued from the main stream
                for order, row in zip(job["order"], self._finish(job)):
                    done.append((order, row))
                pending[lane] = None

        def dispatch() -> None:
            nonlocal n_groups
            lane = n_groups % self.lanes
            flush(lane)                                      # the lane's buffers are about to be reused
            xb = torch.cat([g[0] for g in group]) if len(group) > 1 else group[0][0]
            yb = torch.cat([g[1] for g in group]) if len(group) > 1 else group[0][1]
            stream = self.streams[lane]
            if stream is None:
                job = self._submit(lane, xb, yb)
            else:
                stream.wait_stream(torch.cuda.current_stream(device))      # inputs were prepared on the main stream
                with torch.cuda.stream(stream):
                    job = self._submit(lane, xb, yb)
                for t in (xb, yb, *keep_alive):
                    t.record_stream(stream)
            job.update(order=[g[2] for g in group], index=[g[3] for g in group], domain_id=[g[4] for g in group])
            pending[lane] = job
            n_groups += 1
            group.clear()
            keep_alive.clear()

        for batch in data_loader:
            x, y = self.check_batch(batch, device)
            domains = as_list_str(batch.get("domain", None), batch_size=x.size(0))
            idx = batch.get("index", None)
            if idx is None and world > 1:
                raise KeyError("[seg_tta_eval] a sharded evaluation needs batch['index'] (the volume's position in the whole "
                               "split): rank-local counters collide across ranks")
            keep_alive += [x, y]
            for i in range(x.size(0)):
                if domains[i] not in domain_names:
                    domain_names.append(domains[i])
                group.append((x[i:i + 1], y[i:i + 1], n_local, int(idx[i]) if idx is not None else n_local,
                              domain_names.index(domains[i])))
                n_local += 1
                if len(group) == self.group:
                    dispatch()
        if group:
            dispatch()
        for lane in range(self.lanes):
            flush(lane)
        rows = [row for _, row in sorted(done, key=lambda t: t[0])]
        table = torch.stack(rows) if rows else torch.empty((0, table_width(R, self.enable_surface)), dtype=torch.float64)
        if world > 1:
            table, domain_names = merge_rank_tables(table, domain_names, device)
        self.last_table = table
        if self.gather_masks:
            self.last_masks = gather_masks(self.local_masks, device)
        return metrics_from_table(table, self.region_order, domain_names, self.report_loss, self.enable_surface)
