"""Dataset builders behind the reference's ``register_dataset_builder`` contract (reference
src/datasets/base_builder.py:60-88: ``cls(root_cfg)``, ``get_dataset(split)``,
``get_loader(split, **overrides)``; lookup by ``task.name`` with a ``"default"`` fallback,
src/core/experiment_manager.py:120-124).

Two sources per task, chosen by ``dataset.synthetic.enabled``:

* synthetic volumes (bench, tests; no patient data ships with this repository);
* the reference's on-disk layout: BraTS multi-source ``processed.csv`` files
  (reference src/datasets/brats.py:170-253,328-401) and the single HECKTOR21 ``manifest.csv`` with its dynamic
  leave-one-centre-out split (src/datasets/hecktor21.py:63-298), volumes read by ``nifti.load_canonical``.

Batches carry the reference's keys: image [C,D,H,W] float32 with torch (D,H,W) = array (Z,Y,X), label [R,D,H,W]
float32 in {0,1}, case_id, domain, index (+ profile / center_code / center_id).  The NIfTI datasets hand over RAW
intensities: normalisation runs on the GPU as the evaluator's pre-pass (``transforms.normalize_image``), so
``training.data.transforms.normalize`` is honoured there, not in the loader worker.  Random augmentations of the
train split (reference transforms.py:95-118, MONAI RandRotate90d / RandScaleIntensity / RandShiftIntensity) belong to
source training and are not provided: asking for them raises.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import as_cfg, get_config
from .registry import register_dataset_builder
from .synth import SyntheticSegDataset


class SyntheticSegBuilder:
    def __init__(self, config: Any):
        self.config = as_cfg(config)
        ds = get_config(self.config, "dataset", {}) or {}
        syn = get_config(ds, "synthetic", {}) or {}
        if not bool(get_config(syn, "enabled", True)):
            raise ValueError("SyntheticSegBuilder needs dataset.synthetic.enabled")
        self.num_volumes = int(get_config(syn, "num_volumes", 8))
        self.shape = tuple(int(v) for v in get_config(syn, "shape", [128, 128, 128]))
        self.domain = str(get_config(syn, "domain", "synth"))
        m = get_config(self.config, "model", {}) or {}
        self.channels = int(get_config(m, "in_channels", get_config(m, "num_modalities", 4)))
        self.regions = len(list(get_config(self.config, "evaluation.seg.region_order", ["ET", "TC", "WT"])))
        self.seed = int(get_config(self.config, "task.seed", 42))
        tr = get_config(self.config, "training", {}) or {}
        self.eval_batch_size = int(get_config(tr, "eval_batch_size", 1))
        # synthesising a 4 x 128^3 volume takes ~0.15-0.5 s of host time: without loader workers `main.py` on the synthetic source
        # is bound by it (7 volumes/s end to end against 68 with 12 workers, one MI355X); on a GPU host default to a few
        self.num_workers = int(get_config(tr, "num_workers", 4 if torch.cuda.is_available() else 0))

    def get_dataset(self, split: str = "test", shard: Optional[Tuple[int, int]] = None):
        ds = SyntheticSegDataset(self.num_volumes, self.channels, self.shape, self.regions, self.seed, self.domain)
        if shard is not None and shard[1] > 1:
            rank, world = shard
            ds = torch.utils.data.Subset(ds, list(range(rank, self.num_volumes, world)))
        return ds

    def get_loader(self, split: str = "test", shard: Optional[Tuple[int, int]] = None, **overrides):
        ds = self.get_dataset(split, shard)
        kw: Dict[str, Any] = dict(batch_size=self.eval_batch_size, shuffle=False, drop_last=False,
                                  num_workers=self.num_workers, pin_memory=torch.cuda.is_available())
        kw.update(overrides)
        return torch.utils.data.DataLoader(ds, **kw)


# ----------------------------------------------------------------------------- NIfTI sources
REGION_ORDER = ("ET", "TC", "WT")
DEFAULT_REGION_MAPS = {           # reference src/datasets/brats.py:55-76
    "gli": {"ET": [3], "TC": [1, 3], "WT": [1, 2, 3]},
    "ssa": {"ET": [3], "TC": [1, 3], "WT": [1, 2, 3]},
    "ped": {"ET": [1], "TC": [1, 2, 3], "WT": [1, 2, 3, 4]},
}
_SPLIT_ALIASES = {"validate": "val", "validation": "val", "dev": "val", "train": "train", "test": "test", "val": "val"}


def _normalize_split(split: str) -> str:
    s = _SPLIT_ALIASES.get((split or "").strip().lower(), split)
    if s not in ("train", "val", "test"):
        raise ValueError(f"Unsupported split '{split}'. Allowed: ['test', 'train', 'val']")
    return s


def _resolve_path(path: Any, root_dir: Optional[str]) -> str:
    if path is None or (isinstance(path, float) and np.isnan(path)):
        return ""
    p = str(path)
    if root_dir and not os.path.isabs(p):
        return os.path.join(root_dir, p)
    return p


def _validate_shape(arr: np.ndarray, expected: Optional[Tuple[int, ...]], tag: str, what: str, case_id: str) -> None:
    if expected is not None and tuple(arr.shape) != tuple(expected):
        raise ValueError(f"[{tag}] Shape mismatch for {what} case={case_id}: got {tuple(arr.shape)}, expected "
                         f"{tuple(expected)}. This dataset assumes OFFLINE preprocessing already unified shape.")


def _to_dhw(vol_xyz: np.ndarray) -> torch.Tensor:
    """array (X,Y,Z) -> tensor (D,H,W) = (Z,Y,X)   (reference brats.py:347, hecktor21.py:266)"""
    return torch.from_numpy(np.ascontiguousarray(vol_xyz.transpose(2, 1, 0)))


def _check_spatial(name: str, t: torch.Tensor, spatial: Optional[Tuple[int, int, int]]) -> None:
    if spatial is not None and tuple(int(v) for v in t.shape[-3:]) != tuple(spatial):
        raise ValueError(f"[3DTransforms] {name} spatial mismatch: got {tuple(int(v) for v in t.shape[-3:])}, expected "
                         f"{tuple(spatial)}. This pipeline assumes OFFLINE preprocessing already fixed shapes; "
                         "no online resize/crop/pad is performed.")


def region_masks(y_id: torch.Tensor, region_map: Dict[str, Sequence[int]]) -> torch.Tensor:
    """raw id map [D,H,W] -> [3,D,H,W] float32 in ET, TC, WT order (reference brats.py:132-153)."""
    out = []
    for name in REGION_ORDER:
        m = torch.zeros_like(y_id, dtype=torch.bool)
        for v in region_map.get(name, []):
            m |= (y_id == int(v))
        out.append(m.float())
    return torch.stack(out, 0)


def binary_mask(y: np.ndarray) -> np.ndarray:
    """{0,1} kept, {0,255} -> {0,1}, anything else: non-zero (reference hecktor21.py:50-62)."""
    if y.dtype.kind not in ("i", "u"):
        y = np.rint(y)
    y = y.astype(np.int16, copy=False)
    uniq = np.unique(y)
    if uniq.size == 0 or np.all(np.isin(uniq, [0, 1])):
        return y.astype(np.uint8, copy=False)
    if np.all(np.isin(uniq, [0, 255])):
        return (y // 255).astype(np.uint8, copy=False)
    return (y != 0).astype(np.uint8, copy=False)


def parse_processed_csv(csv_path: str, modality_order: Sequence[str], root_dir: Optional[str], drop_unlabeled: bool,
                        cols: Optional[Dict[str, str]] = None) -> Dict[str, Dict[str, Any]]:
    """One row per (subject, modality) -> cases[subject] = {split, modalities{mod: path}, label}; the first split /
    label of a subject wins; subjects lacking a required modality (or a label) are dropped (reference brats.py:170-253)."""
    import pandas as pd
    c = {"split": "split", "subject": "subject_id", "modality": "modality", "img": "img_path", "label": "label_path"}
    c.update(cols or {})
    df = pd.read_csv(csv_path)
    for k in ("subject", "modality", "img", "split"):
        if c[k] not in df.columns:
            raise ValueError(f"[BraTS-Multi-NIfTI] CSV missing required column '{c[k]}': {csv_path}")
    if c["label"] not in df.columns:
        df[c["label"]] = np.nan
    cases: Dict[str, Dict[str, Any]] = {}
    for _, row in df.iterrows():
        cid = str(row[c["subject"]])
        mod = str(row[c["modality"]]).strip().lower()
        split = str(row[c["split"]]).strip().lower()
        img = _resolve_path(row[c["img"]], root_dir)
        lab = _resolve_path(row[c["label"]], root_dir) if pd.notna(row[c["label"]]) else ""
        if cid not in cases:
            cases[cid] = {"split": split, "modalities": {}, "label": lab}
        elif not cases[cid]["label"] and lab:
            cases[cid]["label"] = lab
        cases[cid]["modalities"][mod] = img
    need = [m.lower() for m in modality_order]
    return {k: v for k, v in cases.items()
            if all(m in v["modalities"] for m in need) and not (drop_unlabeled and not v.get("label"))}


class BratsNiftiDataset(torch.utils.data.Dataset):
    """Multi-source BraTS cases (reference brats.py:260-401)."""

    TAG = "BraTS-Multi-NIfTI"

    def __init__(self, sources: List[Dict[str, Any]], split: str, modality_order: Sequence[str], expected_shape=None,
                 drop_unlabeled: bool = True, strict_label_values: bool = False, image_size=None, cols=None):
        self.split = str(split).lower()
        self.modality_order = [m.lower() for m in modality_order]
        self.expected_shape = tuple(expected_shape) if expected_shape is not None else None
        self.strict_label_values = bool(strict_label_values)
        self.image_size = tuple(int(v) for v in image_size) if image_size is not None else None
        self._index: List[Tuple[Dict[str, Any], str, Dict[str, Any]]] = []
        for src in sources:
            if not os.path.exists(src["csv_path"]):
                raise FileNotFoundError(f"[{self.TAG}] CSV not found: {src['csv_path']}")
            cases = parse_processed_csv(src["csv_path"], self.modality_order, src.get("root_dir"), drop_unlabeled, cols)
            allowed = [str(v).lower() for v in src["include_splits"].get(self.split, [self.split])]
            for cid, info in cases.items():
                if str(info["split"]).lower() in allowed:
                    self._index.append((src, cid, info))
        if not self._index:
            raise ValueError(f"[{self.TAG}] No samples after filtering. split='{self.split}'. "
                             "Check include_splits and CSV 'split' values.")

    def __len__(self) -> int:
        return len(self._index)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        from .nifti import load_canonical
        src, cid, info = self._index[idx]
        vols = []
        for mod in self.modality_order:
            p = info["modalities"][mod]
            if not os.path.exists(p):
                raise FileNotFoundError(f"[{self.TAG}] Missing image file: {p} (case={cid}, mod={mod})")
            v = load_canonical(p, np.float32)
            _validate_shape(v, self.expected_shape, self.TAG, f"image/{mod}", cid)
            vols.append(_to_dhw(v))
        image = torch.stack(vols, 0).float()
        lp = info.get("label", "")
        if not lp or not os.path.exists(lp):
            y_id = torch.zeros(image.shape[1:], dtype=torch.int64)
        else:
            y = load_canonical(lp, np.float32)
            _validate_shape(y, self.expected_shape, self.TAG, "label", cid)
            y_id = _to_dhw(np.rint(y).astype(np.int16)).long()
        if self.strict_label_values:
            bad = [v for v in torch.unique(y_id).tolist() if v < 0 or v > 20]
            if bad:
                raise ValueError(f"[{self.TAG}] Abnormal label values {bad} in case={cid} src={src['name']}. "
                                 "This often indicates non-nearest interpolation in preprocessing.")
        label = region_masks(y_id, src["region_map"])
        _check_spatial("image", image, self.image_size)
        _check_spatial("label", label, self.image_size)
        return {"image": image, "label": label, "case_id": cid, "domain": src["name"], "profile": src["profile"],
                "index": int(idx)}


def sample_val_indices_per_center(index_by_center: Dict[str, np.ndarray], val_per_center: int, seed: int) -> List[int]:
    """k cases per non-target centre, centres in sorted order, one ``RandomState(seed)`` stream across them
    (reference hecktor21.py:65-90): the same split as the reference for the same manifest."""
    if val_per_center <= 0:
        return []
    rng = np.random.RandomState(seed)
    out: List[int] = []
    for c in sorted(index_by_center):
        idxs = index_by_center[c]
        if idxs.size == 0:
            continue
        out.extend(rng.choice(idxs, size=min(val_per_center, int(idxs.size)), replace=False).tolist())
    return out


class Hecktor21NiftiDataset(torch.utils.data.Dataset):
    """One manifest, split on the fly around ``target_center`` (reference hecktor21.py:96-298):
    test = every case of the target centre; val = ``val_per_center`` cases of each other centre; train = the rest."""

    TAG = "HECKTOR21"

    def __init__(self, manifest_csv: str, split: str, target_center: str, val_per_center: int = 5, split_seed: int = 2026,
                 expected_shape=None, drop_unlabeled: bool = True, strict_label_values: bool = True,
                 root_dir: Optional[str] = None, cols: Optional[Dict[str, str]] = None,
                 ok_status_values: Sequence[str] = ("ok",), image_size=None):
        import pandas as pd
        self.split = str(split).lower().strip()
        if self.split not in ("train", "val", "test"):
            raise ValueError(f"[HECKTOR21] split must be in {{train,val,test}}, got '{split}'")
        c = {"patient": "patient_id", "status": "status", "ct": "ct_proc", "pt": "pt_proc", "label": "gtvt_proc",
             "center_code": "center_code", "center_id": "center_id"}
        c.update(cols or {})
        self.c = c
        self.expected_shape = tuple(expected_shape) if expected_shape is not None else None
        self.drop_unlabeled, self.strict_label_values, self.root_dir = bool(drop_unlabeled), bool(strict_label_values), root_dir
        self.image_size = tuple(int(v) for v in image_size) if image_size is not None else None
        if not os.path.exists(manifest_csv):
            raise FileNotFoundError(f"[HECKTOR21] manifest_csv not found: {manifest_csv}")
        df = pd.read_csv(manifest_csv)
        for k in ("patient", "ct", "pt", "center_code"):
            if c[k] not in df.columns:
                raise ValueError(f"[HECKTOR21] manifest missing required column '{c[k]}'")
        if c["label"] not in df.columns:
            df[c["label"]] = np.nan
        if c["status"] not in df.columns:
            df[c["status"]] = "ok"
        df = df[df[c["status"]].astype(str).str.lower().isin({str(v).lower() for v in ok_status_values})].copy()
        if self.drop_unlabeled:
            df = df[df[c["label"]].notna() & (df[c["label"]].astype(str) != "")].copy()
        df[c["center_code"]] = df[c["center_code"]].astype(str).str.upper()
        target = str(target_center).upper().strip()
        if target == "":
            raise ValueError("[HECKTOR21] target_center cannot be empty")
        d_t, d_n = df[df[c["center_code"]] == target], df[df[c["center_code"]] != target]
        if len(d_t) == 0:
            raise ValueError(f"[HECKTOR21] target_center='{target}' has 0 samples after filtering. "
                             "Check center_code values in manifest.")
        if len(d_n) == 0:
            raise ValueError("[HECKTOR21] non-target set is empty; cannot build train/val.")
        by_center = {cc: d_n[d_n[c["center_code"]] == cc].index.to_numpy() for cc in d_n[c["center_code"]].unique()}
        val_idx = sample_val_indices_per_center(by_center, int(val_per_center), int(split_seed))
        use = d_t if self.split == "test" else (d_n.loc[val_idx] if self.split == "val" else d_n.drop(index=val_idx))
        if len(use) == 0:
            raise ValueError(f"[HECKTOR21] split='{self.split}' becomes empty. target_center={target}, "
                             f"val_per_center={val_per_center}.")
        self._rows = [row.to_dict() for _, row in use.iterrows()]

    def __len__(self) -> int:
        return len(self._rows)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        from .nifti import load_canonical
        info, c = self._rows[idx], self.c
        cid = str(info.get(c["patient"]))
        center = str(info.get(c["center_code"], "")).upper()
        try:
            v = info.get(c["center_id"], None)
            center_id = int(v) if (v is not None and str(v) != "nan") else -1
        except Exception:
            center_id = -1
        paths = {k: _resolve_path(info.get(c[k], ""), self.root_dir) for k in ("ct", "pt", "label")}
        if not os.path.exists(paths["ct"]):
            raise FileNotFoundError(f"[HECKTOR21] Missing CT file: {paths['ct']} (case={cid})")
        if not os.path.exists(paths["pt"]):
            raise FileNotFoundError(f"[HECKTOR21] Missing PET file: {paths['pt']} (case={cid})")
        have_label = bool(paths["label"]) and os.path.exists(paths["label"])
        if self.drop_unlabeled and not have_label:
            raise FileNotFoundError(f"[HECKTOR21] Missing label file: {paths['label']} (case={cid})")
        vols = []
        for k in ("ct", "pt"):
            v = load_canonical(paths[k], np.float32)
            _validate_shape(v, self.expected_shape, self.TAG, k, cid)
            vols.append(_to_dhw(v))
        image = torch.stack(vols, 0).float()
        if have_label:
            y = load_canonical(paths["label"], np.float32)
            _validate_shape(y, self.expected_shape, self.TAG, "label", cid)
            y_np = binary_mask(y)
        else:
            y_np = np.zeros(tuple(reversed(image.shape[1:])), dtype=np.uint8)
        label = _to_dhw(y_np.astype(np.float32)).unsqueeze(0)
        if self.strict_label_values:
            uniq = torch.unique(label).tolist()
            if [v for v in uniq if v not in (0.0, 1.0)]:
                raise ValueError(f"[HECKTOR21] Label must be binary {{0,1}}. got={uniq} (case={cid})")
        _check_spatial("image", image, self.image_size)
        _check_spatial("label", label, self.image_size)
        return {"image": image, "label": label, "case_id": cid, "domain": center, "center_code": center,
                "center_id": center_id, "index": int(idx)}


class _NiftiBuilderBase:
    """Loader plumbing of reference base_builder.py:24-109 (batch sizes and worker settings from ``training.*``)."""

    def __init__(self, config: Any):
        self.config = as_cfg(config)
        tr = get_config(self.config, "training", {}) or {}
        self.batch_size = int(get_config(tr, "batch_size", 32))
        self.eval_batch_size = int(get_config(tr, "eval_batch_size", self.batch_size))
        self.num_workers = int(get_config(tr, "num_workers", 4))
        self.pin_memory = bool(get_config(tr, "pin_memory", True))
        self._datasets: Dict[str, Any] = {}
        t = get_config(self.config, "training.data.transforms", {}) or {}
        size = get_config(t, "image_size", None)
        if size is not None and len(list(size)) != 3:
            raise ValueError(f"training.data.transforms.image_size must be [D,H,W], got {list(size)}")
        self.image_size = [int(v) for v in size] if size is not None else None
        self._aug = bool(get_config(t, "geom_aug", False)) or bool(get_config(t, "intensity_aug", False))

    def _check_aug(self, split: str) -> None:
        if split == "train" and self._aug:
            raise NotImplementedError("random train-split augmentations (training.data.transforms.geom_aug / "
                                      "intensity_aug) are source-training features and are not provided")

    def build_dataset(self, split: str, **overrides):
        raise NotImplementedError

    def get_dataset(self, split: str = "test", shard: Optional[Tuple[int, int]] = None, **overrides):
        split = _normalize_split(split)
        if overrides:
            ds = self.build_dataset(split, **overrides)
        else:
            if split not in self._datasets:
                self._datasets[split] = self.build_dataset(split)
            ds = self._datasets[split]
        if ds is not None and shard is not None and shard[1] > 1:
            rank, world = shard
            ds = torch.utils.data.Subset(ds, list(range(rank, len(ds), world)))
        return ds

    def get_loader(self, split: str = "test", shard: Optional[Tuple[int, int]] = None, **overrides):
        split = _normalize_split(split)
        loader_keys = {"batch_size", "num_workers", "pin_memory", "drop_last", "shuffle", "collate_fn", "sampler"}
        ds = self.get_dataset(split, shard, **{k: v for k, v in overrides.items() if k not in loader_keys})
        if ds is None:
            return None
        train = split == "train"
        kw: Dict[str, Any] = dict(batch_size=self.batch_size if train else self.eval_batch_size, shuffle=train,
                                  drop_last=train, num_workers=self.num_workers,
                                  pin_memory=self.pin_memory and torch.cuda.is_available())
        kw.update({k: v for k, v in overrides.items() if k in loader_keys and v is not None})
        return torch.utils.data.DataLoader(ds, **kw)


class BratsNiftiBuilder(_NiftiBuilderBase):
    """reference brats.py:407-574: ``dataset.sources[*]`` = name, csv_path, profile, root_dir, include_splits, region_map."""

    def __init__(self, config: Any):
        super().__init__(config)
        d = get_config(self.config, "dataset", None)
        if d is None:
            raise ValueError("missing config: dataset")
        shape = get_config(d, "expected_shape", None)
        self.expected_shape = tuple(int(v) for v in shape) if shape is not None else None
        self.strict_label_values = bool(get_config(d, "strict_label_values", False))
        srcs = get_config(d, "sources", None)
        if not srcs:
            raise ValueError("[brats_multi_nifti] 'dataset.sources' is required for multi-source loading.")
        self.cols = {k: str(get_config(d, f"{k}_col", v)) for k, v in
                     (("split", "split"), ("subject", "subject_id"), ("modality", "modality"), ("img", "img_path"),
                      ("label", "label_path"))}
        self.sources: List[Dict[str, Any]] = []
        for sc in srcs:
            profile = str(get_config(sc, "profile", "gli")).lower()
            inc = {str(k).lower(): [str(v).lower() for v in list(vals)]
                   for k, vals in dict(get_config(sc, "include_splits", {}) or {}).items()}
            for k in ("train", "val", "test"):
                inc.setdefault(k, [k])
            rmap = get_config(sc, "region_map", None) or DEFAULT_REGION_MAPS.get(profile, DEFAULT_REGION_MAPS["gli"])
            name, csv_path = get_config(sc, "name", None), get_config(sc, "csv_path", None)
            if name is None or csv_path is None:
                raise ValueError("[brats_multi_nifti] every source needs 'name' and 'csv_path'")
            self.sources.append({"name": str(name), "csv_path": str(csv_path), "profile": profile,
                                 "root_dir": get_config(sc, "root_dir", None), "include_splits": inc,
                                 "region_map": {k: [int(x) for x in list(v)] for k, v in dict(rmap).items()}})
        self.modality_order = ("t1n", "t1c", "t2w", "t2f")          # fixed by the reference (brats.py:494)

    def build_dataset(self, split: str, **overrides):
        split = _normalize_split(split)
        if not any(len(s["include_splits"].get(split, [])) > 0 for s in self.sources):
            return None                                             # split disabled for every source (brats.py:504-518)
        self._check_aug(split)
        return BratsNiftiDataset(self.sources, split, self.modality_order,
                                 overrides.get("expected_shape", self.expected_shape), True,
                                 bool(overrides.get("strict_label_values", self.strict_label_values)), self.image_size,
                                 self.cols)


class Hecktor21NiftiBuilder(_NiftiBuilderBase):
    """reference hecktor21.py:304-425."""

    def __init__(self, config: Any):
        super().__init__(config)
        d = get_config(self.config, "dataset", None)
        if d is None:
            raise ValueError("missing config: dataset")
        m = get_config(d, "manifest_csv", None)
        if not isinstance(m, str):
            raise ValueError("missing config: dataset.manifest_csv")
        self.manifest_csv = m
        shape = get_config(d, "expected_shape", None)
        self.expected_shape = tuple(int(v) for v in shape) if shape is not None else None
        t = get_config(d, "target_center", None)
        if t is None:
            raise ValueError("missing config: dataset.target_center")
        self.kw = dict(target_center=str(t), val_per_center=int(get_config(d, "val_per_center", 5)),
                       split_seed=int(get_config(d, "split_seed", 2026)),
                       drop_unlabeled=bool(get_config(d, "drop_unlabeled", True)),
                       strict_label_values=bool(get_config(d, "strict_label_values", True)),
                       root_dir=get_config(d, "root_dir", None))
        self.cols = {k: str(get_config(d, f"{k}_col", v)) for k, v in
                     (("patient", "patient_id"), ("status", "status"), ("ct", "ct_proc"), ("pt", "pt_proc"),
                      ("label", "gtvt_proc"), ("center_code", "center_code"), ("center_id", "center_id"))}
        self.ok_status_values = list(get_config(d, "ok_status_values", ["ok"]))

    def build_dataset(self, split: str, **overrides):
        split = _normalize_split(split)
        self._check_aug(split)
        kw = dict(self.kw)
        for k in list(kw):
            if k in overrides:
                kw[k] = overrides[k]
        return Hecktor21NiftiDataset(self.manifest_csv, split, expected_shape=overrides.get("expected_shape", self.expected_shape),
                                     cols=self.cols, ok_status_values=self.ok_status_values, image_size=self.image_size, **kw)


def _dispatch(nifti_cls):
    def make(config: Any):
        cfg = as_cfg(config)
        syn = get_config(get_config(cfg, "dataset", {}) or {}, "synthetic", {}) or {}
        return SyntheticSegBuilder(cfg) if bool(get_config(syn, "enabled", True)) else nifti_cls(cfg)
    make.__name__ = nifti_cls.__name__
    make.__doc__ = nifti_cls.__doc__
    return make


register_dataset_builder("brats")(_dispatch(BratsNiftiBuilder))
register_dataset_builder("hecktor21")(_dispatch(Hecktor21NiftiBuilder))
register_dataset_builder("default")(SyntheticSegBuilder)
