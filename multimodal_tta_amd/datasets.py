"""Dataset builders behind the reference's ``register_dataset_builder`` contract (reference
src/datasets/base_builder.py:60-88: ``cls(root_cfg)``, ``get_dataset(split)``,
``get_loader(split, **overrides)``; lookup by ``task.name`` with a ``"default"`` fallback,
src/core/experiment_manager.py:120-124).

Only the synthetic source is implemented: the NIfTI / CSV readers of the reference
(src/datasets/brats.py, hecktor21.py) are CPU I/O outside the adaptation hot path (SURVEY.md
section 8f, row 4) and no patient data ships with this repository.  Batches carry exactly the
reference's keys: image, label, domain, case_id, index.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

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
            raise NotImplementedError(
                "only dataset.synthetic is implemented (NIfTI/CSV loading: SURVEY.md section 8f row 4, out of scope)")
        self.num_volumes = int(get_config(syn, "num_volumes", 8))
        self.shape = tuple(int(v) for v in get_config(syn, "shape", [128, 128, 128]))
        self.domain = str(get_config(syn, "domain", "synth"))
        m = get_config(self.config, "model", {}) or {}
        self.channels = int(get_config(m, "in_channels", get_config(m, "num_modalities", 4)))
        self.regions = len(list(get_config(self.config, "evaluation.seg.region_order", ["ET", "TC", "WT"])))
        self.seed = int(get_config(self.config, "task.seed", 42))
        tr = get_config(self.config, "training", {}) or {}
        self.eval_batch_size = int(get_config(tr, "eval_batch_size", 1))
        self.num_workers = int(get_config(tr, "num_workers", 0))

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


for _name in ("brats", "hecktor21", "default"):
    register_dataset_builder(_name)(SyntheticSegBuilder)
