#!/usr/bin/env python
"""Entry point with the call order of the reference's ``main.py`` (reference main.py:39-49:
manager -> setup_model -> setup_data -> ... ), for the one mode this repository implements:
per-volume test-time adaptation + evaluation.

    python main.py task=brats dataset=brats model=unet method=tta_entmin method.steps=10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py task=brats ...

Overrides use the reference's Hydra syntax (group=option, dotted.key=value, +new.key=value).
One process per GPU; volumes are sharded round-robin; one all_gather merges the per-volume table.
"""
from __future__ import annotations

import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per lane (see bench.py); read at HIP start-up


def main(argv=None) -> dict:
    import torch.distributed as dist

    import multimodal_tta_amd  # noqa: F401  (components register themselves, like reference main.py:18-20)
    from multimodal_tta_amd.config import compose, get_config
    from multimodal_tta_amd.registry import get_dataset_builder, get_evaluation_strategy, get_model, DATASET_BUILDERS

    cfg = compose(overrides=list(sys.argv[1:] if argv is None else argv))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("main.py needs an MI355X: the adaptation path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    torch.manual_seed(int(get_config(cfg, "task.seed", 42)))
    # setup_model (reference src/core/experiment_manager.py:88-94)
    model = get_model(cfg.model.name)(cfg.model)
    weights = get_config(cfg, "model.weights", None)
    if weights:
        # CheckpointHook format (reference src/core/hooks.py:55-62); DataParallel prefix stripped
        from multimodal_tta_amd.checkpoint import load_source_weights
        load_source_weights(model, str(weights))
    model.to(device)
    # setup_data('test') (reference :164-196): builder lookup by task name with the "default" fallback (:120-124)
    tname = str(get_config(cfg, "task.name", "default"))
    builder = get_dataset_builder(tname if DATASET_BUILDERS.has(tname) else "default")(cfg)
    loader = builder.get_loader("test", shard=(rank, world))
    kind = str(get_config(cfg, "method.kind", "baseline"))
    strat_name = "seg_tta_eval" if kind == "tta" else str(get_config(cfg, "task.eval_strategy", "seg_eval"))
    metrics = get_evaluation_strategy(strat_name)(cfg).evaluate_epoch(model, loader, device)
    if rank == 0:
        print(json.dumps({"strategy": strat_name, "world_size": world, "metrics": metrics}, indent=1))
    if world > 1:
        dist.destroy_process_group()
    return metrics


if __name__ == "__main__":
    main()
