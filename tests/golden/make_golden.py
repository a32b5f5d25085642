#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/.

Run in the build container (needs /root/reference for the registry capture; everything else comes
from the oracle, which restates the reference path with torch CPU - see oracle/__init__.py for why
parity is otherwise unpinned).  Outputs are DATA only (JSON / npz): inputs and expected outputs.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def registry_behaviour():
    """Behaviour of the reference's importable src.registry."""
    sys.path.insert(0, REF)
    import importlib

    reg = importlib.import_module("src.registry")
    out = {"tables": [], "helpers": sorted(n for n in dir(reg) if n.startswith(("register_", "get_", "list_")))}
    for name in ("MODELS", "DATASETS", "DATASET_BUILDERS", "EVALUATION_STRATEGIES", "CRITERIA", "PROVIDERS", "PLUGINS"):
        out["tables"].append([name, getattr(reg, name).name])
    r = reg.Registry("demo")

    @r.register("a")
    class A:  # noqa: D401
        pass

    ret = r.register("b", A)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r.register("a", int)
    out["duplicate_warning"] = buf.getvalue()
    out["direct_call_returns_object"] = ret is A
    out["after_duplicate_get_a"] = r.get("a").__name__
    out["list_all"] = r.list_all()
    out["has"] = [r.has("a"), r.has("zzz")]
    try:
        r.get("zzz")
    except KeyError as e:
        out["keyerror_args"] = list(e.args)
    r.clear()
    out["after_clear"] = r.list_all()
    out["list_all_components_keys"] = list(reg.list_all_components().keys())
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    return out


def dice_kats():
    """Hand-computable Dice cases (reference src/evaluation/seg_eval.py:41-68): empty GT -> invalid,
    empty prediction -> dice = 1e-7/(g+1e-7), perfect overlap, half overlap."""
    import oracle

    pred = torch.zeros(1, 4, 2, 2, 2, dtype=torch.uint8)
    gt = torch.zeros(1, 4, 2, 2, 2, dtype=torch.uint8)
    gt[0, 1].view(-1)[:3] = 1                       # region 1: GT 3 voxels, empty prediction
    pred[0, 2] = 1; gt[0, 2] = 1                    # region 2: perfect (8 voxels)
    pred[0, 3].view(-1)[:4] = 1; gt[0, 3].view(-1)[2:6] = 1   # region 3: 4 vs 4, overlap 2
    d, i, v = oracle.binary_dice_iou(pred, gt)
    return {"pred": pred.numpy().tolist(), "gt": gt.numpy().tolist(), "dice": d.tolist(), "iou": i.tolist(),
            "valid": v.tolist()}


def model_fixture(name, cfg, shape, seed=42):
    """Seeded default-init weights (as float32 arrays keyed by state_dict name), a seeded input, the
    forward logits, the entropy loss and the parameters after ONE adaptation step."""
    import oracle

    torch.manual_seed(seed)
    model = oracle.MODELS[name](cfg)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(shape, generator=g)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    logits = model(x)
    loss = oracle.entropy_loss(logits)
    train_cfg = {"optimizer": "adam",      # named explicitly: a config without the key gets the reference factory's sgd
                 "optimizers": {"adam": {"lr": 1e-3, "weight_decay": 5e-4, "betas": [0.9, 0.9999], "eps": 1e-8}},
                 "param_groups": {"no_decay_keys": ["bias", "bn", "norm", "LayerNorm"], "treat_1d_as_no_decay": True}}
    res = oracle.adapt_volume(model, x, train_cfg, steps=1, episodic=True)
    arrays = {"x": x.numpy(), "logits": logits.detach().numpy(), "loss": np.float32(loss.item()),
              "adapted_logits": res["logits"].numpy()}
    for k, v in sd0.items():
        arrays["w/" + k] = v.numpy()
    for k, v in res["adapted_state"].items():
        if v.dtype.is_floating_point and "unit0.conv.weight" in k:
            arrays["a/" + k] = v.numpy()     # a few adapted tensors are enough to pin the optimizer
    return arrays, {"name": name, "cfg": cfg, "shape": list(shape), "train_cfg": train_cfg,
                    "state_dict_keys": list(sd0.keys()), "state_dict_shapes": [list(v.shape) for v in sd0.values()]}


def default_key_lists():
    """state_dict keys + shapes of both models at the shipped default configuration."""
    import oracle

    out = {}
    brats = dict(in_channels=4, num_classes=3, spatial_dims=3, channels=[32, 64, 128, 256, 512], strides=[2, 2, 2, 2],
                 num_res_units=2, norm="INSTANCE", act="RELU", dropout=0.0)
    for name, cfg in (("unet", brats), ("unet_multimodal_deepfusion", dict(brats, num_modalities=4))):
        m = oracle.MODELS[name](cfg)
        out[name] = {"keys": list(m.state_dict().keys()), "shapes": [list(v.shape) for v in m.state_dict().values()],
                     "params": sum(p.numel() for p in m.parameters())}
    return out


def main():
    with open(os.path.join(HERE, "registry_behaviour.json"), "w") as fh:
        json.dump(registry_behaviour(), fh, indent=1)
    with open(os.path.join(HERE, "dice_kats.json"), "w") as fh:
        json.dump(dice_kats(), fh, indent=1)
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as fh:
        json.dump(default_key_lists(), fh, indent=1)
    small = dict(in_channels=4, num_classes=3, spatial_dims=3, channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2],
                 num_res_units=2, norm="INSTANCE", act="RELU", dropout=0.0)
    metas = {}
    arrays, metas["unet_small"] = model_fixture("unet", small, (1, 4, 32, 32, 32))
    np.savez_compressed(os.path.join(HERE, "unet_small.npz"), **arrays)
    hk = dict(small, in_channels=2, num_classes=1)
    arrays, metas["unet_hecktor_small"] = model_fixture("unet", hk, (1, 2, 16, 48, 48))
    np.savez_compressed(os.path.join(HERE, "unet_hecktor_small.npz"), **arrays)
    df = dict(small, num_modalities=4, channels=[2, 4, 8, 16, 32])
    df.pop("in_channels")
    arrays, metas["deepfusion_small"] = model_fixture("unet_multimodal_deepfusion", df, (1, 4, 32, 32, 32))
    np.savez_compressed(os.path.join(HERE, "deepfusion_small.npz"), **arrays)
    with open(os.path.join(HERE, "model_fixtures.json"), "w") as fh:
        json.dump(metas, fh, indent=1)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
