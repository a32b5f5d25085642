"""The HIP path against the COMMITTED fixtures (tests/golden/*.npz + model_fixtures.json, written by
tests/golden/make_golden.py from the oracle in the build container): nothing under oracle/ is imported here, so this is
the parity check that needs only the files that travel to the GPU box.

Per fixture: seeded weights -> forward logits (5e-4 of max|logits|), entropy loss (1e-5 relative), then ONE adaptation
step at lr 1e-3: final logits (2e-3 of max|logits|: Adam's first step moves every parameter by ~lr, the direction of
parameters whose gradient is rounding noise is not reproducible, they have no influence on the output) and the adapted
`unit0.conv.weight` tensors (within 2.5 * lr: never further than one step plus the weight decay).

Plus section 8f row 1 on the GPU: a CheckpointHook-format file with DataParallel's ``module.`` prefix (reference
src/core/hooks.py:55-62, experiment_manager.py:95-96) -> `load_source_weights` -> adapt, bitwise equal to the same
weights handed over with `load_state_dict`."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def root_cfg(model_name, model_cfg, train_cfg, steps=1):
    from multimodal_tta_amd.config import compose
    cfg = compose(overrides=["task=brats", f"model={model_name}", "method=tta_entmin"])
    cfg["model"] = dict(model_cfg, name=model_name)
    cfg["method"]["steps"] = steps
    cfg["method"]["precision"] = "fp32"
    cfg["training"]["optimizers"]["adam"].update(train_cfg["optimizers"]["adam"])
    cfg["training"]["param_groups"] = dict(train_cfg["param_groups"])
    cfg["training"]["optimizer"] = "adam"
    return cfg


@pytest.mark.parametrize("fixture", ["unet_small", "unet_hecktor_small", "deepfusion_small"])
def test_hip_path_reproduces_the_committed_fixture(fixture):
    from multimodal_tta_amd.registry import get_model, get_plugin

    meta = json.load(open(os.path.join(HERE, "model_fixtures.json")))[fixture]
    z = np.load(os.path.join(HERE, fixture + ".npz"))
    model = get_model(meta["name"])(dict(meta["cfg"], name=meta["name"]))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    assert list(model.state_dict().keys()) == meta["state_dict_keys"]
    model.load_state_dict(sd)
    x = torch.from_numpy(z["x"]).cuda()
    want = torch.from_numpy(z["logits"])
    model.cuda().train()
    with torch.no_grad():
        got = model(x).cpu()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item() / scale
    assert err <= 5e-4, f"{fixture}: forward logits differ by {err:.3e} of max|logits|"

    lr = float(meta["train_cfg"]["optimizers"]["adam"]["lr"])
    plug = get_plugin("entmin_tta")(root_cfg(meta["name"], meta["cfg"], meta["train_cfg"])).setup(model, "cuda")
    res = plug.adapt_volume(x)
    loss = float(res["losses"].cpu()[0])
    assert abs(loss - float(z["loss"])) <= 1e-5 * abs(float(z["loss"])) + 1e-7, (loss, float(z["loss"]))
    adapted = plug.logits(res).cpu()
    want_a = torch.from_numpy(z["adapted_logits"])
    err_a = (adapted - want_a).abs().max().item() / want_a.abs().max().item()
    assert err_a <= 2e-3, f"{fixture}: adapted logits differ by {err_a:.3e}"
    state = {k: v.cpu() for k, v in model.state_dict().items()}
    checked = 0
    for k in z.files:
        if k.startswith("a/"):
            d = (state[k[2:]] - torch.from_numpy(z[k])).abs().max().item()
            assert d <= 2.5 * lr, f"{fixture}: adapted {k[2:]} differs by {d:.3e}"
            moved = (state[k[2:]] - sd[k[2:]]).abs().max().item()
            assert moved > 0.5 * lr, f"{fixture}: {k[2:]} did not adapt"
            checked += 1
    assert checked > 0
    print(f"[golden {fixture}] logits {err:.2e}  adapted logits {err_a:.2e}  loss {loss:.6f}")


def test_checkpoint_file_to_adapted_volume_on_the_gpu(tmp_path):
    from multimodal_tta_amd.checkpoint import load_source_weights
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.models import UNet
    from multimodal_tta_amd.registry import get_plugin
    from multimodal_tta_amd.synth import synth_volume

    mcfg = dict(name="unet", in_channels=4, num_classes=3, spatial_dims=3, channels=[8, 16, 32, 64, 128],
                strides=[2, 2, 2, 2], num_res_units=2, norm="BATCH", act="RELU", dropout=0.0)
    torch.manual_seed(9)
    src = UNet(mcfg)
    with torch.no_grad():                       # a "trained" source: running statistics away from their defaults
        for name, buf in src.named_buffers():
            if name.endswith("running_mean"):
                buf.normal_(0.0, 0.1)
            elif name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
    path = tmp_path / "checkpoints" / "checkpoints" / "best_model.pth"      # the reference's doubled directory (App. D)
    path.parent.mkdir(parents=True)
    torch.save({"epoch": 12, "model_state_dict": {"module." + k: v.clone() for k, v in src.state_dict().items()},
                "optimizer_state_dict": {"state": {}, "param_groups": []}, "best_metrics": {"loss": 0.4}}, path)
    cfg = compose(overrides=["task=brats", "model=unet", "method=tta_entmin", "method.steps=3", f"model.weights={path}"])
    cfg["model"].update(mcfg)
    cfg["training"]["optimizers"]["adam"]["lr"] = 1e-3
    x = synth_volume(4, 4, (32, 32, 32), 3)["image"].unsqueeze(0).cuda()
    a = UNet(mcfg)
    a.load_state_dict(src.state_dict())
    b = UNet(mcfg)
    meta = load_source_weights(b, str(cfg["model"]["weights"]))
    assert meta == {"epoch": 12, "best_metrics": {"loss": 0.4}}
    outs = []
    for m in (a, b):
        plug = get_plugin("entmin_tta")(cfg).setup(m, "cuda")
        res = plug.adapt_volume(x)
        outs.append((plug.logits(res).clone(), res["losses"].clone()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all() and outs[0][1][-1] < outs[0][1][0]
