"""The CPU oracle against its committed golden fixtures and against independent torch formulations.

The reference holds no tests or vectors for this path (parity unpinned), so what
is pinned here is (a) the oracle against the fixtures generated from it (regression), (b) each oracle
piece against a second, independent formulation (torch.optim.Adam, autograd of the closed-form entropy,
hand-computed Dice cases), (c) the state_dict key lists of SURVEY.md Appendix A.6.
"""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_param_counts_and_key_lists():
    g = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    assert g["unet"]["params"] == 19223961                 # 19.22 M
    assert g["unet_multimodal_deepfusion"]["params"] == 83075815   # 83.07 M
    keys = g["unet"]["keys"]
    assert keys[0] == "model.0.conv.unit0.conv.weight" and "model.0.residual.weight" in keys
    assert "model.1.submodule.1.submodule.1.submodule.1.submodule.conv.unit1.conv.weight" in keys   # bottom
    assert "model.2.0.conv.weight" in keys and "model.2.1.conv.unit0.conv.weight" in keys
    i = keys.index("model.2.0.conv.weight")
    assert g["unet"]["shapes"][i] == [64, 3, 3, 3, 3]      # ConvTranspose3d weight is [Cin, Cout, k, k, k]
    dk = g["unet_multimodal_deepfusion"]["keys"]
    for k in ("specific_encoders.3.layers.4.residual.weight", "fusion_layer.fusion_conv.conv.weight",
              "bottleneck_reduce.weight", "decoder_stages.3.upsample.preconv.bias",
              "decoder_stages.0.conv.residual.weight", "final_conv.bias", "domain_classifier.weight"):
        assert k in dk, k
    assert "bottleneck_reduce.bias" not in dk


@pytest.mark.parametrize("name", ["unet_small", "unet_hecktor_small", "deepfusion_small"])
def test_model_fixture_reproduces(name):
    meta = json.load(open(os.path.join(GOLD, "model_fixtures.json")))[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    model = oracle.MODELS[meta["name"]](meta["cfg"])
    assert list(model.state_dict().keys()) == meta["state_dict_keys"]
    model.load_state_dict({k: torch.from_numpy(z["w/" + k]) for k in meta["state_dict_keys"]})
    x = torch.from_numpy(z["x"])
    model.train()
    logits = model(x)
    assert torch.allclose(logits, torch.from_numpy(z["logits"]), rtol=1e-4, atol=1e-5)
    assert abs(oracle.entropy_loss(logits).item() - float(z["loss"])) < 1e-5
    res = oracle.adapt_volume(model, x, meta["train_cfg"], steps=1)
    assert torch.allclose(res["logits"], torch.from_numpy(z["adapted_logits"]), rtol=1e-3, atol=1e-4)
    # episodic: weights restored
    for k in meta["state_dict_keys"]:
        assert torch.equal(model.state_dict()[k], torch.from_numpy(z["w/" + k]))


def test_dice_known_answers():
    g = json.load(open(os.path.join(GOLD, "dice_kats.json")))
    pred, gt = torch.tensor(g["pred"], dtype=torch.uint8), torch.tensor(g["gt"], dtype=torch.uint8)
    d, i, v = oracle.binary_dice_iou(pred, gt)
    assert d.tolist() == g["dice"] and i.tolist() == g["iou"] and v.tolist() == g["valid"]
    assert v.tolist() == [[False, True, True, True]]
    assert abs(d[0, 1].item() - 1e-7 / (3 + 1e-7)) < 1e-12        # empty prediction
    assert abs(d[0, 2].item() - 1.0) < 1e-7                        # perfect overlap
    assert abs(d[0, 3].item() - (2 * 2 + 1e-7) / (8 + 1e-7)) < 1e-7 and abs(i[0, 3].item() - 2 / 6) < 1e-6


def test_region_accumulator_keys_and_empty_gt_gating():
    acc = oracle.RegionAccumulator(["ET", "TC", "WT"])
    dice = torch.tensor([[0.5, 0.25, 1.0], [0.7, 0.0, 0.0]])
    valid = torch.tensor([[True, True, False], [True, False, False]])
    acc.add(dice, dice * 0.5, valid, ["a", ""])
    acc.add_loss(2.0, 2)
    m = acc.metrics(report_loss=True)
    assert m["et_dc"] == pytest.approx(0.6) and m["tc_dc"] == pytest.approx(0.25) and m["wt_dc"] == 0.0
    assert m["avg_dc"] == pytest.approx((0.6 + 0.25) / 2)            # regions without a valid sample are left out
    assert m["jc"] == m["miou"] and m["loss"] == 2.0
    assert "dom/a/avg_dc" in m and "dom/unknown/et_dc" in m and m["dom/unknown/et_dc"] == pytest.approx(0.7)
    assert acc.metrics(report_loss=False)["loss"] == 0.0


def test_entropy_closed_form_gradients():
    z = (torch.randn(2, 3, 4, 5, 6, dtype=torch.float64) * 3).requires_grad_(True)
    oracle.bernoulli_entropy_loss(z.float()).backward
    L = (F.softplus(z) - z * torch.sigmoid(z)).mean()
    (g,) = torch.autograd.grad(L, z)
    s = torch.sigmoid(z)
    assert torch.allclose(g, -z * s * (1 - s) / z.numel(), atol=1e-15)
    logp = F.log_softmax(z, dim=1)
    H = -(logp.exp() * logp).sum(1)
    (g2,) = torch.autograd.grad(H.mean(), z)
    want = -(logp.exp() * (logp + H.unsqueeze(1))) / (z.numel() // z.shape[1])
    assert torch.allclose(g2, want, atol=1e-15)
    assert abs(oracle.entropy_loss(z.float(), softmax=True).item() - H.mean().item()) < 1e-6
    assert abs(oracle.entropy_loss(torch.zeros(1, 3, 2, 2, 2)).item() - math.log(2.0)) < 1e-7


def test_adam_reference_step_equals_torch():
    torch.manual_seed(0)
    p0, wd, lr, b1, b2, eps = torch.randn(50), 5e-4, 1e-3, 0.9, 0.9999, 1e-8
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    q, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for t in range(1, 4):
        g = torch.randn(50)
        p.grad = g.clone()
        opt.step()
        q, m, v = oracle.adam_reference_step(q, g, m, v, t, lr, b1, b2, eps, wd)
        assert torch.allclose(q, p.detach(), rtol=1e-6, atol=1e-7)


def test_param_group_split_matches_reference_rules():
    """reference src/core/experiment_manager.py:214-228 with configs/training/default.yaml:54-56."""
    cfg = dict(in_channels=4, num_classes=3, channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2], num_res_units=2,
               norm="BATCH", act="RELU")
    m = oracle.UNet(cfg)
    groups = oracle.split_param_groups(m.named_parameters(), 5e-4, ["bias", "bn", "norm", "LayerNorm"], True)
    decay, no_decay = groups
    assert decay["weight_decay"] == 5e-4 and no_decay["weight_decay"] == 0.0
    assert all(p.ndim == 5 for p in decay["params"])             # conv / conv-transpose weights only
    assert all(p.ndim == 1 for p in no_decay["params"])          # biases and BatchNorm gamma/beta (caught by ndim==1)
    assert len(decay["params"]) + len(no_decay["params"]) == len(list(m.parameters()))
    opt = oracle.build_adam(m.named_parameters(), {"optimizers": {"adam": {"lr": 1e-5, "weight_decay": 5e-4,
                            "betas": [0.9, 0.9999], "eps": 1e-8}}, "param_groups": {"no_decay_keys": ["bias"],
                            "treat_1d_as_no_decay": True}})
    assert opt.defaults["lr"] == 1e-5 and opt.defaults["betas"] == (0.9, 0.9999)


def test_dice_ce_loss_against_manual():
    torch.manual_seed(1)
    z, y = torch.randn(2, 3, 4, 4, 4), (torch.rand(2, 3, 4, 4, 4) > 0.5).float()
    loss = oracle.DiceCELoss(include_background=True, sigmoid=True, lambda_dice=1.0, lambda_ce=1.0)(z, y)
    p = torch.sigmoid(z)
    inter, den = (p * y).sum((2, 3, 4)), p.sum((2, 3, 4)) + y.sum((2, 3, 4))
    dice = (1 - (2 * inter + 1e-5) / (den + 1e-5)).mean()
    ce = -(y * F.log_softmax(z, 1)).sum(1).mean()       # R != 1: soft-label softmax CE even with sigmoid=True
    assert torch.allclose(loss, dice + ce, atol=1e-6)
    z1, y1 = z[:, :1], y[:, :1]
    l1 = oracle.DiceCELoss(include_background=False, sigmoid=True, lambda_dice=5.0, lambda_ce=1.0,
                           weight=torch.tensor([50.0]))(z1, y1)
    p1 = torch.sigmoid(z1)
    d1 = (1 - (2 * (p1 * y1).sum((2, 3, 4)) + 1e-5) / (p1.sum((2, 3, 4)) + y1.sum((2, 3, 4)) + 1e-5)).mean()
    b1 = F.binary_cross_entropy_with_logits(z1, y1, pos_weight=torch.tensor([50.0]))
    assert torch.allclose(l1, 5 * d1 + b1, atol=1e-5)


def test_unet_structure_details():
    cfg = dict(in_channels=4, num_classes=3, channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2], num_res_units=2,
               norm="INSTANCE", act="RELU", dropout=0.0)
    m = oracle.UNet(cfg)
    top_up = m.model[2]
    assert isinstance(top_up[0].conv, torch.nn.ConvTranspose3d) and hasattr(top_up[0], "adn")  # top convT keeps its ADN
    assert not hasattr(top_up[1].conv.unit0, "adn")                                            # last RU is conv-only
    assert isinstance(top_up[1].residual, torch.nn.Identity)
    assert m.model[0].residual.kernel_size == (3, 3, 3) and m.model[0].residual.stride == (2, 2, 2)
    with pytest.raises(ValueError, match="auto"):
        oracle.UNet(dict(cfg, in_channels="auto"))
    assert oracle.UNet(dict(cfg, in_channels="auto"), in_channels=2).in_channels == 2
    assert sum(p.numel() for p in torch.nn.InstanceNorm3d(8).parameters()) == 0


def test_surface_oracle_known_answers():
    """Hand-computable cases for oracle/surface.py (MONAI semantics restated; reference seg_eval.py:312-360)."""
    import math
    import numpy as np
    import oracle
    a = np.zeros((8, 9, 10), bool)
    b = np.zeros((8, 9, 10), bool)
    a[1, 2, 3] = True
    b[4, 6, 3] = True
    hd, asd = oracle.hd_asd(a, b)                                  # single voxels: 3-4-5 triangle
    assert hd == 5.0 and asd == 5.0
    hd, asd = oracle.hd_asd(a, b, spacing=(2.0, 1.0, 1.0))
    assert hd == float(np.float32(math.sqrt(36 + 16)))
    cube = np.zeros((12, 12, 12), bool)
    cube[3:9, 3:9, 3:9] = True
    e = oracle.mask_edges(cube)
    assert e.sum() == 6 ** 3 - 4 ** 3 and not e[5, 5, 5] and e[3, 5, 5]      # the shell of the cube
    shifted = np.roll(cube, 2, axis=2)
    hd, asd = oracle.hd_asd(cube, shifted, percentile=100.0)
    assert hd == 2.0 and 0.0 < asd < 2.0
    border = np.zeros((6, 6, 6), bool)
    border[0:3] = True                                              # touches the volume border: still an edge there
    assert oracle.mask_edges(border)[0].all() and oracle.mask_edges(border)[2].all() and oracle.mask_edges(border)[1, 0, 0]
    assert not oracle.mask_edges(border)[1, 2, 2]
    empty = np.zeros_like(cube)
    assert math.isnan(oracle.hd_asd(empty, empty)[0]) and math.isnan(oracle.hd_asd(empty, empty)[1])
    hd, asd = oracle.hd_asd(cube, empty)
    assert math.isnan(hd) and math.isinf(asd)
    # evaluator fix-ups: empty prediction with GT -> diagonal; empty GT -> left alone (not accumulated by the caller)
    pred = torch.zeros(1, 2, 12, 12, 12, dtype=torch.uint8)
    gt = torch.zeros(1, 2, 12, 12, 12, dtype=torch.uint8)
    gt[0, 0] = torch.from_numpy(cube)
    pred[0, 1] = torch.from_numpy(cube)
    hd, asd = oracle.evaluator_surface(pred, gt, (1.0, 1.0, 1.0))
    assert abs(float(hd[0, 0]) - math.sqrt(3 * 121)) < 1e-5 and abs(float(asd[0, 0]) - math.sqrt(3 * 121)) < 1e-5
    assert not math.isfinite(float(hd[0, 1]))


def test_optimizer_factory_follows_the_reference_rules():
    """reference src/core/experiment_manager.py:199-237: class by `training.optimizer`, kwargs from
    `training.optimizers.<name>`, decay / no-decay groups, lr / momentum fall-backs, unknown name -> ValueError."""
    import pytest
    m = torch.nn.Sequential(torch.nn.Conv3d(2, 3, 1), torch.nn.BatchNorm3d(3))
    tr = {"optimizer": "sgd", "learning_rate": 0.5, "momentum": 0.7,
          "optimizers": {"sgd": {"weight_decay": 1e-4, "nesterov": False},
                         "adamw": {"lr": 5e-4, "weight_decay": 0.05, "betas": [0.9, 0.95], "eps": 1e-6}},
          "param_groups": {"no_decay_keys": ["bias", "bn", "norm", "LayerNorm"], "treat_1d_as_no_decay": True}}
    opt = oracle.build_optimizer(m.named_parameters(), tr)
    assert isinstance(opt, torch.optim.SGD)
    g_decay, g_nodecay = opt.param_groups
    assert g_decay["lr"] == 0.5 and g_decay["momentum"] == 0.7 and g_decay["weight_decay"] == 1e-4
    assert g_nodecay["weight_decay"] == 0.0 and len(g_decay["params"]) == 1 and len(g_nodecay["params"]) == 3
    opt = oracle.build_optimizer(m.named_parameters(), dict(tr, optimizer="adamw"))
    assert isinstance(opt, torch.optim.AdamW) and opt.param_groups[0]["betas"] == (0.9, 0.95)
    assert opt.param_groups[0]["weight_decay"] == 0.05 and opt.param_groups[0]["eps"] == 1e-6
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        oracle.build_optimizer(m.named_parameters(), dict(tr, optimizer="lion"))
