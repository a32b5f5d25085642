"""Host-side logic of the product that needs no GPU: parameter selection, modality masks, the
decay / no-decay / frozen grouping and arena layout, evaluator argument checks, synthetic data."""
import pytest
import torch

import oracle
from multimodal_tta_amd.engine import GROUP_DECAY, GROUP_FROZEN, GROUP_NO_DECAY, Arena, Runtime
from multimodal_tta_amd.evaluation import SegmentationEvaluationStrategy, as_list_str, dice_iou_from_counts
from multimodal_tta_amd.models import MultimodalUNetDeepFusion, UNet
from multimodal_tta_amd.ops import MmttaError
from multimodal_tta_amd.synth import synth_volume
from multimodal_tta_amd.tta import drop_modality, modality_mask, select_params

SMALL = dict(in_channels=4, num_classes=3, channels=[4, 8, 16, 32, 64], strides=[2, 2, 2, 2], num_res_units=2,
             norm="BATCH", act="RELU")


def test_select_params_matches_oracle():
    hip, ref = UNet(SMALL), oracle.UNet(SMALL)
    for spec in ("all", "norm_affine", ["model.2."], "residual"):
        assert select_params(hip, spec) == [n for n, _ in oracle.select_params(ref, spec)]
    assert all(".adn.N." in n for n in select_params(hip, "norm_affine")) and select_params(hip, "norm_affine")
    inst = UNet(dict(SMALL, norm="INSTANCE"))
    assert select_params(inst, "norm_affine") == []          # nothing to adapt under INSTANCE norm


def test_modality_mask_sequence_matches_oracle():
    from oracle.tta import modality_mask as ref_mask
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    for _ in range(20):
        assert modality_mask(4, [1], 0.6, g1) == ref_mask(4, [1], 0.6, g2)
    assert modality_mask(4, [1], 0.0, None) == [True, False, True, True]
    g = torch.Generator().manual_seed(0)
    for _ in range(50):
        m = modality_mask(2, [], 0.99, g)
        assert any(m)                                            # one modality always survives
    x = torch.ones(1, 4, 2, 2, 2)
    y = drop_modality(x, [True, False, True, True])
    assert y[0, 1].abs().sum() == 0 and y[0, 0].sum() == 8 and drop_modality(x, [True] * 4) is x


def test_arena_groups_and_layout_on_cpu():
    """The arena is plain torch: its layout logic runs on the CPU device."""
    model = UNet(SMALL)
    rt = Runtime(torch.device("cpu"))
    for n, p in model.named_parameters():
        rt.make_ref(n, p)
    trainable = {n for n in select_params(model, ["model.2.", "adn.N"])}
    rt.assign_groups(trainable, ["bias", "bn", "norm", "LayerNorm"], True)
    groups = {r.name: r.group for r in rt.refs}
    assert groups["model.2.0.conv.weight"] == GROUP_DECAY and groups["model.2.0.conv.bias"] == GROUP_NO_DECAY
    assert groups["model.0.conv.unit0.adn.N.weight"] == GROUP_NO_DECAY       # 1-D: caught by treat_1d
    assert groups["model.0.conv.unit0.conv.weight"] == GROUP_FROZEN
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    ar = rt.build_arena()
    assert ar.owns() and 0 < ar.n_decay < ar.n_train < ar.total
    for r in rt.refs:
        assert r.offset % 4 == 0 and torch.equal(r.param.detach(), before[r.name])
        if r.group == GROUP_DECAY:
            assert r.offset + r.numel <= ar.n_decay
        elif r.group == GROUP_NO_DECAY:
            assert ar.n_decay <= r.offset and r.offset + r.numel <= ar.n_train
        else:
            assert r.offset >= ar.n_train and not r.param.requires_grad
    ar.snapshot_source()
    ar.params.add_(1.0)
    ar.step.fill_(3)
    ar.restore_source()
    assert all(torch.equal(r.param.detach(), before[r.name]) for r in rt.refs) and int(ar.step) == 0
    model.load_state_dict({k: v + 2 for k, v in model.state_dict().items() if v.dtype.is_floating_point}, strict=False)
    assert ar.owns()                                              # load_state_dict copies in place
    model.double()
    assert not ar.owns()                                          # a dtype/device move re-homes the parameters


def test_models_refuse_cpu_inputs_and_unsupported_configs():
    m = UNet(dict(SMALL, norm="INSTANCE"))
    with pytest.raises(MmttaError, match="no CPU"):
        m(torch.zeros(1, 4, 16, 16, 16))
    with pytest.raises(NotImplementedError):
        UNet(dict(SMALL, act="PRELU"))
    with pytest.raises(ValueError, match="auto"):
        UNet(dict(SMALL, in_channels="auto"))
    d = MultimodalUNetDeepFusion(dict(num_modalities=2, num_classes=1, channels=[2, 4, 8, 16, 32]))
    assert d.get_domain_loss_weight() == 0.1 and d.num_modalities == 2
    with pytest.raises(MmttaError, match="MI355X"):               # the auxiliary outputs are computed on the GPU as well
        d(torch.zeros(1, 2, 16, 16, 16), return_domain_logits=True)


def test_state_dict_roundtrip_with_oracle_keys():
    for hip_cls, ref_cls, cfg in ((UNet, oracle.UNet, SMALL),
                                  (MultimodalUNetDeepFusion, oracle.MultimodalUNetDeepFusion,
                                   dict(num_modalities=4, num_classes=3, channels=[2, 4, 8, 16, 32]))):
        ref, hip = ref_cls(cfg), hip_cls(cfg)
        assert list(ref.state_dict().keys()) == list(hip.state_dict().keys())
        hip.load_state_dict(ref.state_dict())
        wrapped = {"module." + k: v for k, v in ref.state_dict().items()}      # DataParallel prefix, hooks.py:57
        hip.load_state_dict({k[len("module."):]: v for k, v in wrapped.items()})
        for k, v in hip.state_dict().items():
            assert torch.equal(v, ref.state_dict()[k])


def test_evaluator_config_and_batch_checks():
    cfg = {"evaluation": {"seg": {"threshold": 0.3, "region_order": ["gtvt"], "spacing": [1, 1, 3]},
                          "loss": {"report_loss": True}},
           "training": {"criterion": {"include_background": False, "lambda_dice": 5.0, "weight": [50.0]}}}
    s = SegmentationEvaluationStrategy(cfg)
    assert s.threshold == 0.3 and s.region_order == ["gtvt"] and s.spacing == (1.0, 1.0, 3.0) and s.report_loss
    assert s.loss_fn.lambda_dice == 5.0 and s.loss_fn.weight == [50.0] and not s.loss_fn.include_background
    with pytest.raises(ValueError, match="spacing"):
        SegmentationEvaluationStrategy({"evaluation": {"seg": {"spacing": [1, 1]}}})
    assert not s.enable_surface and not s.asd_symmetric                   # off by default (reference seg_eval.py:193-196)
    on = SegmentationEvaluationStrategy({"evaluation": {"surface": {"enable": True, "asd_symmetric": True}}})
    assert on.enable_surface and on.asd_symmetric
    with pytest.raises(KeyError, match="label"):
        s.check_batch({"image": torch.zeros(1, 2, 4, 4, 4)}, "cpu")
    with pytest.raises(ValueError, match="channels=3"):
        s.check_batch({"image": torch.zeros(1, 2, 4, 4, 4), "label": torch.zeros(1, 3, 4, 4, 4)}, "cpu")
    x, y = s.check_batch({"image": torch.zeros(2, 2, 4, 4, 4), "label": torch.ones(1, 4, 4, 4)}, "cpu")
    assert y.shape == (2, 1, 4, 4, 4) and y.dtype == torch.float32        # [R,D,H,W] is broadcast over the batch
    assert as_list_str(None, 2) == ["", ""] and as_list_str("d", 2) == ["d", "d"]
    assert as_list_str(torch.tensor([3, 4]), 2) == ["3", "4"] and as_list_str(("a", "b"), 2) == ["a", "b"]


def test_dice_from_counts_equals_reference_formula():
    counts = torch.tensor([[[2, 4, 4], [0, 0, 3], [0, 5, 0]]], dtype=torch.int64)
    d, i, v = dice_iou_from_counts(counts)
    pred = torch.zeros(1, 3, 8, dtype=torch.uint8)
    gt = torch.zeros(1, 3, 8, dtype=torch.uint8)
    pred[0, 0, :4] = 1; gt[0, 0, 2:6] = 1
    gt[0, 1, :3] = 1
    pred[0, 2, :5] = 1
    dr, ir, vr = oracle.binary_dice_iou(pred.view(1, 3, 2, 2, 2), gt.view(1, 3, 2, 2, 2))
    assert torch.equal(d, dr) and torch.equal(i, ir) and torch.equal(v, vr)


def test_synthetic_volume_contract():
    a, b = synth_volume(3, 4, (16, 20, 24), 3), synth_volume(3, 4, (16, 20, 24), 3)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["label"], b["label"])     # seeded
    assert a["image"].shape == (4, 16, 20, 24) and a["image"].dtype == torch.float32
    assert a["label"].shape == (3, 16, 20, 24) and set(a["label"].unique().tolist()) <= {0.0, 1.0}
    et, tc, wt = a["label"]
    assert (et <= tc).all() and (tc <= wt).all() and wt.sum() > et.sum() > 0               # ET in TC in WT
    assert a["image"][:, 0, 0, 0].abs().sum() == 0                                          # background exactly 0
    assert not torch.equal(a["image"], synth_volume(4, 4, (16, 20, 24), 3)["image"])
    h = synth_volume(0, 2, (12, 36, 36), 1)
    assert h["label"].shape == (1, 12, 36, 36) and isinstance(h["domain"], str) and h["index"] == 0


def test_checkpoint_interchange_with_the_reference_format(tmp_path):
    """A CheckpointHook-format file written from the oracle model (MONAI key names, optionally with the
    DataParallel ``module.`` prefix, reference src/core/hooks.py:53-93) loads into the product containers, and a
    file saved by the product reads back into the oracle model."""
    import torch
    import oracle
    from multimodal_tta_amd.checkpoint import load_source_weights, read_state_dict, save_checkpoint
    from multimodal_tta_amd.models import UNet

    cfg = dict(name="unet", in_channels=2, num_classes=1, spatial_dims=3, channels=[4, 8, 16], strides=[2, 2],
               num_res_units=2, norm="BATCH", act="RELU", dropout=0.0)
    torch.manual_seed(0)
    ref = oracle.UNet(cfg)
    path = tmp_path / "checkpoints" / "best_model.pth"
    path.parent.mkdir()
    torch.save({"epoch": 7, "model_state_dict": {"module." + k: v for k, v in ref.state_dict().items()},
                "optimizer_state_dict": {"state": {}, "param_groups": []}, "best_metrics": {"avg_dc": 0.5}}, path)
    hip = UNet(cfg)
    meta = load_source_weights(hip, str(path))
    assert meta == {"epoch": 7, "best_metrics": {"avg_dc": 0.5}}
    for (ka, va), (kb, vb) in zip(ref.state_dict().items(), hip.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    out = tmp_path / "out" / "checkpoint_epoch_0.pth"
    save_checkpoint(str(out), hip, epoch=0, best_metrics={"avg_dc": 0.25})
    ref2 = oracle.UNet(cfg)
    ref2.load_state_dict(read_state_dict(str(out)))
    assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), ref2.state_dict().values()))
    ck = torch.load(str(out), weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "best_metrics"}


HECKTOR_POLICY = {"enabled": True, "channel_names": ["ct", "pt"],
                  "channels": {"ct": {"clip": [-1000, 1000], "zscore": {"masked": True, "mask_gt": -900, "eps": 1.0e-6}},
                               "pt": {"clip": [0.0, 15.0], "zscore": {"masked": True, "mask_gt": 0.0, "eps": 1.0e-6}}}}


def test_intensity_rules_follow_the_reference_policy_keys():
    """The rule table handed to libmmtta mirrors reference configs/_global_patches/hecktor21.yaml:26-46 and the
    defaults of reference src/datasets/transforms.py:178-183 (masked True, mask_gt -inf, eps 1e-6, min_count 16)."""
    import pytest
    from multimodal_tta_amd.transforms import build_rules
    r = build_rules(2, HECKTOR_POLICY, mean=[0, 0], std=[1, 1])
    assert [x.clip for x in r] == [1, 1] and (r[0].lo, r[0].hi) == (-1000.0, 1000.0) and (r[1].lo, r[1].hi) == (0.0, 15.0)
    assert all(x.zscore == 1 and x.masked == 1 and x.min_count == 16 and x.legacy == 0 for x in r)
    assert r[0].mask_gt == -900.0 and r[1].mask_gt == 0.0 and abs(r[0].eps - 1e-6) < 1e-12
    r = build_rules(3, {"enabled": True, "channels": {"1": {"zscore": {"masked": False}}}})
    assert [x.zscore for x in r] == [0, 1, 0] and r[1].masked == 0 and r[1].mask_gt == float("-inf")
    r = build_rules(4, {"enabled": False}, mean=[0.5], std=[2.0])
    assert all(x.legacy == 1 and x.mean == 0.5 and x.std == 2.0 for x in r)
    with pytest.raises(RuntimeError, match="channel_names"):
        build_rules(3, HECKTOR_POLICY)
    with pytest.raises(RuntimeError, match="mean/std"):
        build_rules(3, None, mean=[0.0, 1.0])


def test_oracle_normalize_image_semantics():
    """Known answers of the restated reference transform (reference src/datasets/transforms.py:163-223)."""
    import torch
    import oracle
    x = torch.tensor([[[[-2000.0, -950.0, 0.0, 100.0, 3000.0]]], [[[0.0, 0.0, 1.0, 3.0, 20.0]]]])   # [2,1,1,5]
    pol = {"enabled": True, "channel_names": ["ct", "pt"],
           "channels": {"ct": {"clip": [-1000, 1000], "zscore": {"masked": True, "mask_gt": -900, "min_count": 2}},
                        "pt": {"clip": [0.0, 15.0], "zscore": {"masked": True, "mask_gt": 0.0, "min_count": 4}}}}
    y = oracle.normalize_image(x, intensity_policy=pol)
    ct = torch.tensor([-1000.0, -950.0, 0.0, 100.0, 1000.0])
    vals = ct[ct > -900]                                     # 0, 100, 1000
    assert torch.allclose(y[0].flatten(), (ct - vals.mean()) / vals.std(unbiased=False))
    pt = torch.tensor([0.0, 0.0, 1.0, 3.0, 15.0])            # only 3 voxels > 0 < min_count 4: statistics over all five
    assert torch.allclose(y[1].flatten(), (pt - pt.mean()) / pt.std(unbiased=False))
    z = oracle.normalize_image(x, intensity_policy={"enabled": False}, mean=[1.0, 2.0], std=[2.0, 4.0])
    assert torch.allclose(z[1], (x[1] - 2.0) / 4.0) and torch.equal(oracle.normalize_image(x, normalize=False), x)


def test_surface_columns_travel_through_the_gather_table():
    """With evaluation.surface.enable the per-volume table grows by hd95[R] + asd[R] and the replayed aggregation
    emits the reference's key set (src/evaluation/seg_eval.py:424-440,459-476)."""
    from multimodal_tta_amd.evaluation import RegionAccumulator, metrics_from_table, table_width
    regions = ["ET", "TC", "WT"]
    R = 3
    assert table_width(R) == 12 and table_width(R, True) == 18
    rows = torch.tensor([[0, 0, 0.5, .9, .8, .7, .8, .7, .6, 1, 1, 0, 2.0, 3.0, 9.0, 1.0, 1.5, 9.0],
                         [1, 1, 0.7, .5, .4, .3, .4, .3, .2, 1, 0, 1, 4.0, 9.0, 6.0, 2.0, 9.0, 2.5]], dtype=torch.float64)
    m = metrics_from_table(rows, regions, ["a", "b"], True, surface=True)
    assert m["et_hd95"] == 3.0 and m["tc_hd95"] == 3.0 and m["wt_hd95"] == 6.0      # invalid entries (9.0) skipped
    assert m["avg_hd95"] == 4.0 and m["et_asd"] == 1.5 and m["avg_asd"] == (1.5 + 1.5 + 2.5) / 3
    assert m["dom/a/et_hd95"] == 2.0 and m["dom/b/wt_asd"] == 2.5 and m["dom/a/wt_hd95"] == 0.0
    keys = list(m)
    assert keys.index("loss") < keys.index("et_hd95") < keys.index("avg_hd95") < keys.index("et_asd") < keys.index("avg_asd")
    acc = RegionAccumulator(regions, surface=False)
    acc.add_row([.9, .8, .7], [.8, .7, .6], [True, True, False], "a")
    assert not any("hd95" in k or "asd" in k for k in acc.metrics(False))


def test_hardware_queue_dependency_is_loud(monkeypatch):
    """more than two lanes only overlap when the HIP runtime was started with GPU_MAX_HW_QUEUES >= 8; a host
    that initialised the GPU before importing the package gets a warning (once) with the measured cost, and
    `lanes_effective` says what to expect."""
    import warnings

    import multimodal_tta_amd as pkg
    from multimodal_tta_amd import ops

    class _FakeStream:
        def __init__(self, device=None):
            pass

    monkeypatch.setattr(torch.cuda, "Stream", _FakeStream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: __import__("contextlib").nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: None)
    monkeypatch.setattr(pkg, "HIP_STARTED_BEFORE_IMPORT", True)
    monkeypatch.setattr(pkg, "HW_QUEUES_AT_HIP_START", None)
    monkeypatch.setattr(ops, "_QUEUE_WARNED", False)
    assert ops.hw_queue_status()["queues"] == 4 and ops.lanes_effective(4) == 2 and ops.lanes_effective(2) == 2
    with pytest.warns(RuntimeWarning, match="GPU_MAX_HW_QUEUES"):
        ops.lane_streams(4, "cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ops.lane_streams(4, "cpu")                       # once per process
        monkeypatch.setattr(ops, "_QUEUE_WARNED", False)
        ops.lane_streams(2, "cpu")                       # two lanes fit the default queues
        monkeypatch.setattr(pkg, "HW_QUEUES_AT_HIP_START", "8")
        ops.lane_streams(4, "cpu")
    assert ops.lanes_effective(4) == 4


def test_workspace_selectors_are_per_thread():
    """`ops.Workspace.lane / slot / frozen` pick the scratch buffer of the launches that follow: a second Python thread driving
    its own lane starts from the defaults and does not disturb the first one's selection ("smaller")."""
    import threading

    from multimodal_tta_amd import ops
    ops.Workspace.lane, ops.Workspace.slot = 3, 1
    seen = {}

    def other():
        seen["start"] = (ops.Workspace.lane, ops.Workspace.slot, ops.Workspace.frozen)
        ops.Workspace.lane, ops.Workspace.frozen = 7, True
        seen["end"] = (ops.Workspace.lane, ops.Workspace.frozen)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    try:
        assert seen == {"start": (0, 0, False), "end": (7, True)}
        assert (ops.Workspace.lane, ops.Workspace.slot, ops.Workspace.frozen) == (3, 1, False)
    finally:
        ops.Workspace.lane, ops.Workspace.slot = 0, 0
