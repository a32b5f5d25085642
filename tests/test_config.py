"""The PyYAML composer reproduces the Hydra/OmegaConf behaviours the reference's config tree relies on
('Config / flags'), and get_config / require_config keep the reference's semantics
(reference src/utils/config.py:7-32)."""
import os
import textwrap

import pytest

from multimodal_tta_amd.config import Cfg, compose, get_config, parse_value, require_config


def test_default_tree_composes_brats_unet():
    cfg = compose(overrides=["task=brats", "dataset=brats", "model=unet"])
    assert cfg.model.name == "unet" and cfg.model.in_channels == 4 and cfg.model.num_classes == 3
    assert cfg.model.norm == "INSTANCE" and cfg.model.channels == [32, 64, 128, 256, 512]
    assert cfg.task.name == "brats" and cfg.task.seed == 42 and cfg.task.eval_strategy == "seg_eval"
    assert cfg.evaluation.seg.threshold == 0.5 and cfg.evaluation.seg.region_order == ["ET", "TC", "WT"]
    assert cfg.method.name == "entmin_tta" and cfg.method.steps == 10


def test_yaml12_floats_like_omegaconf():
    cfg = compose(overrides=["task=brats"])
    adam = cfg.training.optimizers.adam
    assert isinstance(adam.weight_decay, float) and adam.weight_decay == 5e-4      # written "5e-4"
    assert isinstance(cfg.training.optimizers.sgd.lr, float) and cfg.training.optimizers.sgd.lr == 1e-4
    assert adam.lr == 1e-5 and adam.betas == [0.9, 0.9999] and adam.eps == 1e-8
    assert parse_value("5e-3") == 5e-3 and parse_value("1e-4") == 1e-4 and parse_value("7") == 7
    assert parse_value("[1,2]") == [1, 2] and parse_value("null") is None and parse_value("abc") == "abc"


def test_hecktor_global_patch_and_overrides():
    cfg = compose(overrides=["task=hecktor21", "dataset=hecktor21", "training.optimizers.adam.lr=5e-3",
                             "+extra.key=3", "method.steps=4"])
    assert cfg.model.in_channels == 2 and cfg.model.num_classes == 1
    assert cfg.evaluation.seg.threshold == 0.3 and cfg.evaluation.seg.region_order == ["gtvt"]
    assert cfg.training.criterion.ce_weight == [50.0] and cfg.training.criterion.lambda_dice == 5.0
    assert cfg.training.optimizers.adam.lr == 5e-3 and cfg.extra.key == 3 and cfg.method.steps == 4
    assert cfg.training.data.transforms.intensity_policy.channels.ct.zscore.mask_gt == -900


def test_group_selection_and_moddrop_method():
    cfg = compose(overrides=["model=unet_multimodal_deepfusion", "method=tta_moddrop"])
    assert cfg.model.name == "unet_multimodal_deepfusion" and cfg.model.num_modalities == 4
    assert cfg.method.missing_modalities == [1] and cfg.method.moddrop.enabled is True


def test_defaults_self_global_package_and_interpolation(tmp_path):
    (tmp_path / "grp").mkdir()
    (tmp_path / "_patch").mkdir()
    (tmp_path / "config.yaml").write_text(textwrap.dedent("""
        defaults:
          - grp: a
          - _self_
          - late: x
        top: 1
        grp:
          from_root: yes_root
        ref: ${grp.value}-${top}
        run: ${now:%Y}
    """))
    (tmp_path / "grp" / "_base.yaml").write_text("value: base\nkeep: 7\n")
    (tmp_path / "grp" / "a.yaml").write_text("defaults:\n  - /_patch: p\n  - _base\nvalue: from_a\n")
    (tmp_path / "_patch" / "p.yaml").write_text("# @package _global_\ntop: 99\npatched:\n  deep: 1\n")
    (tmp_path / "late").mkdir()
    (tmp_path / "late" / "x.yaml").write_text("z: 5\n")
    cfg = compose(str(tmp_path))
    assert cfg.grp.value == "from_a" and cfg.grp.keep == 7          # file content overrides its defaults
    assert cfg.patched.deep == 1                                     # '# @package _global_' merged at the root
    assert cfg.top == 1                                              # root _self_ comes after the patch
    assert cfg.grp.from_root == "yes_root" and cfg.late.z == 5
    assert cfg.ref == "from_a-1" and len(cfg.run) == 4 and cfg.run.isdigit()
    with pytest.raises(FileNotFoundError):
        compose(str(tmp_path), overrides=["grp=missing"])


def test_get_and_require_config_semantics():
    cfg = Cfg({"a": {"b": None, "c": 3, "l": [10, 20]}})
    assert get_config(cfg, "a.b", "dflt") == "dflt"          # stored None => default (reference config.py:28-29)
    assert get_config(cfg, "a.zzz", 5) == 5 and get_config(cfg, "a.c") == 3 and get_config(cfg, "a.l.1") == 20
    assert get_config({"plain": {"dict": 1}}, "plain.dict") == 1
    with pytest.raises(TypeError):
        get_config(cfg, "a.c", type_=str)
    with pytest.raises(ValueError, match="Required configuration missing: a.b"):
        require_config(cfg, "a.b")
    with pytest.raises(TypeError):
        require_config(cfg, "a.c", str)
    with pytest.raises(TypeError):
        get_config(42, "x")
    cfg.a.new = {"k": 1}
    assert cfg.a.new.k == 1 and isinstance(cfg.a.new, Cfg)
