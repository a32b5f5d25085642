"""The registry keeps the reference's observable behaviour (captured from the importable reference
module src.registry into golden/registry_behaviour.json by golden/make_golden.py)."""
import contextlib
import io
import json
import os

import pytest

from multimodal_tta_amd import registry as reg

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "registry_behaviour.json")))


def test_tables_and_helper_names_match_reference():
    for const, name in GOLD["tables"]:
        assert getattr(reg, const).name == name
    for helper in GOLD["helpers"]:
        assert callable(getattr(reg, helper)), helper
    assert list(reg.list_all_components().keys()) == GOLD["list_all_components_keys"]


def test_register_get_duplicate_clear_behave_like_reference():
    r = reg.Registry("demo")

    @r.register("a")
    class A:
        pass

    assert r.register("b", A) is A
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r.register("a", int)
    assert buf.getvalue() == GOLD["duplicate_warning"]
    assert r.get("a").__name__ == GOLD["after_duplicate_get_a"]
    assert r.list_all() == GOLD["list_all"]
    assert [r.has("a"), r.has("zzz")] == GOLD["has"]
    with pytest.raises(KeyError) as ei:
        r.get("zzz")
    assert list(ei.value.args) == GOLD["keyerror_args"]
    r.clear()
    assert r.list_all() == GOLD["after_clear"]


def test_package_registers_the_hot_path_components():
    import multimodal_tta_amd  # noqa: F401
    assert {"unet", "unet_multimodal_deepfusion", "unet_multimodal_midfusion"} <= set(reg.list_models())
    assert {"seg_eval", "seg_tta_eval"} <= set(reg.list_evaluation_strategies())
    assert "entmin_tta" in reg.list_plugins()
    assert {"brats", "hecktor21", "default"} <= set(reg.list_dataset_builders())
    with pytest.raises(KeyError, match="nope is not registered in models"):
        reg.get_model("nope")
