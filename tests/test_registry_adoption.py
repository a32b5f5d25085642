"""Build-container check of the drop-in boundary: when the reference's own ``src.registry`` is
imported FIRST (what its ``main.py`` does, reference main.py:18-20), this package registers into the reference's LIVE
tables, so ``get_model("unet")`` of an unmodified reference process returns the HIP-backed model.  Skipped where the
reference tree is absent (the GPU box); runs in a child process so the adoption happens at a clean import."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {ref!r})
import src.registry as ref_registry          # the reference's module, imported before the package
sys.path.insert(0, {root!r})
import multimodal_tta_amd
from multimodal_tta_amd import registry as mine
for table in ("MODELS", "DATASETS", "DATASET_BUILDERS", "EVALUATION_STRATEGIES", "CRITERIA", "PROVIDERS", "PLUGINS"):
    assert getattr(mine, table) is getattr(ref_registry, table), table
from multimodal_tta_amd.models import UNet
from multimodal_tta_amd.evaluation import SegmentationEvaluationStrategy
assert ref_registry.get_model("unet") is UNet
assert ref_registry.get_model("unet_multimodal_deepfusion") is ref_registry.get_model("unet_multimodal_midfusion")
assert ref_registry.get_evaluation_strategy("seg_eval") is SegmentationEvaluationStrategy
assert "entmin_tta" in ref_registry.PLUGINS.list_all() and "seg_supervised_step" in ref_registry.PLUGINS.list_all()
assert ref_registry.get_plugin("entmin_tta") is mine.get_plugin("entmin_tta")
try:
    ref_registry.get_model("no_such_model")
except KeyError as e:
    assert "no_such_model is not registered in models" in str(e)
else:
    raise AssertionError("unknown name must raise KeyError")
print("ADOPTED")
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "registry.py")), reason="reference tree not present")
def test_components_register_into_the_references_live_tables():
    out = subprocess.run([sys.executable, "-c", CHILD.format(ref=REF, root=ROOT)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ADOPTED" in out.stdout, out.stdout + out.stderr
