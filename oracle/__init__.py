"""CPU oracle for the per-volume adaptation hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``multimodal_tta_amd/`` imports it, and the product path raises when the HIP library is missing
instead of falling back to anything in here.

PARITY UNPINNED.  The reference (zhm1205/Multimodal_TTA) holds no tests, golden vectors or
known-answer fixtures for this path, its model arithmetic lives in MONAI,
which the reference neither vendors nor pins (requirements.txt lists no monai; inferred >= 1.3
from the ``weight=`` kwarg at src/core/trainers/seg_trainer.py:78) and which is not installed
here, and the reference modules on the path cannot be imported for lack of
omegaconf/monai/torchvision (ordinary ModuleNotFoundError).  What *is* pinned:

* ``src.registry`` and ``src.utils.metrics`` import fine; their behaviour is captured into
  tests/golden/registry_behaviour.json by tests/golden/make_golden.py.
* Everything else is a restatement of MONAI's published block semantics
  composed from ``torch.nn`` primitives, anchored on the reference's own call sites, which are
  cited function by function.  torch CPU fp32 is the arithmetic oracle.
* The entropy-minimisation step does not exist in the reference at all; its
  definition is this repo's and the oracle is its specification.
"""
from .blocks import Convolution, ResidualUnit, UpSample, SkipConnection  # noqa: F401
from .unet import MonaiUNet, UNet  # noqa: F401
from .deepfusion import MultimodalUNetDeepFusion  # noqa: F401
from .losses import (  # noqa: F401
    bernoulli_entropy_loss,
    categorical_entropy_loss,
    entropy_loss,
    DiceCELoss,
)
from .dice import binary_dice_iou, masks_from_logits, RegionAccumulator  # noqa: F401
from .adam import split_param_groups, build_adam, build_optimizer, adam_reference_step  # noqa: F401
from .tta import adapt_volume, select_params  # noqa: F401

MODELS = {
    "unet": UNet,
    "unet_multimodal_deepfusion": MultimodalUNetDeepFusion,
    "unet_multimodal_midfusion": MultimodalUNetDeepFusion,
}
from .transforms import normalize_image  # noqa: F401,E402
from .surface import hd_asd, evaluator_surface, mask_edges, diag_mm  # noqa: F401,E402
