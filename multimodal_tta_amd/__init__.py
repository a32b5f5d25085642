"""multimodal_tta_amd: the per-test-volume adaptation hot path of zhm1205/Multimodal_TTA on MI355X.

Importing the package registers, under the reference's own registry names
(reference src/registry.py:60-124):
  models                 unet, unet_multimodal_deepfusion, unet_multimodal_midfusion
  evaluation strategies  seg_eval, seg_tta_eval
  plugins                entmin_tta, seg_supervised_step
  dataset builders       brats, hecktor21, default (synthetic volumes)
All arithmetic runs in csrc/libmmtta.so (HIP, gfx950); there is no CPU or PyTorch fallback.
"""
from . import registry  # noqa: F401
from .config import Cfg, compose, get_config, require_config  # noqa: F401
from . import models  # noqa: F401  (registers the models)
from . import evaluation  # noqa: F401
from . import tta  # noqa: F401
from . import trainer  # noqa: F401  (plugin seg_supervised_step)
from . import datasets  # noqa: F401

__all__ = ["registry", "compose", "get_config", "require_config", "Cfg", "models", "evaluation", "tta", "datasets"]
__version__ = "0.1.0"
