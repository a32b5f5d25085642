"""multimodal_tta_amd: the per-test-volume adaptation hot path of zhm1205/Multimodal_TTA on MI355X.

Importing the package registers, under the reference's own registry names
(reference src/registry.py:60-124):
  models                 unet, unet_multimodal_deepfusion, unet_multimodal_midfusion
  evaluation strategies  seg_eval, seg_tta_eval
  plugins                entmin_tta, seg_supervised_step
  dataset builders       brats, hecktor21, default (synthetic volumes)
All arithmetic runs in csrc/libmmtta.so (HIP, gfx950); there is no CPU or PyTorch fallback.
"""
import os as _os

# `method.lanes` volumes are adapted concurrently per GPU, one stream each; their kernels only overlap when the streams
# sit on different hardware queues, and the HIP runtime spreads all streams of a process over GPU_MAX_HW_QUEUES (default
# 4) queues.  8 is the measured optimum for 3-4 lanes (42.0 -> 50.1 volumes/s; 16 is slower).  The variable is read when
# HIP starts up, i.e. this only takes effect when the package is imported before the first CUDA call of the process
# (bench.py and main.py set it first thing).
def _hip_already_started() -> bool:
    import sys as _sys
    t = _sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# what the HIP runtime of THIS process saw (or will see) when it started: the value in the environment before this import if
# HIP was already running (then setting the variable now changes nothing), else the value set here
HIP_STARTED_BEFORE_IMPORT = _hip_already_started()
HW_QUEUES_AT_HIP_START = _os.environ.get("GPU_MAX_HW_QUEUES") if HIP_STARTED_BEFORE_IMPORT else None
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if not HIP_STARTED_BEFORE_IMPORT:
    HW_QUEUES_AT_HIP_START = _os.environ["GPU_MAX_HW_QUEUES"]

from . import registry  # noqa: F401,E402
from .config import Cfg, compose, get_config, require_config  # noqa: F401,E402
from . import models  # noqa: F401,E402  (registers the models)
from . import evaluation  # noqa: F401,E402
from . import tta  # noqa: F401,E402
from . import trainer  # noqa: F401,E402  (plugin seg_supervised_step)
from . import datasets  # noqa: F401,E402

__all__ = ["registry", "compose", "get_config", "require_config", "Cfg", "models", "evaluation", "tta", "datasets"]
__version__ = "0.1.0"
