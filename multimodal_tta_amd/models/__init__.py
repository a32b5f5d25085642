"""HIP-backed models, registered under the reference's names (reference src/models/__init__.py:29-30)."""
from .unet import UNet  # noqa: F401
from .deepfusion import MultimodalUNetDeepFusion  # noqa: F401

__all__ = ["UNet", "MultimodalUNetDeepFusion"]
