"""MONAI block semantics restated with torch.nn (oracle; test infrastructure; parity unpinned).

Third-party origin: ``monai.networks.blocks.{Convolution, ADN, ResidualUnit, UpSample}`` and
``monai.networks.layers.SkipConnection`` (PyPI ``monai``, unpinned by the reference, >= 1.3
inferred).  Reference call sites that fix the arguments:
src/models/unet_multimodal_midfusion.py:45-55 (ResidualUnit in the encoders), :84-92
(Convolution in the fusion layer), :114-120 (UpSample "nontrainable"), :121-131 (ResidualUnit
in the decoder) and src/models/unet.py:56-66 (monai UNet).  Child-module names are the ones
MONAI uses, so ``state_dict()`` keys equal those of a checkpoint written by the reference's
CheckpointHook (src/core/hooks.py:55-62).
"""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

NormSpec = Union[str, Tuple[str, dict], Sequence[Any], None]


def parse_norm(norm: NormSpec) -> Tuple[Optional[str], dict]:
    """('GROUP', {'num_groups': 4}) | 'INSTANCE' | 'batch' | None -> (UPPER name, kwargs)."""
    if norm is None:
        return None, {}
    if isinstance(norm, str):
        return norm.upper(), {}
    name, kwargs = norm[0], (norm[1] if len(norm) > 1 else {})
    return str(name).upper(), dict(kwargs)


def make_norm(norm: NormSpec, channels: int) -> Optional[nn.Module]:
    name, kw = parse_norm(norm)
    if name is None:
        return None
    if name == "INSTANCE":
        return nn.InstanceNorm3d(channels, **kw)  # torch defaults: affine=False, no running stats
    if name == "BATCH":
        return nn.BatchNorm3d(channels, **kw)
    if name == "GROUP":
        return nn.GroupNorm(num_channels=channels, **kw)
    raise ValueError(f"unsupported norm {norm!r}")


def make_act(act: Optional[str]) -> Optional[nn.Module]:
    if act is None:
        return None
    name = act.upper() if isinstance(act, str) else str(act[0]).upper()
    if name == "RELU":
        return nn.ReLU()
    if name == "PRELU":
        return nn.PReLU()
    if name == "LEAKYRELU":
        return nn.LeakyReLU()
    raise ValueError(f"unsupported act {act!r}")


class ADN(nn.Sequential):
    """Norm -> Dropout -> Act in the order given (MONAI default for these blocks is "NDA")."""

    def __init__(self, channels: int, ordering: str = "NDA", act=None, norm=None, dropout=None):
        super().__init__()
        for item in ordering.upper():
            if item == "N" and norm is not None:
                self.add_module("N", make_norm(norm, channels))
            elif item == "D" and dropout is not None:
                self.add_module("D", nn.Dropout(float(dropout)))
            elif item == "A" and act is not None:
                self.add_module("A", make_act(act))


class Convolution(nn.Sequential):
    """conv (or transposed conv) followed by ADN; "same" padding (k-1)//2."""

    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        strides: int = 1,
        kernel_size: int = 3,
        adn_ordering: str = "NDA",
        act="PRELU",
        norm="INSTANCE",
        dropout=None,
        bias: bool = True,
        conv_only: bool = False,
        is_transposed: bool = False,
    ):
        super().__init__()
        assert spatial_dims == 3, "the adaptation path is 3-D only"
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = nn.ConvTranspose3d(
                in_channels, out_channels, kernel_size, stride=strides, padding=pad,
                output_padding=strides - 1, bias=bias,
            )
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=strides, padding=pad, bias=bias)
        self.add_module("conv", conv)
        if conv_only:
            return
        if act is None and norm is None and dropout is None:
            return
        self.add_module("adn", ADN(out_channels, adn_ordering, act, norm, dropout))


class ResidualUnit(nn.Module):
    """``conv(x) + residual(x)``; no norm/act after the add."""

    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        strides: int = 1,
        kernel_size: int = 3,
        subunits: int = 2,
        adn_ordering: str = "NDA",
        act="PRELU",
        norm="INSTANCE",
        dropout=None,
        bias: bool = True,
        last_conv_only: bool = False,
    ):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual: nn.Module = nn.Identity()
        subunits = max(1, int(subunits))
        sch, sst = in_channels, strides
        for su in range(subunits):
            unit = Convolution(
                spatial_dims, sch, out_channels, strides=sst, kernel_size=kernel_size,
                adn_ordering=adn_ordering, act=act, norm=norm, dropout=dropout, bias=bias,
                conv_only=last_conv_only and su == subunits - 1,
            )
            self.conv.add_module(f"unit{su:d}", unit)
            sch, sst = out_channels, 1
        if strides != 1 or in_channels != out_channels:
            rk, rp = kernel_size, (kernel_size - 1) // 2
            if strides == 1:  # channel adaptation only: 1x1x1, no padding
                rk, rp = 1, 0
            self.residual = nn.Conv3d(in_channels, out_channels, rk, strides, rp, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        res = self.residual(x)
        cx = self.conv(x)
        return cx + res


class UpSample(nn.Sequential):
    """mode="nontrainable": optional 1x1x1 ``preconv`` then trilinear x``scale``, align_corners=True."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, scale_factor: int = 2,
                 mode: str = "nontrainable", bias: bool = True):
        super().__init__()
        assert spatial_dims == 3 and mode == "nontrainable"
        if out_channels != in_channels:
            self.add_module("preconv", nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=bias))
        self.add_module(
            "upsample_non_trainable",
            nn.Upsample(scale_factor=(float(scale_factor),) * 3, mode="trilinear", align_corners=True),
        )


class SkipConnection(nn.Module):
    """cat([x, submodule(x)], dim=1) - the skip comes first."""

    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.submodule = submodule

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.cat([x, self.submodule(x)], dim=1)
