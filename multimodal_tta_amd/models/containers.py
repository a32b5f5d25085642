"""Parameter containers with MONAI's module tree and child names, so that ``state_dict()`` keys
and shapes equal those of a checkpoint written by the reference (src/core/hooks.py:55-62; key
list in SURVEY.md Appendix A.6) and default initialisation equals torch's.

These classes hold parameters ONLY.  They are never called: all arithmetic goes through
libmmtta.so (engine.py).  Calling one raises, so there is no silent PyTorch path.
"""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

NormSpec = Union[str, Tuple[str, dict], Sequence[Any], None]


class Holder(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the adaptation path computes through libmmtta.so only"
        )


def parse_norm(norm: NormSpec):
    if norm is None:
        return None, {}
    if isinstance(norm, str):
        return norm.upper(), {}
    return str(norm[0]).upper(), dict(norm[1]) if len(norm) > 1 else {}


def parse_act(act) -> Optional[str]:
    if act is None:
        return None
    return (act if isinstance(act, str) else str(act[0])).upper()


class ADN(Holder):
    def __init__(self, channels: int, ordering: str, act, norm: NormSpec, dropout):
        super().__init__()
        for item in ordering.upper():
            if item == "N" and norm is not None:
                name, kw = parse_norm(norm)
                if name == "INSTANCE":
                    mod = nn.InstanceNorm3d(channels, **kw)
                elif name == "BATCH":
                    mod = nn.BatchNorm3d(channels, **kw)
                elif name == "GROUP":
                    mod = nn.GroupNorm(num_channels=channels, **kw)
                else:
                    raise ValueError(f"unsupported norm {norm!r}")
                self.add_module("N", mod)
            elif item == "D" and dropout is not None:
                self.add_module("D", nn.Dropout(float(dropout)))
            elif item == "A" and act is not None:
                a = parse_act(act)
                if a != "RELU":
                    raise NotImplementedError(
                        f"activation {act!r}: the gfx950 kernels fuse ReLU only (every shipped config uses RELU: "
                        "reference configs/_global_patches/brats.yaml:18)"
                    )
                self.add_module("A", nn.ReLU())


class Convolution(Holder):
    def __init__(self, in_channels: int, out_channels: int, strides: int = 1, kernel_size: int = 3,
                 adn_ordering: str = "NDA", act="PRELU", norm: NormSpec = "INSTANCE", dropout=None, bias: bool = True,
                 conv_only: bool = False, is_transposed: bool = False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=strides, padding=pad,
                                      output_padding=strides - 1, bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=strides, padding=pad, bias=bias)
        self.add_module("conv", conv)
        if conv_only or (act is None and norm is None and dropout is None):
            return
        self.add_module("adn", ADN(out_channels, adn_ordering, act, norm, dropout))


class UnitSeq(Holder):
    pass


class ResidualUnit(Holder):
    def __init__(self, in_channels: int, out_channels: int, strides: int = 1, kernel_size: int = 3, subunits: int = 2,
                 adn_ordering: str = "NDA", act="PRELU", norm: NormSpec = "INSTANCE", dropout=None, bias: bool = True,
                 last_conv_only: bool = False):
        super().__init__()
        self.conv = UnitSeq()
        self.residual: nn.Module = nn.Identity()
        subunits = max(1, int(subunits))
        sch, sst = in_channels, strides
        for su in range(subunits):
            self.conv.add_module(
                f"unit{su:d}",
                Convolution(sch, out_channels, strides=sst, kernel_size=kernel_size, adn_ordering=adn_ordering, act=act,
                            norm=norm, dropout=dropout, bias=bias, conv_only=last_conv_only and su == subunits - 1),
            )
            sch, sst = out_channels, 1
        if strides != 1 or in_channels != out_channels:
            rk, rp = (kernel_size, (kernel_size - 1) // 2) if strides != 1 else (1, 0)
            self.residual = nn.Conv3d(in_channels, out_channels, rk, strides, rp, bias=bias)


class UpSample(Holder):
    def __init__(self, in_channels: int, out_channels: int, scale_factor: int = 2, bias: bool = True):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("trilinear resample kernel is x2 (every stride of the shipped configs is 2)")
        if out_channels != in_channels:
            self.add_module("preconv", nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=bias))
        self.add_module("upsample_non_trainable",
                        nn.Upsample(scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=True))


class SkipConnection(Holder):
    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.submodule = submodule


class Seq(Holder):
    """nn.Sequential-style numbered children ("0", "1", ...)."""

    def __init__(self, *mods: nn.Module):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i: int) -> nn.Module:
        return self._modules[str(i)]

    def __len__(self) -> int:
        return len(self._modules)
