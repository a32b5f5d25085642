"""Oracle restatement of ``monai.networks.nets.UNet`` and of the reference wrapper around it.

Test infrastructure; parity unpinned (MONAI absent, see oracle/__init__.py).
Follows: reference src/models/unet.py:14-69 (config keys, defaults: in_channels 3 or ctor
argument, num_classes 1, channels [32,64,128,256,512], strides [2,2,2,2], num_res_units 0,
act "relu", norm "BATCH", dropout 0.0, spatial_dims 3; ``in_channels: auto`` raises) and
MONAI's recursive block construction.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import torch
import torch.nn as nn

from .blocks import Convolution, ResidualUnit, SkipConnection


def cfg_get(cfg: Any, key: str, default: Any = None) -> Any:
    """Dotted read with the reference's None => default rule (src/utils/config.py:21-32)."""
    node = cfg
    for part in key.split("."):
        if not isinstance(node, dict) or part not in node:
            return default
        node = node[part]
    return default if node is None else node


class MonaiUNet(nn.Module):
    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        channels: Sequence[int],
        strides: Sequence[int],
        kernel_size: int = 3,
        up_kernel_size: int = 3,
        num_res_units: int = 0,
        act="PRELU",
        norm="INSTANCE",
        dropout: Optional[float] = 0.0,
        bias: bool = True,
        adn_ordering: str = "NDA",
    ):
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        if len(strides) < len(channels) - 1:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        self.dimensions = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channels, self.strides = list(channels), list(strides)
        self.kernel_size, self.up_kernel_size = kernel_size, up_kernel_size
        self.num_res_units, self.act, self.norm = num_res_units, act, norm
        self.dropout, self.bias, self.adn_ordering = dropout, bias, adn_ordering

        def create_block(inc: int, outc: int, chans: Sequence[int], strs: Sequence[int], is_top: bool) -> nn.Module:
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub = create_block(c, c, chans[1:], strs[1:], False)
                upc = c * 2
            else:
                sub = self._down(c, chans[1], 1, False)  # bottom layer
                upc = c + chans[1]
            down = self._down(inc, c, s, is_top)
            up = self._up(upc, outc, s, is_top)
            return nn.Sequential(down, SkipConnection(sub), up)

        self.model = create_block(in_channels, out_channels, self.channels, self.strides, True)

    def _down(self, inc: int, outc: int, stride: int, is_top: bool) -> nn.Module:
        if self.num_res_units > 0:
            return ResidualUnit(
                self.dimensions, inc, outc, strides=stride, kernel_size=self.kernel_size,
                subunits=self.num_res_units, act=self.act, norm=self.norm, dropout=self.dropout,
                bias=self.bias, adn_ordering=self.adn_ordering,
            )
        return Convolution(
            self.dimensions, inc, outc, strides=stride, kernel_size=self.kernel_size, act=self.act,
            norm=self.norm, dropout=self.dropout, bias=self.bias, adn_ordering=self.adn_ordering,
        )

    def _up(self, inc: int, outc: int, stride: int, is_top: bool) -> nn.Module:
        conv: nn.Module = Convolution(
            self.dimensions, inc, outc, strides=stride, kernel_size=self.up_kernel_size, act=self.act,
            norm=self.norm, dropout=self.dropout, bias=self.bias,
            conv_only=is_top and self.num_res_units == 0, is_transposed=True,
            adn_ordering=self.adn_ordering,
        )
        if self.num_res_units > 0:
            ru = ResidualUnit(
                self.dimensions, outc, outc, strides=1, kernel_size=self.kernel_size, subunits=1,
                act=self.act, norm=self.norm, dropout=self.dropout, bias=self.bias,
                last_conv_only=is_top, adn_ordering=self.adn_ordering,
            )
            conv = nn.Sequential(conv, ru)
        return conv

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)


class UNet(MonaiUNet):
    """cfg -> MonaiUNet, same key handling as reference src/models/unet.py:22-66."""

    def __init__(self, cfg: Dict[str, Any], in_channels: Optional[int] = None):
        c_in_cfg = cfg_get(cfg, "in_channels", 3)
        c_in = in_channels if in_channels is not None else (None if c_in_cfg == "auto" else int(c_in_cfg))
        if c_in is None:
            raise ValueError("[UNet] in_channels is 'auto'; please pass in_channels at construction time.")
        super().__init__(
            spatial_dims=int(cfg_get(cfg, "spatial_dims", 3)),
            in_channels=c_in,
            out_channels=int(cfg_get(cfg, "num_classes", 1)),
            channels=list(cfg_get(cfg, "channels", [32, 64, 128, 256, 512])),
            strides=list(cfg_get(cfg, "strides", [2, 2, 2, 2])),
            num_res_units=int(cfg_get(cfg, "num_res_units", 0)),
            act=cfg_get(cfg, "act", "relu"),
            norm=cfg_get(cfg, "norm", "BATCH"),
            dropout=float(cfg_get(cfg, "dropout", 0.0)),
        )
