"""``unet``: MONAI's residual U-Net as the reference configures it, computed by libmmtta.so.

Drop-in for reference src/models/unet.py:14-69 (same registry name, same config keys and
defaults, same ``state_dict`` keys as ``monai.networks.nets.UNet``; block semantics in
SURVEY.md Appendix A.4).  ``forward(x[B,C,D,H,W]) -> logits[B,R,D,H,W]`` like the reference;
with gradients enabled the backward runs through the same engine (one autograd node for the
whole network), so ``loss.backward(); optimizer.step()`` of the reference's
``SegTrainer.run_step`` (src/core/trainers/seg_trainer.py:141-143) works unchanged.
"""
from __future__ import annotations

import os

from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..config import as_cfg, get_config
from ..engine import (ConvolutionBlock, ResidualUnitBlock, Runtime, build_convolution, build_residual_unit)
from ..registry import register_model
from .base import HipSegModel
from .containers import Convolution, ResidualUnit, Seq, SkipConnection


class UNetRuntime(Runtime):
    """Flat execution plan of the recursive ``down / skip(sub) / up`` structure."""

    def __init__(self, model: "UNet", device: torch.device):
        super().__init__(device, model.conv_dtype, group=getattr(model, "group", 1))
        self.channels, self.strides = list(model.channels), list(model.strides)
        self.nru = model.num_res_units
        self.in_channels, self.out_channels = model.in_channels, model.out_channels
        # bf16 precision stores the wide forward activations as bf16 (method.storage: bf16, the default of that
        # precision): every layer kind of the residual INSTANCE-norm U-Net has storage-agnostic kernels
        norm_name = model.norm[0] if isinstance(model.norm, (tuple, list)) else model.norm
        # (the thin first / last layers run on dedicated kernels: 4 -> 32|64 channels in, <= 4 out; narrower toy networks
        # take generic direct kernels that work on fp32-stored tensors only and keep fp32 storage)
        self.act_bf16 = (model.conv_dtype == ops.BF16 and getattr(model, "act_storage", "bf16") == "bf16" and self.nru > 0
                         and str(norm_name).upper() == "INSTANCE" and self.channels[0] in (32, 64)
                         and all(c % 8 == 0 and c >= 32 for c in self.channels)
                         and self.in_channels <= 4 and self.out_channels <= 4)
        self.grad_bf16 = self.act_bf16 and getattr(model, "grad_storage", "bf16") == "bf16"
        # the staged network input as bf16 too (8-byte voxels): its only readers - the first level's thin-K convolutions and
        # their weight gradients - round it to bf16 while staging, so the results are the same bit for bit
        self.input_bf16 = self.act_bf16 and self.channels[0] == 32 and 2 <= self.in_channels <= 4
        L = len(self.strides)
        self.L = L
        self.down: List[Any] = []
        self.upconv: List[ConvolutionBlock] = []
        self.upru: List[Optional[ResidualUnitBlock]] = []
        node, prefix = model.model, "model"
        levels = []
        for i in range(L):
            levels.append((node, prefix))
            node, prefix = node[1].submodule, prefix + ".1.submodule"
        self.bottom = self._down_block(prefix, node)
        for i, (nd, pf) in enumerate(levels):
            self.down.append(self._down_block(pf + ".0", nd[0]))
            up = nd[2]
            if self.nru > 0:
                self.upconv.append(build_convolution(self, pf + ".2.0", up[0]))
                self.upru.append(build_residual_unit(self, pf + ".2.1", up[1]))
            else:
                self.upconv.append(build_convolution(self, pf + ".2", up))
                self.upru.append(None)
        # the network input needs no gradient
        self.input_shape = None

    def input_dtype(self) -> torch.dtype:
        return torch.bfloat16 if self.input_bf16 else torch.float32

    def thin_grad_dtype(self) -> torch.dtype:
        # d(logits) and the thin gradients behind it (the top up-convolution's output gradient, before and after its norm
        # backward): 8-byte voxels.  Their readers - the 3x3x3 matrix-tile input gradient, the thin weight gradients, the
        # thin-K input gradient of the up-convolution - round them to bf16 while staging either way
        # (MMTTA_THIN_GRAD_FP32=1: measurement switch for same-box A/B runs)
        if os.environ.get("MMTTA_THIN_GRAD_FP32", "0") == "1":
            return torch.float32
        return torch.bfloat16 if (self.grad_bf16 and self.out_channels <= 4) else torch.float32

    def _down_block(self, prefix: str, cont: nn.Module):
        if isinstance(cont, ResidualUnit):
            return build_residual_unit(self, prefix, cont)
        return build_convolution(self, prefix, cont)

    # ---- buffers
    def _cat(self, i: int, n, dims, grad: bool = False) -> torch.Tensor:
        c = self.channels
        width = 2 * c[i] if i < self.L - 1 else c[i] + c[i + 1]
        d, h, w = dims[i]
        return self.pool.cl(("dcat" if grad else "cat", i), n, d, h, w, width,
                            dtype=self.grad_dtype(width) if grad else self.act_dtype(width))

    def _level_dims(self, d, h, w):
        dims = []
        for s in self.strides:
            d, h, w = ((d + 1) // 2, (h + 1) // 2, (w + 1) // 2) if s == 2 else (d, h, w)
            dims.append((d, h, w))
        return dims

    def _run_down(self, blk, x, out, x_nl=None):
        if isinstance(blk, ResidualUnitBlock):
            blk.fwd(x, x_nl, out)
        else:  # plain Convolution (num_res_units == 0): materialise norm+ReLU into the concat slice
            y, nl = blk.fwd(x, x_nl)
            ops.combine(y, nl, None, None, out)

    # the per-step modality mask of the missing-modality / modality-dropout runs (BASELINE configs[4]) is applied by the
    # first level's kernels WHILE THEY STAGE the input - a per-(item, channel) scale of 1 or 0 through the norm-on-load hook,
    # x * keep exactly as the oracle's `x * keep.view(1, -1, 1, 1, 1)` - instead of masking and re-staging the volume every step
    supports_present = True
    input_mask_on_load = True

    def _input_mask(self, n: int, present) -> ops.NL:
        key = ("in_keep", tuple(bool(p) for p in present))
        scale = self.pool.flat(key, n * self.in_channels)
        if getattr(scale, "_mmtta_filled", None) != n:
            keep = torch.tensor([1.0 if p else 0.0 for p in present], dtype=torch.float32, device=scale.device)
            scale.copy_(keep.repeat(n))
            scale._mmtta_filled = n
        zero = self.pool.flat("in_keep_shift", n * self.in_channels, zero=True)
        return ops.NL(zero, zero, None, None, relu=False, scale=scale, shift=zero)

    def forward_cl(self, x_cl: torch.Tensor, present=None) -> torch.Tensor:
        n, D, H, W, _ = x_cl.shape
        if present is not None and len(present) != self.in_channels:
            raise ValueError(f"modality mask has {len(present)} entries for {self.in_channels} input channels")
        x_nl = self._input_mask(n, present) if (present is not None and not all(present)) else None
        f = 2 ** len(self.strides)
        if D % f or H % f or W % f:
            raise ValueError(
                f"input extent {(D, H, W)} is not divisible by {f}: the up path could not be concatenated with its "
                "skip (torch.cat fails in the reference as well)")
        dims = self._level_dims(D, H, W)
        c, L = self.channels, self.L
        self.dims, self.n = dims, n
        cur = x_cl
        for i in range(L):
            cat = self._cat(i, n, dims)
            out = cat[..., :c[i]]
            self._run_down(self.down[i], cur, out, x_nl if i == 0 else None)
            cur = out
        cat = self._cat(L - 1, n, dims)
        self._run_down(self.bottom, cur, cat[..., c[L - 1]:])
        logits = None
        for i in range(L - 1, -1, -1):
            cat = self._cat(i, n, dims)
            if i > 0:
                out = self._cat(i - 1, n, dims)[..., c[i - 1]:]
            else:
                out = self.pool.cl("logits", n, D, H, W, self.out_channels, ldc=(self.out_channels + 3) // 4 * 4)
                logits = out
            if self.upru[i] is not None:
                yt, nl = self.upconv[i].fwd(cat, None)
                self.upru[i].fwd(yt, nl, out)
            elif self.upconv[i].norm is not None:
                yt, nl = self.upconv[i].fwd(cat, None)
                ops.combine(yt, nl, None, None, out)
            else:
                self.upconv[i].fwd(cat, None, y=out)
        self.x_cl = x_cl
        return logits

    def _bwd_down(self, blk, dout, dx, need_dx):
        if isinstance(blk, ResidualUnitBlock):
            blk.bwd(dout, dx, accumulate=True, need_dx=need_dx)
        else:
            blk.bwd(dout, dx, accumulate=True, need_dx=need_dx)

    def backward_cl(self, dlogits: torch.Tensor) -> None:
        n, dims, c, L = self.n, self.dims, self.channels, self.L
        d = dlogits
        for i in range(L):
            dcat = self._cat(i, n, dims, grad=True)
            if self.upru[i] is not None:
                yt = self.upconv[i].saved[2]
                dT = self.pool.cl(("dT", i), *yt.shape, dtype=self.grad_dtype(yt.shape[-1], like=d))
                self.upru[i].bwd(d, dT, accumulate=False, need_dx=True)
                self.upconv[i].bwd(dT, dcat, accumulate=False)
            else:
                self.upconv[i].bwd(d, dcat, accumulate=False)
            d = dcat[..., c[i]:]
        dcat = self._cat(L - 1, n, dims, grad=True)
        self._bwd_down(self.bottom, d, dcat[..., :c[L - 1]], True)
        for i in range(L - 1, -1, -1):
            dout = self._cat(i, n, dims, grad=True)[..., :c[i]]
            if i > 0:
                self._bwd_down(self.down[i], dout, self._cat(i - 1, n, dims, grad=True)[..., :c[i - 1]], True)
            else:
                self._bwd_down(self.down[0], dout, None, False)


@register_model("unet")
class UNet(HipSegModel):
    runtime_cls = UNetRuntime

    def __init__(self, cfg: Dict[str, Any], in_channels: Optional[int] = None, eps: Optional[float] = None):
        super().__init__()
        cfg = as_cfg(cfg)
        c_in_cfg = get_config(cfg, "in_channels", 3)
        c_in = in_channels if in_channels is not None else (None if c_in_cfg == "auto" else int(c_in_cfg))
        if c_in is None:
            raise ValueError("[UNet] in_channels is 'auto'; please pass in_channels at construction time.")
        self.in_channels = c_in
        self.out_channels = int(get_config(cfg, "num_classes", 1))
        self.channels = list(get_config(cfg, "channels", [32, 64, 128, 256, 512]))
        self.strides = list(get_config(cfg, "strides", [2, 2, 2, 2]))
        self.num_res_units = int(get_config(cfg, "num_res_units", 0))
        self.act = get_config(cfg, "act", "relu")
        self.norm = get_config(cfg, "norm", "BATCH")
        self.dropout = float(get_config(cfg, "dropout", 0.0))
        spatial_dims = int(get_config(cfg, "spatial_dims", 3))
        if spatial_dims != 3:
            raise NotImplementedError("the MI355X adaptation path is 3-D (spatial_dims=3)")
        if len(self.channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        if len(self.strides) < len(self.channels) - 1:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        self.strides = self.strides[: len(self.channels) - 1]
        if any(s not in (1, 2) for s in self.strides):
            raise NotImplementedError("strides other than 1 or 2")
        if any(s != 2 for s in self.strides):
            raise NotImplementedError("the transposed-conv kernel is stride 2 (every shipped config uses [2,2,2,2])")

        def block(inc: int, outc: int, chans: Sequence[int], strs: Sequence[int], is_top: bool) -> nn.Module:
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub = block(c, c, chans[1:], strs[1:], False)
                upc = c * 2
            else:
                sub = self._down(c, chans[1], 1)
                upc = c + chans[1]
            return Seq(self._down(inc, c, s), SkipConnection(sub), self._up(upc, outc, s, is_top))

        self.model = block(self.in_channels, self.out_channels, self.channels, self.strides, True)

    def _down(self, inc: int, outc: int, stride: int) -> nn.Module:
        if self.num_res_units > 0:
            return ResidualUnit(inc, outc, strides=stride, subunits=self.num_res_units, act=self.act, norm=self.norm,
                                dropout=self.dropout)
        return Convolution(inc, outc, strides=stride, act=self.act, norm=self.norm, dropout=self.dropout)

    def _up(self, inc: int, outc: int, stride: int, is_top: bool) -> nn.Module:
        conv = Convolution(inc, outc, strides=stride, act=self.act, norm=self.norm, dropout=self.dropout,
                           conv_only=is_top and self.num_res_units == 0, is_transposed=True)
        if self.num_res_units > 0:
            ru = ResidualUnit(outc, outc, strides=1, subunits=1, act=self.act, norm=self.norm, dropout=self.dropout,
                              last_conv_only=is_top)
            return Seq(conv, ru)
        return conv
