"""Oracle restatement of the reference's multimodal deep-fusion U-Net (test infrastructure).

Follows reference src/models/unet_multimodal_midfusion.py: SpecificEncoder :16-77,
CompositionalLayer :80-96, DecoderStage :99-136, MultimodalUNetDeepFusion :139-270
(wiring of forward :204-267).  Parity unpinned (MONAI blocks restated in oracle/blocks.py).

``present`` extends the reference for the missing-modality configs:
the M-way means (:221, :229, :247) are taken over present modalities only and an absent
branch feeds the shared mean to ``bottleneck_reduce``.  With ``present=None`` the forward is
the reference's.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .blocks import Convolution, ResidualUnit, UpSample
from .unet import cfg_get


class SpecificEncoder(nn.Module):
    def __init__(self, spatial_dims, in_channels, channels, strides, num_res_units, act, norm, dropout):
        super().__init__()
        self.layers = nn.ModuleList()
        cur = in_channels
        for out_ch, s in zip(channels, list(strides) + [1]):
            self.layers.append(
                ResidualUnit(spatial_dims, cur, out_ch, strides=s, kernel_size=3, subunits=num_res_units,
                             act=act, norm=norm, dropout=dropout)
            )
            cur = out_ch

    def forward(self, x):
        skips: List[torch.Tensor] = []
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < last:
                skips.append(x)
        glob = torch.mean(x, dim=[2, 3, 4], keepdim=True)
        return x, glob, skips


class CompositionalLayer(nn.Module):
    def __init__(self, in_channels, spatial_dims, norm, act):
        super().__init__()
        self.fusion_conv = Convolution(spatial_dims, in_channels * 2, in_channels, kernel_size=3, strides=1,
                                       act=act, norm=norm)

    def forward(self, f_shared, f_specific):
        return f_shared + self.fusion_conv(torch.cat([f_shared, f_specific], dim=1))


class DecoderStage(nn.Module):
    def __init__(self, spatial_dims, in_channels, skip_channels, out_channels, stride, num_res_units, act, norm,
                 dropout):
        super().__init__()
        self.upsample = UpSample(spatial_dims, in_channels, out_channels, scale_factor=stride, mode="nontrainable")
        self.conv = ResidualUnit(spatial_dims, out_channels + skip_channels, out_channels, strides=1, kernel_size=3,
                                 subunits=num_res_units, act=act, norm=norm, dropout=dropout)

    def forward(self, x, skip):
        return self.conv(torch.cat([self.upsample(x), skip], dim=1))


class MultimodalUNetDeepFusion(nn.Module):
    def __init__(self, cfg: Dict[str, Any]):
        super().__init__()
        self.num_modalities = int(cfg_get(cfg, "num_modalities", 4))
        num_classes = int(cfg_get(cfg, "num_classes", 3))
        sd = int(cfg_get(cfg, "spatial_dims", 3))
        channels = list(cfg_get(cfg, "channels", [32, 64, 128, 256, 512]))
        strides = list(cfg_get(cfg, "strides", [2, 2, 2, 2]))
        nru = int(cfg_get(cfg, "num_res_units", 2))
        act = cfg_get(cfg, "act", "RELU")
        norm = cfg_get(cfg, "norm", "INSTANCE")
        dropout = float(cfg_get(cfg, "dropout", 0.0))
        dom = cfg_get(cfg, "domain_classifier", {})
        self.domain_enabled = bool(cfg_get(dom, "enabled", True))
        self.domain_loss_weight = float(cfg_get(dom, "loss_weight", 0.1))

        self.specific_encoders = nn.ModuleList(
            [SpecificEncoder(sd, 1, channels, strides, nru, act, norm, dropout) for _ in range(self.num_modalities)]
        )
        self.fusion_layer = CompositionalLayer(channels[-1], sd, norm, act)
        self.bottleneck_reduce = nn.Conv3d(channels[-1] * self.num_modalities, channels[-1], 1, bias=False)
        self.decoder_stages = nn.ModuleList()
        skip_ch = [channels[2], channels[1], channels[0], 1]
        for i in range(len(channels) - 1):
            idx = len(channels) - 1 - i
            self.decoder_stages.append(
                DecoderStage(sd, channels[idx], skip_ch[i], channels[idx - 1], strides[idx - 1], nru, act, norm,
                             dropout)
            )
        self.final_conv = nn.Conv3d(channels[0], num_classes, kernel_size=1)
        if self.domain_enabled:
            self.domain_classifier = nn.Linear(channels[-1], self.num_modalities)

    def forward(self, x: torch.Tensor, return_domain_logits: bool = False,
                return_intermediate_features: bool = False, present: Optional[Sequence[bool]] = None):
        B, C = x.shape[:2]
        modalities = torch.split(x, 1, dim=1)
        feats, globs, all_skips = [], [], []
        for enc, modal in zip(self.specific_encoders, modalities):
            f, g, s = enc(modal)
            feats.append(f)
            globs.append(g)
            all_skips.append(s)
        keep = list(range(len(feats))) if present is None else [i for i, p in enumerate(present) if p]

        def mean_over(ts):
            return torch.stack([ts[i] for i in keep], dim=0).mean(dim=0)

        shared = mean_over(feats)
        fused = [self.fusion_layer(shared, feats[i]) if i in keep else shared for i in range(len(feats))]
        x_dec = self.bottleneck_reduce(torch.cat(fused, dim=1))
        fused_skips = [mean_over([m[i] for m in all_skips]) for i in range(len(all_skips[0]))]
        input_mean = mean_over(list(modalities))
        skips_for_dec = [fused_skips[2], fused_skips[1], fused_skips[0], input_mean]
        for i, stage in enumerate(self.decoder_stages):
            x_dec = stage(x_dec, skips_for_dec[i])
        logits = self.final_conv(x_dec)
        if return_intermediate_features and self.domain_enabled:
            shared_rep = [shared.mean(dim=[2, 3, 4]) for _ in range(C)]
            return logits, shared_rep, [g.view(B, -1) for g in globs]
        if return_domain_logits and self.domain_enabled:
            return logits, self.domain_classifier(torch.cat(globs, dim=0).view(B * C, -1))
        return logits

    def get_domain_loss_weight(self) -> float:
        return self.domain_loss_weight if getattr(self, "domain_enabled", False) else 0.0
