"""``unet_multimodal_deepfusion`` / ``unet_multimodal_midfusion``: the reference's multimodal deep-fusion
U-Net (reference src/models/unet_multimodal_midfusion.py:16-270), computed by libmmtta.so.

Same registry names (:139-140), config keys and defaults (:147-160), ``state_dict`` keys (SURVEY.md
Appendix A.6) and ``forward`` contract (:204-209).  What the MI355X build does differently from the
reference's Python-serial graph:

* the M per-modality encoders read their modality straight out of the channels-last input (a C=1
  channel slice) - no ``torch.split``;
* every ``torch.stack(...).mean`` (:221, :229, :247) is one linear-combination kernel that writes
  into the consumer's concat slice; every ``torch.cat`` (:94, :135, :223) is a slice write;
* the fusion convolution's weights are shared by the M calls (:222): its weight gradient accumulates
  across calls in the same arena slot;
* ``fused_skips[3]`` (computed and discarded by the reference, :226-229/:250) is not computed.

``present`` (missing-modality configs): means run over present modalities only,
an absent branch feeds the shared mean to ``bottleneck_reduce`` and its encoder is skipped.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..config import as_cfg, get_config
from ..engine import (ConvolutionBlock, Runtime, build_convolution, build_convolution_family, build_residual_unit,
                      build_residual_unit_family)
from ..registry import register_model
from .base import HipSegModel
from .containers import Convolution, Holder, ResidualUnit, UpSample


class SpecificEncoder(Holder):
    def __init__(self, in_channels, channels, strides, num_res_units, act, norm, dropout):
        super().__init__()
        self.layers = nn.ModuleList()
        cur = in_channels
        for out_ch, s in zip(channels, list(strides) + [1]):
            self.layers.append(ResidualUnit(cur, out_ch, strides=s, kernel_size=3, subunits=num_res_units, act=act,
                                            norm=norm, dropout=dropout))
            cur = out_ch


class CompositionalLayer(Holder):
    def __init__(self, in_channels, norm, act):
        super().__init__()
        self.fusion_conv = Convolution(in_channels * 2, in_channels, kernel_size=3, strides=1, act=act, norm=norm)


class DecoderStage(Holder):
    def __init__(self, in_channels, skip_channels, out_channels, stride, num_res_units, act, norm, dropout):
        super().__init__()
        self.upsample = UpSample(in_channels, out_channels, scale_factor=stride)
        self.conv = ResidualUnit(out_channels + skip_channels, out_channels, strides=1, kernel_size=3,
                                 subunits=num_res_units, act=act, norm=norm, dropout=dropout)


class DeepFusionLoopRuntime(Runtime):
    """One modality encoder after another, the way the reference runs them (:214-218).  Kept for models whose norm layers
    carry parameters or cross-item statistics (BATCH / GROUP / affine norms): those cannot share one norm object across a
    family of encoders.  The shipped INSTANCE-norm configs run on :class:`DeepFusionRuntime`."""
    supports_present = True

    def __init__(self, model: "MultimodalUNetDeepFusion", device: torch.device):
        super().__init__(device, model.conv_dtype)
        self.M = model.num_modalities
        self.in_channels = self.M
        self.channels = list(model.channels)
        self.out_channels = model.num_classes
        c = self.channels
        self.enc = [[build_residual_unit(self, f"specific_encoders.{m}.layers.{i}", ru)
                     for i, ru in enumerate(encm.layers)] for m, encm in enumerate(model.specific_encoders)]
        first = build_convolution(self, "fusion_layer.fusion_conv", model.fusion_layer.fusion_conv)
        # one block object per call site; all share the first one's conv / norm layers (shared weights)
        self.fusion = [first] + [ConvolutionBlock(self, first.conv, first.norm) for _ in range(self.M - 1)]
        self.bott = self.make_conv("bottleneck_reduce", model.bottleneck_reduce)
        self.pre, self.dec = [], []
        for j, st in enumerate(model.decoder_stages):
            if not hasattr(st.upsample, "preconv"):
                raise NotImplementedError("decoder stage without preconv (in_channels == out_channels)")
            self.pre.append(self.make_conv(f"decoder_stages.{j}.upsample.preconv", st.upsample.preconv))
            self.dec.append(build_residual_unit(self, f"decoder_stages.{j}.conv", st.conv))
        self.final = self.make_conv("final_conv", model.final_conv)
        self.nstage = len(self.dec)

    # ---------------------------------------------------------------- forward
    def forward_cl(self, x_cl: torch.Tensor, present: Optional[Sequence[bool]] = None) -> torch.Tensor:
        n, D, H, W, M = x_cl.shape
        if M != self.M:
            raise ValueError(f"model has {self.M} modality encoders but the input has {M} channels "
                             "(reference: zip() would silently truncate, unet_multimodal_midfusion.py:214)")
        f = 2 ** (len(self.channels) - 1)
        if D % f or H % f or W % f:
            raise ValueError(f"input extent {(D, H, W)} is not divisible by {f}")
        keep = [m for m in range(M) if present is None or present[m]]
        if not keep:
            raise ValueError("at least one modality must be present")
        K = len(keep)
        c, pool = self.channels, self.pool
        nlev = len(c)                                  # 5 encoder layers
        dims = [(D, H, W)]
        for i in range(nlev - 1):
            d, h, w = dims[-1]
            dims.append((d // 2, h // 2, w // 2))
        bd = dims[nlev - 1]                            # bottleneck extent (layer 4 keeps it)
        self.state = dict(n=n, dims=dims, keep=keep, x=x_cl)
        # encoders ---------------------------------------------------------------------------------
        skips: List[List[torch.Tensor]] = [[None] * M for _ in range(nlev - 1)]
        catf = [pool.cl(("catf", m), n, *bd, 2 * c[-1]) for m in range(M)]
        for m in keep:
            cur = x_cl[..., m:m + 1]
            for i in range(nlev - 1):
                out = pool.cl(("skip", m, i), n, *dims[i + 1], c[i])
                self.enc[m][i].fwd(cur, None, out)
                skips[i][m] = out
                cur = out
            self.enc[m][nlev - 1].fwd(cur, None, catf[m][..., c[-1]:])
        # fusion -----------------------------------------------------------------------------------
        shared = pool.cl("shared", n, *bd, c[-1])
        ops.lincomb([catf[m][..., c[-1]:] for m in keep], [1.0 / K] * K, shared)
        bcat = pool.cl("bcat", n, *bd, M * c[-1])
        for m in range(M):
            dst = bcat[..., m * c[-1]:(m + 1) * c[-1]]
            if m in keep:
                ops.lincomb([shared], [1.0], catf[m][..., :c[-1]])
                y, nl = self.fusion[m].fwd(catf[m], None)
                ops.combine(y, nl, shared, None, dst)
            else:
                ops.lincomb([shared], [1.0], dst)
        xdec = pool.cl(("xdec", -1), n, *bd, c[-1])
        self.bott.op.forward(bcat, None, None, xdec)
        # decoder ----------------------------------------------------------------------------------
        skip_src = [2, 1, 0, None]                     # fused_skips[2], [1], [0], input mean
        cur = xdec
        self.cats = []
        for j in range(self.nstage):
            cin, cout = c[nlev - 1 - j], c[nlev - 2 - j]
            lo, hi = dims[nlev - 1 - j], dims[nlev - 2 - j]
            p = pool.cl(("pre", j), n, *lo, cout)
            self.pre[j].op.forward(cur, None, self.pre[j].bias_data(), p)
            sc = c[skip_src[j]] if skip_src[j] is not None else 1
            cat = pool.cl(("dcat_in", j), n, *hi, cout + sc, ldc=(cout + sc + 3) // 4 * 4)
            ops.upsample2x_fwd(p, cat[..., :cout])
            if skip_src[j] is not None:
                ops.lincomb([skips[skip_src[j]][m] for m in keep], [1.0 / K] * K, cat[..., cout:])
            else:
                ops.lincomb([x_cl[..., m:m + 1] for m in keep], [1.0 / K] * K, cat[..., cout:])
            out = pool.cl(("xdec", j), n, *hi, cout)
            self.dec[j].fwd(cat, None, out)
            self.cats.append((cur, p, cat))
            cur = out
        logits = pool.cl("logits", n, D, H, W, self.out_channels, ldc=(self.out_channels + 3) // 4 * 4)
        self.final.op.forward(cur, None, self.final.bias_data(), logits)
        self.state.update(catf=catf, shared=shared, bcat=bcat, xdec=xdec, last=cur, skips=skips)
        return logits

    # ---------------------------------------------------------------- auxiliary outputs
    def _global_mean(self, key, feat: torch.Tensor) -> torch.Tensor:
        """mean over the voxels of a channels-last feature map -> [n, C] (reference :76 ``torch.mean(x, dim=[2,3,4])``):
        the two-stage per-(n,c) sum kernels of the norm layers, count = voxels."""
        n, d, h, w, c = feat.shape
        rows = ops.reduce_rows_per_n(feat)
        part = self.pool.flat((key, "gpart"), n * rows * 2 * c)
        mean = self.pool.flat((key, "gmean"), n * c)
        rstd = self.pool.flat((key, "grstd"), n * c)
        scratch = self.pool.flat((key, "gtot"), n * c * 2, dtype=torch.float64)
        ops.channel_stats(feat, part)
        ops.norm_stats_finalize(ops.NORM_INSTANCE, 1, part, rows, n, c, d * h * w, 1e-5, True, None, None, 0.1, mean, rstd,
                                scratch)
        return mean.view(n, c)

    def global_features(self):
        """(specific_globals_flat: list of M [B, C_b], shared global [B, C_b]) of the last forward."""
        st, cb = self.state, self.channels[-1]
        spec = [self._global_mean(("spec", m), st["catf"][m][..., cb:]).clone() for m in range(self.M)]
        shared = self._global_mean("sharedg", st["shared"]).clone()
        return spec, shared

    def domain_logits(self, spec: List[torch.Tensor], classifier: nn.Linear) -> torch.Tensor:
        """``domain_classifier(torch.cat(specific_globals, dim=0).view(B*C, -1))`` (reference :263-264): rows are
        modality-major; the Linear runs as a 1x1x1 convolution over B*M one-voxel items."""
        B, cb = spec[0].shape
        g = torch.cat(spec, dim=0).contiguous().view(B * self.M, 1, 1, 1, cb)
        if getattr(self, "_dom_op", None) is None:
            self._dom_op = ops.ConvOp(cb, self.M, 1, 1, False, self.device, dtype=ops.F32)
        w = classifier.weight.detach().to(self.device, torch.float32).contiguous().view(self.M, cb, 1, 1, 1)
        self._dom_op.pack(w)
        out = ops.new_cl(B * self.M, 1, 1, 1, self.M, self.device, ldc=(self.M + 3) // 4 * 4)
        self._dom_op.forward(g, None, classifier.bias.detach().to(self.device, torch.float32).contiguous(), out)
        return out.reshape(B * self.M, -1)[:, :self.M].clone()

    # ---------------------------------------------------------------- backward
    def backward_cl(self, dlogits: torch.Tensor) -> None:
        st, pool, c = self.state, self.pool, self.channels
        n, dims, keep = st["n"], st["dims"], st["keep"]
        K, M, nlev = len(keep), self.M, len(c)
        bd = dims[nlev - 1]
        # silent parameters of this step (encoders of absent modalities) contribute a zero gradient
        for m in range(M):
            if m not in keep:
                for ru in self.enc[m]:
                    for unit in ru.units:
                        unit.conv.weight.grad.zero_()
                        if unit.conv.bias is not None:
                            unit.conv.bias.grad.zero_()
                    if ru.residual is not None:
                        ru.residual.weight.grad.zero_()
                        if ru.residual.bias is not None:
                            ru.residual.bias.grad.zero_()
        last = st["last"]
        self.final.wgrad(last, None, dlogits)
        d = pool.cl(("dxdec", self.nstage - 1), *last.shape)
        self.final.op.dgrad(dlogits, d)
        dskips = {}
        for j in range(self.nstage - 1, -1, -1):
            src, p, cat = self.cats[j]
            cout = p.shape[-1]
            dcat = pool.cl(("dcat", j), *cat.shape[:4], cat.shape[-1], ldc=(cat.shape[-1] + 3) // 4 * 4)
            self.dec[j].bwd(d, dcat, accumulate=False, need_dx=True)
            dskips[j] = dcat[..., cout:]
            dp = pool.cl(("dpre", j), *p.shape)
            ops.upsample2x_bwd(dcat[..., :cout], dp)
            self.pre[j].wgrad(src, None, dp)
            d = pool.cl(("dxdec", j - 1), *src.shape)
            self.pre[j].op.dgrad(dp, d)
        bcat = st["bcat"]
        self.bott.wgrad(bcat, None, d)
        dbcat = pool.cl("dbcat", *bcat.shape)
        self.bott.op.dgrad(d, dbcat)
        # fusion: fused_m = shared + T(conv(cat[shared, feat_m]))
        dcatf = {}
        terms = []
        for idx, m in enumerate(keep):
            dfused = dbcat[..., m * c[-1]:(m + 1) * c[-1]]
            dcatf[m] = pool.cl(("dcatf", m), *st["catf"][m].shape)
            self.fusion[m].bwd(dfused, dcatf[m], accumulate=False, need_dx=True, grad_accumulate=idx > 0)
            terms += [dfused, dcatf[m][..., :c[-1]]]
        for m in range(M):
            if m not in keep:
                terms.append(dbcat[..., m * c[-1]:(m + 1) * c[-1]])
        dshared = pool.cl("dshared", *st["shared"].shape)
        for k0 in range(0, len(terms), 8):
            chunk = terms[k0:k0 + 8]
            ops.lincomb(chunk, [1.0] * len(chunk), dshared, accumulate=k0 > 0)
        skip_src = [2, 1, 0, None]
        for m in keep:
            dfeat = dcatf[m][..., c[-1]:]
            ops.lincomb([dshared], [1.0 / K], dfeat, accumulate=True)
            dout = dfeat
            for i in range(nlev - 1, -1, -1):
                if i == 0:
                    self.enc[m][0].bwd(dout, None, need_dx=False)
                    break
                dx = pool.cl(("dskip", m, i - 1), *st["skips"][i - 1][m].shape)
                stage = skip_src.index(i - 1) if (i - 1) in skip_src else None
                if stage is not None:
                    ops.lincomb([dskips[stage]], [1.0 / K], dx)
                    self.enc[m][i].bwd(dout, dx, accumulate=True, need_dx=True)
                else:
                    self.enc[m][i].bwd(dout, dx, accumulate=False, need_dx=True)
                dout = dx


class DeepFusionRuntime(Runtime):
    """The M modality encoders as ONE launch sequence: layer i of every encoder is a layer *family* (engine.ConvLayer
    members, mmtta_param_sets ``inner`` = M) whose batch items are (volume, modality) pairs, so a launch at the 8^3 / 16^3 /
    32^3 levels carries M (x ``method.group``) times the workgroups of the reference's one-encoder-at-a-time loop
    (reference src/models/unet_multimodal_midfusion.py:214-218; SURVEY.md row a2 "batchable as grouped conv").  The fusion
    convolution the reference applies M times with shared weights (:222) is one launch over the same batch with
    ``items_per_set`` = M: its weight gradient sums over the M items of a volume inside the launch.

    Batch layout of every encoder-side tensor: [n * M, ...], item v * M + m = (volume v, modality m).  An ABSENT modality
    (``present``, missing-modality configs) still runs through its encoder - its input is all zeros - but is left out of
    every mean, feeds the shared mean to ``bottleneck_reduce`` and receives zero gradients, which is exactly what skipping
    it gives (its weight gradients come out as exact zeros)."""
    supports_present = True

    def __init__(self, model: "MultimodalUNetDeepFusion", device: torch.device):
        super().__init__(device, model.conv_dtype, group=getattr(model, "group", 1))
        self.M = model.num_modalities
        self.in_channels = self.M
        self.channels = list(model.channels)
        self.out_channels = model.num_classes
        M, nlev = self.M, len(self.channels)
        # bf16 precision stores the wide forward activations as bf16 (method.storage: bf16), like the U-Net: every layer kind
        # of this runtime has storage-agnostic kernels (one-channel stems on the matrix-core thin-K kernel, trilinear
        # resample, modality means, the 1x1x1 head) - and since round 3 so has every consumer of a wide GRADIENT
        # (method.grad_storage: the head's input gradient, trilinear-resample backward, the linear combinations).
        self.act_bf16 = (model.conv_dtype == ops.BF16 and getattr(model, "act_storage", "bf16") == "bf16"
                         and self.channels[0] in (32, 64) and all(ch % 8 == 0 and ch >= 32 for ch in self.channels)
                         and self.out_channels <= 4)
        self.grad_bf16 = self.act_bf16 and getattr(model, "grad_storage", "bf16") == "bf16"
        encs = list(model.specific_encoders)
        self.encf = [build_residual_unit_family(self, [f"specific_encoders.{m}.layers.{i}" for m in range(M)],
                                                [e.layers[i] for e in encs]) for i in range(nlev)]
        self.fusion = build_convolution_family(self, ["fusion_layer.fusion_conv"], [model.fusion_layer.fusion_conv],
                                               items_per_set=M)
        self.bott = self.make_conv("bottleneck_reduce", model.bottleneck_reduce)
        self.pre, self.dec = [], []
        for j, st in enumerate(model.decoder_stages):
            if not hasattr(st.upsample, "preconv"):
                raise NotImplementedError("decoder stage without preconv (in_channels == out_channels)")
            self.pre.append(self.make_conv(f"decoder_stages.{j}.upsample.preconv", st.upsample.preconv))
            self.dec.append(build_residual_unit(self, f"decoder_stages.{j}.conv", st.conv))
        self.final = self.make_conv("final_conv", model.final_conv)
        self.nstage = len(self.dec)

    # ---------------------------------------------------------------- input
    def stage_input(self, x: torch.Tensor) -> torch.Tensor:
        """x [n, M, D, H, W] -> the channels-last volume (decoder skip: the mean over modalities) AND the family batch
        [n * M, D, H, W, 1] the encoders read (one one-channel item per (volume, modality))."""
        x_cl = super().stage_input(x)
        n, M, D, H, W = x.shape
        xm = self.pool.cl("xm", n * M, D, H, W, 1, ldc=4, zero=True)
        ops.to_cl(x.contiguous().view(n * M, 1, D, H, W), out=xm)
        return x_cl

    @staticmethod
    def _member(t: torch.Tensor, M: int, m: int) -> torch.Tensor:
        """[n * M, ...] family batch -> the [n, ...] view of member m (batch stride M items)."""
        return t.view(t.shape[0] // M, M, *t.shape[1:])[:, m]

    # ---------------------------------------------------------------- forward
    def forward_cl(self, x_cl: torch.Tensor, present: Optional[Sequence[bool]] = None) -> torch.Tensor:
        n, D, H, W, M = x_cl.shape
        if M != self.M:
            raise ValueError(f"model has {self.M} modality encoders but the input has {M} channels "
                             "(reference: zip() would silently truncate, unet_multimodal_midfusion.py:214)")
        f = 2 ** (len(self.channels) - 1)
        if D % f or H % f or W % f:
            raise ValueError(f"input extent {(D, H, W)} is not divisible by {f}")
        keep = [m for m in range(M) if present is None or present[m]]
        if not keep:
            raise ValueError("at least one modality must be present")
        K = len(keep)
        c, pool = self.channels, self.pool
        nlev = len(c)
        dims = [(D, H, W)]
        for i in range(nlev - 1):
            d, h, w = dims[-1]
            dims.append((d // 2, h // 2, w // 2))
        bd = dims[nlev - 1]
        mem = lambda t, m: self._member(t, M, m)
        xm = pool.cl("xm", n * M, D, H, W, 1, ldc=4, zero=True)          # filled by stage_input
        self.state = dict(n=n, dims=dims, keep=keep, x=x_cl)
        # encoders: every layer ONE launch sequence over the n * M (volume, modality) items -------------------
        skips: List[torch.Tensor] = []
        act = self.act_dtype
        catf = pool.cl("catf", n * M, *bd, 2 * c[-1], dtype=act(2 * c[-1]))
        cur = xm
        for i in range(nlev - 1):
            out = pool.cl(("skip", i), n * M, *dims[i + 1], c[i], dtype=act(c[i]))
            self.encf[i].fwd(cur, None, out)
            skips.append(out)
            cur = out
        self.encf[nlev - 1].fwd(cur, None, catf[..., c[-1]:])
        # fusion -----------------------------------------------------------------------------------------------
        shared = pool.cl("shared", n, *bd, c[-1], dtype=act(c[-1]))
        ops.lincomb([mem(catf, m)[..., c[-1]:] for m in keep], [1.0 / K] * K, shared)
        for m in range(M):
            ops.lincomb([shared], [1.0], mem(catf, m)[..., :c[-1]])
        y, nl = self.fusion.fwd(catf, None)                               # shared weights: one set per volume, M items each
        fused = pool.cl("fused", n * M, *bd, c[-1], dtype=act(c[-1]))
        ops.combine(y, nl, catf[..., :c[-1]], None, fused)                # shared + relu(norm(conv(cat[shared, feat_m])))
        bcat = pool.cl("bcat", n, *bd, M * c[-1], dtype=act(M * c[-1]))
        for m in range(M):
            src = mem(fused, m) if m in keep else shared                  # an absent branch feeds the shared mean
            ops.lincomb([src], [1.0], bcat[..., m * c[-1]:(m + 1) * c[-1]])
        xdec = pool.cl(("xdec", -1), n, *bd, c[-1], dtype=act(c[-1]))
        self.bott.op.forward(bcat, None, None, xdec)
        # decoder ----------------------------------------------------------------------------------------------
        skip_src = [2, 1, 0, None]                     # fused_skips[2], [1], [0], input mean
        cur = xdec
        self.cats = []
        for j in range(self.nstage):
            cin, cout = c[nlev - 1 - j], c[nlev - 2 - j]
            lo, hi = dims[nlev - 1 - j], dims[nlev - 2 - j]
            p = pool.cl(("pre", j), n, *lo, cout, dtype=act(cout))
            self.pre[j].op.forward(cur, None, self.pre[j].bias_data(), p)
            sc = c[skip_src[j]] if skip_src[j] is not None else 1
            # rows padded to 4 floats / 8 bf16; zero-filled once: the pad lanes behind a ragged channel count (32 + 1) are read
            # by the 8-channel loaders of the implicit GEMM and must hold finite values
            cat = pool.cl(("dcat_in", j), n, *hi, cout + sc, zero=True, dtype=act(cout + sc))
            ops.upsample2x_fwd(p, cat[..., :cout])
            if skip_src[j] is not None:
                ops.lincomb([mem(skips[skip_src[j]], m) for m in keep], [1.0 / K] * K, cat[..., cout:])
            else:
                ops.lincomb([x_cl[..., m:m + 1] for m in keep], [1.0 / K] * K, cat[..., cout:])
            out = pool.cl(("xdec", j), n, *hi, cout, dtype=act(cout))
            self.dec[j].fwd(cat, None, out)
            self.cats.append((cur, p, cat))
            cur = out
        logits = pool.cl("logits", n, D, H, W, self.out_channels, ldc=(self.out_channels + 3) // 4 * 4)
        self.final.op.forward(cur, None, self.final.bias_data(), logits)
        self.state.update(catf=catf, shared=shared, bcat=bcat, xdec=xdec, last=cur, skips=skips, fused=fused)
        return logits

    # ---------------------------------------------------------------- auxiliary outputs
    _global_mean = DeepFusionLoopRuntime._global_mean
    domain_logits = DeepFusionLoopRuntime.domain_logits

    def global_features(self):
        """(specific_globals_flat: list of M [B, C_b], shared global [B, C_b]) of the last forward."""
        st, cb, M = self.state, self.channels[-1], self.M
        spec_all = self._global_mean("spec", st["catf"][..., cb:]).clone()          # [n * M, C_b], item v * M + m
        n = spec_all.shape[0] // M
        spec = [spec_all.view(n, M, cb)[:, m].contiguous() for m in range(M)]
        shared = self._global_mean("sharedg", st["shared"]).clone()
        return spec, shared

    # ---------------------------------------------------------------- backward
    def backward_cl(self, dlogits: torch.Tensor) -> None:
        st, pool, c = self.state, self.pool, self.channels
        n, dims, keep = st["n"], st["dims"], st["keep"]
        K, M, nlev = len(keep), self.M, len(c)
        mem = lambda t, m: self._member(t, M, m)
        last = st["last"]
        # gradients of the wide tensors in the runtime's gradient storage (method.grad_storage: bf16 next to bf16-stored
        # activations): every buffer below is read by convolutions that round it to bf16 anyway, by the norm backward and by
        # linear combinations with fp32 arithmetic
        gd = lambda t: dict(dtype=self.grad_dtype(t.shape[-1]))
        self.final.wgrad(last, None, dlogits)
        d = pool.cl(("dxdec", self.nstage - 1), *last.shape, **gd(last))
        self.final.op.dgrad(dlogits, d)
        dskips = {}
        for j in range(self.nstage - 1, -1, -1):
            src, p, cat = self.cats[j]
            cout = p.shape[-1]
            dcat = pool.cl(("dcat", j), *cat.shape[:4], cat.shape[-1], **gd(cat))
            self.dec[j].bwd(d, dcat, accumulate=False, need_dx=True)
            dskips[j] = dcat[..., cout:]
            dp = pool.cl(("dpre", j), *p.shape, **gd(p))
            ops.upsample2x_bwd(dcat[..., :cout], dp)
            self.pre[j].wgrad(src, None, dp)
            d = pool.cl(("dxdec", j - 1), *src.shape, **gd(src))
            self.pre[j].op.dgrad(dp, d)
        bcat = st["bcat"]
        self.bott.wgrad(bcat, None, d)
        dbcat = pool.cl("dbcat", *bcat.shape, **gd(bcat))
        self.bott.op.dgrad(d, dbcat)
        # fusion: fused_m = shared + T(conv(cat[shared, feat_m])); absent members get a ZERO output gradient
        fused = st["fused"]
        dfused = pool.cl("dfused", *fused.shape, **gd(fused))
        for m in range(M):
            ops.lincomb([dbcat[..., m * c[-1]:(m + 1) * c[-1]]], [1.0 if m in keep else 0.0], mem(dfused, m))
        dcatf = pool.cl("dcatf", *st["catf"].shape, **gd(st["catf"]))
        self.fusion.bwd(dfused, dcatf, accumulate=False, need_dx=True)
        terms = []
        for m in keep:
            terms += [dbcat[..., m * c[-1]:(m + 1) * c[-1]], mem(dcatf, m)[..., :c[-1]]]
        for m in range(M):
            if m not in keep:
                terms.append(dbcat[..., m * c[-1]:(m + 1) * c[-1]])
        dshared = pool.cl("dshared", *st["shared"].shape, **gd(st["shared"]))
        for k0 in range(0, len(terms), 8):
            chunk = terms[k0:k0 + 8]
            ops.lincomb(chunk, [1.0] * len(chunk), dshared, accumulate=k0 > 0)
        for m in keep:
            ops.lincomb([dshared], [1.0 / K], mem(dcatf, m)[..., c[-1]:], accumulate=True)
        # encoders: the family backward, every layer one launch sequence over the n * M items
        skip_src = [2, 1, 0, None]
        dout = dcatf[..., c[-1]:]
        for i in range(nlev - 1, -1, -1):
            if i == 0:
                self.encf[0].bwd(dout, None, need_dx=False)
                break
            dx = pool.cl(("dskip", i - 1), *st["skips"][i - 1].shape, **gd(st["skips"][i - 1]))
            stage = skip_src.index(i - 1) if (i - 1) in skip_src else None
            if stage is not None:
                for m in range(M):
                    ops.lincomb([dskips[stage]], [1.0 / K if m in keep else 0.0], mem(dx, m))
                self.encf[i].bwd(dout, dx, accumulate=True, need_dx=True)
            else:
                self.encf[i].bwd(dout, dx, accumulate=False, need_dx=True)
            dout = dx


def _parameter_free_norm(norm) -> bool:
    name = norm[0] if isinstance(norm, (tuple, list)) else norm
    args = norm[1] if isinstance(norm, (tuple, list)) and len(norm) > 1 else {}
    return str(name).upper() == "INSTANCE" and not bool((args or {}).get("affine", False))


@register_model("unet_multimodal_deepfusion")
@register_model("unet_multimodal_midfusion")
class MultimodalUNetDeepFusion(HipSegModel):
    def runtime_cls(self, model, device):
        """Encoders as one launch family for the shipped parameter-free norms; the modality loop otherwise."""
        return (DeepFusionRuntime if _parameter_free_norm(self.norm) else DeepFusionLoopRuntime)(model, device)

    def __init__(self, cfg: Dict[str, Any]):
        super().__init__()
        cfg = as_cfg(cfg)
        self.num_modalities = int(get_config(cfg, "num_modalities", 4))
        self.num_classes = int(get_config(cfg, "num_classes", 3))
        if int(get_config(cfg, "spatial_dims", 3)) != 3:
            raise NotImplementedError("the MI355X adaptation path is 3-D (spatial_dims=3)")
        self.channels = list(get_config(cfg, "channels", [32, 64, 128, 256, 512]))
        strides = list(get_config(cfg, "strides", [2, 2, 2, 2]))
        nru = int(get_config(cfg, "num_res_units", 2))
        act = get_config(cfg, "act", "RELU")
        norm = get_config(cfg, "norm", "INSTANCE")
        self.norm = norm
        dropout = float(get_config(cfg, "dropout", 0.0))
        if len(self.channels) != 5 or strides != [2, 2, 2, 2]:
            raise NotImplementedError("deep-fusion decoder is wired for 5 channel levels and strides [2,2,2,2] "
                                      "(reference unet_multimodal_midfusion.py:171-193)")
        dom = get_config(cfg, "domain_classifier", {}) or {}
        self.domain_enabled = bool(get_config(dom, "enabled", True))
        self.domain_loss_weight = float(get_config(dom, "loss_weight", 0.1))
        ch = self.channels
        self.specific_encoders = nn.ModuleList(
            [SpecificEncoder(1, ch, strides, nru, act, norm, dropout) for _ in range(self.num_modalities)])
        self.fusion_layer = CompositionalLayer(ch[-1], norm, act)
        self.bottleneck_reduce = nn.Conv3d(ch[-1] * self.num_modalities, ch[-1], 1, bias=False)
        self.decoder_stages = nn.ModuleList()
        skip_ch = [ch[2], ch[1], ch[0], 1]
        for i in range(len(ch) - 1):
            idx = len(ch) - 1 - i
            self.decoder_stages.append(DecoderStage(ch[idx], skip_ch[i], ch[idx - 1], strides[idx - 1], nru, act, norm,
                                                    dropout))
        self.final_conv = nn.Conv3d(ch[0], self.num_classes, kernel_size=1)
        if self.domain_enabled:
            self.domain_classifier = nn.Linear(ch[-1], self.num_modalities)

    def forward(self, x: torch.Tensor, return_domain_logits: bool = False,
                return_intermediate_features: bool = False, present: Optional[Sequence[bool]] = None):
        """Reference contract (src/models/unet_multimodal_midfusion.py:204-267): logits, or - with the domain classifier
        enabled - ``(logits, shared_globals_rep, specific_globals_flat)`` / ``(logits, domain_logits)`` (the first
        flag wins, like the reference's two ``if``s).  The auxiliary outputs are computed from the bottleneck
        features the forward just produced (global means: the per-(n,c) statistics kernels; classifier: a 1x1x1
        convolution launch) and are returned DETACHED: the reference trainer never requests them
        (seg_trainer.py:110), so no backward is wired through them."""
        aux = (return_domain_logits or return_intermediate_features) and self.domain_enabled
        if present is not None or aux:
            if x.device.type != "cuda":
                raise ops.MmttaError("this model computes on an MI355X through libmmtta.so")
        if present is not None:
            rt = self.runtime(x.device)
            rt.training = self.training
            rt.pack_all()
            logits = ops.from_cl(rt.forward_cl(rt.stage_input(x.float()), present=present))
        else:
            logits = super().forward(x)
        if not aux:
            return logits
        if present is not None and not all(present):
            raise ValueError("auxiliary outputs need every modality present (the reference has no missing-modality path)")
        rt = self.runtime(x.device)
        spec, shared = rt.global_features()
        if return_intermediate_features:
            return logits, [shared for _ in range(x.shape[1])], spec
        return logits, rt.domain_logits(spec, self.domain_classifier)

    def get_domain_loss_weight(self) -> float:
        return self.domain_loss_weight if getattr(self, "domain_enabled", False) else 0.0
