"""Host-side execution engine of the adaptation path: explicit forward AND backward of the MONAI
block vocabulary (Convolution / ADN / ResidualUnit / UpSample / SkipConnection) over
channels-last device buffers, every arithmetic step a libmmtta.so kernel.

Design (MI355X-first, not a translation of the reference's module-by-module torch graph):

* no autograd graph: each block keeps references to the buffers it needs for backward, shapes
  are static per input shape, so one adaptation step is a fixed launch sequence that is
  captured into a hipGraph and replayed (tta.py);
* norm + ReLU never materialise: a Convolution produces the raw conv output plus per-(n,c)
  mean/rstd ("lazy" output, ops.NL); the consumer applies them while staging its input;
* ``torch.cat`` never happens: producers write into channel slices of the concat buffer;
* ResidualUnit's add is the epilogue of the residual conv (or one combine kernel when the
  residual is the identity);
* parameters, gradients and Adam moments live in four flat fp32 arenas (decay segment first,
  reference src/core/experiment_manager.py:214-228), so the optimizer is one kernel launch
  and the episodic reset one memcpy.

Reference semantics of each block: SURVEY.md Appendix A (MONAI, restated in oracle/blocks.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import NL, ConvOp, MmttaError

GROUP_DECAY, GROUP_NO_DECAY, GROUP_FROZEN = 0, 1, 2


# ----------------------------------------------------------------------------- parameters
class ParamRef:
    """A parameter adopted into the arena: ``data``/``grad`` are views into the flat buffers."""

    def __init__(self, name: str, param: torch.nn.Parameter):
        self.name, self.param = name, param
        self.offset = -1
        self.numel = param.numel()
        self.shape = tuple(param.shape)
        self.group = GROUP_DECAY
        self.data: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None

    @property
    def trainable(self) -> bool:
        return self.group != GROUP_FROZEN


class Arena:
    """Flat fp32 storage [decay | no-decay | frozen] for parameters, gradients and Adam moments.

    ``replicas`` > 1 (a group of volumes adapting side by side, ``method.group``): the four buffers are ``[replicas, total]``;
    replica 0 is the one the nn.Parameters view (the source weights), replica g holds volume g's adapted copy.  ``params`` /
    ``grads`` / ``exp_avg`` / ``exp_avg_sq`` stay the flat views of replica 0; the ``*_all`` tensors are the whole thing."""

    def __init__(self, refs: Sequence[ParamRef], device: torch.device, replicas: int = 1):
        self.device = device
        self.replicas = max(1, int(replicas))
        order = sorted(range(len(refs)), key=lambda i: (refs[i].group, i))
        off = 0
        self.n_decay = self.n_train = 0
        for i in order:
            r = refs[i]
            r.offset = off
            off += (r.numel + 3) // 4 * 4          # keep every parameter 16-byte aligned
            if r.group == GROUP_DECAY:
                self.n_decay = off
            if r.group != GROUP_FROZEN:
                self.n_train = off
        self.n_decay = min(self.n_decay, self.n_train) if self.n_train else 0
        self.total = off
        self.refs = list(refs)
        self.total = max(off, 4)
        self.params_all = torch.zeros((self.replicas, self.total), dtype=torch.float32, device=device)
        self.grads_all = torch.zeros_like(self.params_all)
        self.exp_avg_all = torch.zeros_like(self.params_all)
        self.exp_avg_sq_all = torch.zeros_like(self.params_all)
        self.params, self.grads = self.params_all[0], self.grads_all[0]
        self.exp_avg, self.exp_avg_sq = self.exp_avg_all[0], self.exp_avg_sq_all[0]
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        for r in self.refs:
            view = self.params[r.offset:r.offset + r.numel].view(r.shape)
            with torch.no_grad():
                view.copy_(r.param.detach().to(device=device, dtype=torch.float32))
            r.param.data = view
            r.data = view
            r.grad = self.grads[r.offset:r.offset + r.numel].view(r.shape)
            r.param.requires_grad_(r.group != GROUP_FROZEN)
        self.source: Optional[torch.Tensor] = None

    def owns(self) -> bool:
        """True while every nn.Parameter still points into this arena (``module.to()`` can break it)."""
        base = self.params.data_ptr()
        return all(r.param.data_ptr() == base + 4 * r.offset for r in self.refs)

    def snapshot_source(self) -> None:
        self.source = self.params.clone()

    def restore_source(self) -> None:
        """Episodic reset: source weights back in every replica, optimizer state cleared."""
        if self.source is None:
            raise MmttaError("no source snapshot taken")
        self.params_all.copy_(self.source.unsqueeze(0).expand_as(self.params_all))
        # (the moments are not cleared: step 0 of mmtta_optim_step starts from zero moments whatever the buffers hold)
        self.step.zero_()

    def zero_grad(self) -> None:
        self.grads_all.zero_()

    def replica_data(self, ref: ParamRef, g: int) -> torch.Tensor:
        """Parameter ``ref`` of replica ``g`` (a view)."""
        return self.params_all[g, ref.offset:ref.offset + ref.numel].view(ref.shape)

    def publish_grads(self) -> None:
        """Expose gradients as ``param.grad`` for external torch optimizers (drop-in run_step path)."""
        for r in self.refs:
            r.param.grad = r.grad if r.trainable else None


# ----------------------------------------------------------------------------- buffer pool
class Pool:
    """Named persistent device buffers (static addresses: required for graph capture)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._b: Dict[Tuple, torch.Tensor] = {}

    def cl(self, key, n, d, h, w, c, ldc=None, zero=False, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        if ldc is None:
            ldc = ops.row_pad(c, dtype)  # aligned vector accesses per 4 (fp32) / 8 (bf16) channels, for any C
        k = ("cl", key, n, d, h, w, c, ldc, dtype)
        t = self._b.get(k)
        if t is None:
            t = ops.new_cl(n, d, h, w, c, self.device, ldc, zero=zero, dtype=dtype)
            self._b[k] = t
        return t

    def flat(self, key, numel, dtype=torch.float32, zero=False) -> torch.Tensor:
        k = ("flat", key, numel, dtype)
        t = self._b.get(k)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(max(int(numel), 1), dtype=dtype, device=self.device)
            self._b[k] = t
        return t

    def nbytes(self) -> int:
        return sum(t.untyped_storage().nbytes() for t in self._b.values())


# ----------------------------------------------------------------------------- layers
class NormLayer:
    """MONAI ADN's "N": InstanceNorm3d (no parameters), BatchNorm3d or GroupNorm (affine)."""

    def __init__(self, kind: str, channels: int, groups: int = 1, eps: float = 1e-5, momentum: float = 0.1,
                 gamma: Optional[ParamRef] = None, beta: Optional[ParamRef] = None,
                 bn_module: Optional[torch.nn.Module] = None):
        self.kind_name = kind
        self.kind = ops.NORM_KINDS[kind]
        self.C, self.groups, self.eps, self.momentum = channels, groups, eps, momentum
        self.gamma, self.beta = gamma, beta
        self.bn = bn_module        # owns running_mean / running_var / num_batches_tracked (BatchNorm only)

    def finalize(self, pool: Pool, key, part, rows_per_n: int, n: int, count: int, training: bool) -> NL:
        mean = pool.flat((key, "mean"), n * self.C)
        rstd = pool.flat((key, "rstd"), n * self.C)
        scale = pool.flat((key, "scale"), n * self.C)
        shift = pool.flat((key, "shift"), n * self.C)
        scratch = pool.flat((key, "tot"), n * self.C * 2, dtype=torch.float64)
        use_batch = training or self.kind != ops.NORM_BATCH
        rm = self.bn.running_mean if self.bn is not None else None
        rv = self.bn.running_var if self.bn is not None else None
        g = self.gamma.data if self.gamma else None
        b = self.beta.data if self.beta else None
        ops.norm_stats_finalize(self.kind, self.groups, part, rows_per_n, n, self.C, count, self.eps,
                                use_batch, rm, rv, self.momentum, mean, rstd, scratch, g, b, scale, shift)
        if self.bn is not None and training and self.bn.num_batches_tracked is not None:
            self.bn.num_batches_tracked.add_(1)
        return NL(mean, rstd, g, b, True, scale, shift)

    def backward(self, pool: Pool, key, dT: torch.Tensor, y: torch.Tensor, nl: NL, dy: torch.Tensor,
                 training: bool, accumulate: bool = False) -> None:
        n, d, h, w, c = y.shape
        train_g = self.gamma is not None and self.gamma.trainable
        # the deep levels: reduce + finalize + apply were three launch latencies; one workgroup per 32 channels does it all
        if (self.kind == ops.NORM_INSTANCE and not train_g and d * h * w <= ops.small_norm_backward_max()
                and ops.norm_bwd_small_ok(dT, y, nl, dy)):
            ops.norm_bwd_small(dT, y, nl, d * h * w, dy)
            return
        rows = ops.reduce_rows_per_n(y)
        part = pool.flat((key, "bpart"), n * rows * 2 * c)
        m1 = pool.flat((key, "m1"), n * c)
        m2 = pool.flat((key, "m2"), n * c)
        scratch = pool.flat((key, "tot"), n * c * 2, dtype=torch.float64)
        ops.norm_bwd_reduce(dT, y, nl, part)
        use_batch = training or self.kind != ops.NORM_BATCH
        ops.norm_bwd_finalize(self.kind, self.groups, part, rows, n, c, d * h * w,
                              self.gamma.data if self.gamma else None, use_batch, m1, m2,
                              self.gamma.grad if train_g else None,
                              self.beta.grad if (train_g and self.beta is not None) else None, accumulate, scratch)
        ops.norm_bwd_apply(dT, y, nl, m1, m2, dy)


class ConvLayer:
    """One convolution module - or a FAMILY of identical modules with different weights (``members``: the M modality
    encoders of the deep-fusion network) that run as the batch items of one launch."""

    def __init__(self, op: ConvOp, weight: ParamRef, bias: Optional[ParamRef], rt: Optional["Runtime"] = None,
                 members: Optional[List[Tuple[ParamRef, Optional[ParamRef]]]] = None, items_per_set: int = 1):
        self.op, self.weight, self.bias, self.rt = op, weight, bias, rt
        self.members = members if members is not None else [(weight, bias)]      # member 0 = (weight, bias)
        self.items_per_set = int(items_per_set)      # consecutive batch items sharing a set (the fusion layer: M)
        self.side_index = 0          # which side stream takes this layer's weight gradient (Runtime.make_conv)

    def pack(self) -> None:
        for m, (w, _) in enumerate(self.members):
            self.op.pack(w.data, m)

    def bind_sets(self, arena: "Arena") -> None:
        """Tell the op which batch item uses which parameter set: replica stride = the arena's, member stride = the
        distance between the members' parameters (must be uniform: the members are registered in one order)."""
        inner = len(self.members)
        if self.op.n_sets != arena.replicas * inner:
            return                     # an op without per-item sets (built before the group size was known)
        wi = bi = 0
        if inner > 1:
            wd = {self.members[m + 1][0].offset - self.members[m][0].offset for m in range(inner - 1)}
            bd = ({self.members[m + 1][1].offset - self.members[m][1].offset for m in range(inner - 1)}
                  if self.bias is not None else {0})
            if len(wd) != 1 or len(bd) != 1:
                raise MmttaError("the members of a layer family must sit at a uniform stride in the arena")
            wi, bi = wd.pop(), bd.pop()
        self.op.set_param_sets(self.items_per_set, inner, arena.total, wi, arena.total, bi, self.rt)

    def bias_data(self):
        return self.bias.data if self.bias is not None else None

    def _wgrad_launch(self, x, x_nl, dy, db, accumulate) -> None:
        self.op.wgrad(x, x_nl, dy, self.weight.grad, db, accumulate)

    def wgrad(self, x, x_nl, dy, accumulate=False) -> None:
        if not self.weight.trainable and not (self.bias is not None and self.bias.trainable):
            return
        # a frozen weight with a trainable bias still needs db; dw then lands in the (ignored) frozen grads
        db = self.bias.grad if self.bias is not None else None
        side = self.rt.side_stream(self.side_index) if self.rt is not None else None
        if side is None:
            self._wgrad_launch(x, x_nl, dy, db, accumulate)
            return
        # The weight gradient feeds nothing before the optimizer: it runs on a side stream, concurrently with the
        # input-gradient / norm-backward chain of the main stream (neither kernel fills the chip on its own).
        # Its operands (saved activations, per-block dy buffers) are never rewritten inside a step.
        side.wait_stream(torch.cuda.current_stream())
        ops.Workspace.slot = 1 + self.side_index % self.rt.n_side
        try:
            with torch.cuda.stream(side):
                self._wgrad_launch(x, x_nl, dy, db, accumulate)
        finally:
            ops.Workspace.slot = 0
        self.rt.side_pending = True


class Block:
    """Common bookkeeping: a unique key for pooled buffers, the runtime that owns the pool."""
    _next_id = 0

    def __init__(self, rt: "Runtime"):
        self.rt = rt
        Block._next_id += 1
        self.key = Block._next_id


class ConvolutionBlock(Block):
    """monai Convolution: conv (or transposed conv) [+ ADN = norm -> dropout(0) -> ReLU]."""

    def __init__(self, rt, conv: ConvLayer, norm: Optional[NormLayer]):
        super().__init__(rt)
        self.conv, self.norm = conv, norm
        self.saved = None

    def fwd(self, x: torch.Tensor, x_nl: Optional[NL], y: Optional[torch.Tensor] = None):
        op, pool = self.conv.op, self.rt.pool
        n, d, h, w, c = op.out_shape(x)
        if y is None:
            y = pool.cl((self.key, "y"), n, d, h, w, c, dtype=self.rt.act_dtype(c))
        nl = None
        if self.norm is not None:
            rows = op.stats_rows(x, y)
            stats = pool.flat((self.key, "stats"), rows * 2 * c)
            op.forward(x, x_nl, self.conv.bias_data(), y, stats=stats)
            nl = self.norm.finalize(pool, self.key, stats, rows // n, n, d * h * w, self.rt.training)
        else:
            op.forward(x, x_nl, self.conv.bias_data(), y)
        self.saved = (x, x_nl, y, nl)
        return y, nl

    def bwd(self, dT: torch.Tensor, dx: Optional[torch.Tensor], accumulate: bool = False, need_dx: bool = True,
            grad_accumulate: bool = False, add: Optional[torch.Tensor] = None) -> None:
        """``add``: another gradient term of the block's input, summed into dx by the input-gradient epilogue."""
        x, x_nl, y, nl = self.saved
        dy = dT
        if self.norm is not None:
            dy = self.rt.pool.cl((self.key, "dy"), *y.shape, dtype=dT.dtype)      # gradients keep their storage type
            self.norm.backward(self.rt.pool, self.key, dT, y, nl, dy, self.rt.training, grad_accumulate)
        self.conv.wgrad(x, x_nl, dy, grad_accumulate)
        if need_dx:
            self.conv.op.dgrad(dy, dx, accumulate, add=add)


class ResidualUnitBlock(Block):
    """monai ResidualUnit: ``conv(x) + residual(x)``, residual = identity | conv k3 (strided) | conv k1."""

    def __init__(self, rt, units: List[ConvolutionBlock], residual: Optional[ConvLayer]):
        super().__init__(rt)
        self.units, self.residual = units, residual
        self.saved = None
        self.fused_last = False

    def out_shape(self, x):
        return self.units[0].conv.op.out_shape(x)[:4] + (self.units[-1].conv.op.cout,)

    def fwd(self, x: torch.Tensor, x_nl: Optional[NL], out: torch.Tensor) -> torch.Tensor:
        cur, cur_nl = x, x_nl
        self.fused_last = False
        for u, unit in enumerate(self.units):
            last = u == len(self.units) - 1
            if last and unit.norm is None and self.residual is None:
                # conv-only last unit + identity residual: the add is this conv's epilogue
                unit.conv.op.forward(cur, cur_nl, unit.conv.bias_data(), out, add=x, add_nl=x_nl)
                unit.saved = (cur, cur_nl, out, None)
                self.fused_last = True
                self.saved = (x, x_nl)
                return out
            cur, cur_nl = unit.fwd(cur, cur_nl)
        if self.residual is not None:
            self.residual.op.forward(x, x_nl, self.residual.bias_data(), out, add=cur, add_nl=cur_nl)
        else:
            ops.combine(cur, cur_nl, x, x_nl, out)
        self.saved = (x, x_nl)
        return out

    def bwd(self, dout: torch.Tensor, dx: Optional[torch.Tensor], accumulate: bool = False, need_dx: bool = True,
            grad_accumulate: bool = False) -> None:
        x, x_nl = self.saved
        pool = self.rt.pool
        d = dout
        for u in range(len(self.units) - 1, -1, -1):
            unit = self.units[u]
            if u > 0:
                inp = unit.saved[0]
                dprev = pool.cl((self.key, "dprev", u), *inp.shape, dtype=self.rt.grad_dtype(inp.shape[-1], like=d))
                unit.bwd(d, dprev, accumulate=False, need_dx=True, grad_accumulate=grad_accumulate)
                d = dprev
            else:
                # identity residual: its gradient term (dout itself) rides in the epilogue of this input gradient
                unit.bwd(d, dx, accumulate=accumulate, need_dx=need_dx, grad_accumulate=grad_accumulate,
                         add=dout if (self.residual is None and need_dx) else None)
        if self.residual is not None:
            self.residual.wgrad(x, x_nl, dout, grad_accumulate)
            if need_dx:
                self.residual.op.dgrad(dout, dx, accumulate=True)


# ----------------------------------------------------------------------------- runtime base
class Runtime:
    """Owns the arena, the buffer pool and the conv layers of one model instance on one device."""

    def __init__(self, device: torch.device, conv_dtype: int = ops.F32, group: int = 1):
        self.device = device
        self.conv_dtype = conv_dtype   # ops.F32: exact fp32 MFMA; ops.BF16: bf16 operands / fp32 accumulate
        # `group` volumes adapt side by side as the batch items of every launch, each with its own parameter replica
        # (method.group; Arena.replicas, mmtta_param_sets).  `use_sets` is switched on by the adaptation plugin around its
        # launches; the nn.Module facade (plain batched forward, one weight set) leaves it off
        self.group = max(1, int(group))
        self.use_sets = False
        self.pool = Pool(device)
        # forward activations of more than 4 channels stored as bf16 (torch-autocast style): set by runtimes whose every
        # layer kind has storage-agnostic kernels (models/unet.py); gradients, logits, statistics, weights stay fp32
        self.act_bf16 = False
        # gradients of more than 4 channels stored as bf16 as well (method.grad_storage; only next to bf16-stored
        # activations): every consumer of such a gradient - input-gradient and weight-gradient convolutions - rounds it to
        # bf16 while staging it anyway; statistics, reductions, the reduced weight gradients and the optimizer stay fp32
        self.grad_bf16 = False
        self.training = False
        self.overlap_wgrad = True      # weight gradients on a side stream (joined before the optimizer)
        self.n_side = 2                # layers alternate between the side streams (a layer always uses the same one)
        self._side: List[torch.cuda.Stream] = []
        self.side_pending = False
        self.convs: List[ConvLayer] = []
        self.refs: List[ParamRef] = []
        self.buffers: List[torch.nn.Module] = []     # modules owning running statistics (BatchNorm)
        self.arena: Optional[Arena] = None

    def act_dtype(self, channels: int) -> torch.dtype:
        """Storage type of a forward activation with `channels` channels."""
        return torch.bfloat16 if (self.act_bf16 and channels > 4) else torch.float32

    def grad_dtype(self, channels: int, like: Optional[torch.Tensor] = None) -> torch.dtype:
        """Storage type of an activation GRADIENT with `channels` channels.  A thin gradient (<= 4 channels: the
        full-resolution tensors around the head) takes the storage of the gradient it is computed from (`like`): the loss
        writes bf16 where the runtime says so (`thin_grad_dtype`) and every thin gradient downstream follows."""
        if channels <= 4:
            return like.dtype if like is not None else torch.float32
        return torch.bfloat16 if (self.grad_bf16 and self.act_bf16) else torch.float32

    def thin_grad_dtype(self) -> torch.dtype:
        """Storage the loss should give d(logits) (<= 4 channels).  bf16 (8-byte voxels) where every consumer of a thin gradient
        has that form - models/unet.py in bf16 precision; its consumers round the values to bf16 while staging anyway."""
        return torch.float32

    # -- construction helpers
    def make_ref(self, name: str, param: torch.nn.Parameter) -> ParamRef:
        r = ParamRef(name, param)
        self.refs.append(r)
        return r

    def make_conv(self, name: str, module: torch.nn.Module, stride_hint: Optional[int] = None) -> ConvLayer:
        transposed = isinstance(module, torch.nn.ConvTranspose3d)
        k = module.kernel_size[0]
        s = module.stride[0]
        cin, cout = module.in_channels, module.out_channels
        op = ConvOp(cin, cout, k, s, transposed, self.device, dtype=self.conv_dtype, n_sets=self.group)
        w = self.make_ref(name + ".weight", module.weight)
        b = self.make_ref(name + ".bias", module.bias) if module.bias is not None else None
        layer = ConvLayer(op, w, b, self)
        layer.side_index = len(self.convs)
        self.convs.append(layer)
        return layer

    def make_conv_family(self, names: Sequence[str], modules: Sequence[torch.nn.Module], items_per_set: int = 1) -> ConvLayer:
        """ONE layer object for a family of identical convolution modules with different weights (``modules``: the same layer
        of the M modality encoders) - they run as consecutive batch items of one launch (item n uses member n % M of volume
        n // M).  ``items_per_set`` > 1 with a single module: consecutive batch items share that module's weights (the fusion
        convolution the reference applies M times)."""
        m0 = modules[0]
        transposed = isinstance(m0, torch.nn.ConvTranspose3d)
        k, s = m0.kernel_size[0], m0.stride[0]
        for mod in modules[1:]:
            if (type(mod), mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.bias is None) != \
                    (type(m0), m0.in_channels, m0.out_channels, m0.kernel_size, m0.stride, m0.bias is None):
                raise MmttaError("the members of a layer family must be identical modules")
        op = ConvOp(m0.in_channels, m0.out_channels, k, s, transposed, self.device, dtype=self.conv_dtype,
                    n_sets=self.group * len(modules))
        ws = [self.make_ref(n + ".weight", mod.weight) for n, mod in zip(names, modules)]
        bs = [self.make_ref(n + ".bias", mod.bias) if mod.bias is not None else None for n, mod in zip(names, modules)]
        layer = ConvLayer(op, ws[0], bs[0], self, members=list(zip(ws, bs)), items_per_set=items_per_set)
        layer.side_index = len(self.convs)
        self.convs.append(layer)
        return layer

    def side_stream(self, index: int = 0) -> Optional[torch.cuda.Stream]:
        if not self.overlap_wgrad or ops.PROFILER is not None:
            return None
        while len(self._side) < self.n_side:
            self._side.append(torch.cuda.Stream(device=self.device))
        return self._side[index % self.n_side]

    def join_side(self) -> None:
        """Main stream waits for the side streams' weight gradients (call before anything reads .grad)."""
        if self.side_pending:
            for s in self._side:
                torch.cuda.current_stream().wait_stream(s)
        self.side_pending = False

    def assign_groups(self, trainable: Optional[set], no_decay_keys: Sequence[str], treat_1d: bool) -> None:
        """decay / no-decay split of reference src/core/experiment_manager.py:214-228; parameters whose
        name is not in ``trainable`` (None = all) are frozen."""
        for r in self.refs:
            if trainable is not None and r.name not in trainable:
                r.group = GROUP_FROZEN
            elif any(k in r.name for k in no_decay_keys) or (treat_1d and len(r.shape) == 1):
                r.group = GROUP_NO_DECAY
            else:
                r.group = GROUP_DECAY

    def build_arena(self) -> Arena:
        if self.group > 1:
            # a volume group needs every parameter to be per-item: convolution weights and biases are (mmtta_param_sets);
            # norm affines and BatchNorm's cross-item statistics are not
            bad = [r.name for r in self.refs if ".adn.N." in r.name]
            if bad or self.buffers:
                raise NotImplementedError(
                    f"method.group = {self.group} needs per-volume norms without parameters (INSTANCE, the shipped configs); "
                    f"this model has {bad[:2] or 'BatchNorm running statistics'}: use method.lanes with method.group = 1")
        self.arena = Arena(self.refs, self.device, replicas=self.group)
        for c in self.convs:
            c.bind_sets(self.arena)
        for mod in self.buffers:                    # BatchNorm running statistics follow the parameters
            for name, buf in list(mod.named_buffers(recurse=False)):
                if buf is not None and buf.device != self.device:
                    setattr(mod, name, buf.to(self.device))
        return self.arena

    def pack_all(self) -> None:
        """Refresh every packed weight image from the arena (one launch)."""
        if getattr(self, "_packer", None) is None or self._packer_base != self.arena.params.data_ptr():
            items = []
            seen = set()
            for c in self.convs:
                if id(c.op) in seen:
                    continue
                seen.add(id(c.op))
                inner = len(c.members)
                for g in range(self.arena.replicas if c.op.n_sets == self.arena.replicas * inner else 1):
                    for m, (w, _) in enumerate(c.members):
                        wd = self.arena.replica_data(w, g)
                        items.append((c.op.d_fwd, wd, c.op.packed_image(False, g * inner + m)))
                        if c.op.need_dgrad:
                            items.append((c.op.d_dgrad, wd, c.op.packed_image(True, g * inner + m)))
            self._packer = ops.BatchedPacker(items, self.device)
            self._packer_base = self.arena.params.data_ptr()
        self._packer.run()

    def snapshot_buffers(self) -> None:
        """Source values of the running statistics (BatchNorm), restored with the weights per volume."""
        self._buf_src = [(mod, name, buf.clone()) for mod in self.buffers
                         for name, buf in mod.named_buffers(recurse=False) if buf is not None]

    def restore_buffers(self) -> None:
        for mod, name, src in getattr(self, "_buf_src", []):
            getattr(mod, name).copy_(src)

    # -- whole-network entry points (subclasses provide forward_cl / backward_cl)
    def stage_input(self, x: torch.Tensor) -> torch.Tensor:
        n, c, d, h, w = x.shape
        if c != self.in_channels:
            raise ValueError(f"model expects {self.in_channels} input channels, got {c}")
        x_cl = self.pool.cl("x", n, d, h, w, c, ldc=(c + 3) // 4 * 4, zero=True, dtype=self.input_dtype())
        ops.to_cl(x, out=x_cl)
        return x_cl

    def input_dtype(self) -> torch.dtype:
        """Storage of the staged network input.  Runtimes whose every reader of the input rounds it to bf16 while staging
        (models/unet.py in bf16 precision) store it as bf16: the same values from half the bytes."""
        return torch.float32

    def run_forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: NCDHW fp32 on this device -> logits as a channels-last view [N,D,H,W,R]."""
        self.pack_all()
        return self.forward_cl(self.stage_input(x))

    def run_backward(self, dlogits_cl: torch.Tensor) -> None:
        self.backward_cl(dlogits_cl)
        self.join_side()


# ----------------------------------------------------------------------------- container -> block builders
def build_norm(rt: Runtime, prefix: str, adn: Optional[torch.nn.Module], channels: int) -> Optional[NormLayer]:
    """ADN container -> NormLayer.  Dropout must be p=0 (every shipped config), activation ReLU."""
    if adn is None:
        return None
    mods = dict(adn.named_children())
    drop = mods.get("D")
    if drop is not None and float(drop.p) != 0.0:
        raise NotImplementedError("dropout p > 0 is not on the adaptation path (shipped configs use 0.0)")
    nmod = mods.get("N")
    if nmod is None:
        raise NotImplementedError("an ADN without a norm layer is not supported by the fused norm-on-load path")
    if "A" not in mods:
        raise NotImplementedError("an ADN without an activation is not supported")
    if isinstance(nmod, torch.nn.InstanceNorm3d):
        if nmod.track_running_stats:
            raise NotImplementedError("InstanceNorm3d(track_running_stats=True)")
        g = rt.make_ref(prefix + ".N.weight", nmod.weight) if nmod.affine else None
        b = rt.make_ref(prefix + ".N.bias", nmod.bias) if nmod.affine else None
        return NormLayer("INSTANCE", channels, 1, nmod.eps, 0.1, g, b)
    if isinstance(nmod, torch.nn.BatchNorm3d):
        if not (nmod.affine and nmod.track_running_stats):
            raise NotImplementedError("BatchNorm3d without affine / running statistics")
        g = rt.make_ref(prefix + ".N.weight", nmod.weight)
        b = rt.make_ref(prefix + ".N.bias", nmod.bias)
        rt.buffers.append(nmod)
        mom = 0.1 if nmod.momentum is None else float(nmod.momentum)
        return NormLayer("BATCH", channels, 1, nmod.eps, mom, g, b, bn_module=nmod)
    if isinstance(nmod, torch.nn.GroupNorm):
        g = rt.make_ref(prefix + ".N.weight", nmod.weight) if nmod.affine else None
        b = rt.make_ref(prefix + ".N.bias", nmod.bias) if nmod.affine else None
        return NormLayer("GROUP", channels, nmod.num_groups, nmod.eps, 0.1, g, b)
    raise NotImplementedError(f"norm module {type(nmod).__name__}")


def build_convolution(rt: Runtime, prefix: str, cont: torch.nn.Module) -> ConvolutionBlock:
    conv = rt.make_conv(prefix + ".conv", cont.conv)
    adn = getattr(cont, "adn", None)
    norm = build_norm(rt, prefix + ".adn", adn, cont.conv.out_channels)
    return ConvolutionBlock(rt, conv, norm)


def _family_norm(rt: Runtime, prefix: str, cont: torch.nn.Module) -> Optional[NormLayer]:
    """The norm of a layer family: one NormLayer serves every member, so it must carry no parameters and no cross-item
    statistics (InstanceNorm3d without affine: statistics per batch item and channel - the shipped configs)."""
    adn = getattr(cont, "adn", None)
    if adn is None:
        return None
    nmod = dict(adn.named_children()).get("N")
    if not isinstance(nmod, torch.nn.InstanceNorm3d) or nmod.affine or nmod.track_running_stats:
        raise NotImplementedError("a layer family (modality encoders in one launch) needs parameter-free per-item norms (INSTANCE)")
    return build_norm(rt, prefix + ".adn", adn, cont.conv.out_channels)


def build_convolution_family(rt: Runtime, prefixes: Sequence[str], conts: Sequence[torch.nn.Module],
                             items_per_set: int = 1) -> ConvolutionBlock:
    conv = rt.make_conv_family([p + ".conv" for p in prefixes], [c.conv for c in conts], items_per_set)
    return ConvolutionBlock(rt, conv, _family_norm(rt, prefixes[0], conts[0]))


def build_residual_unit_family(rt: Runtime, prefixes: Sequence[str], conts: Sequence[torch.nn.Module]) -> ResidualUnitBlock:
    """monai ResidualUnit for a family of identical units (same layer of every modality encoder)."""
    names = [name for name, _ in conts[0].conv.named_children()]
    units = [build_convolution_family(rt, [f"{p}.conv.{name}" for p in prefixes], [getattr(c.conv, name) for c in conts])
             for name in names]
    residual = None
    if isinstance(conts[0].residual, torch.nn.Conv3d):
        residual = rt.make_conv_family([p + ".residual" for p in prefixes], [c.residual for c in conts])
    return ResidualUnitBlock(rt, units, residual)


def build_residual_unit(rt: Runtime, prefix: str, cont: torch.nn.Module) -> ResidualUnitBlock:
    units = [build_convolution(rt, f"{prefix}.conv.{name}", unit) for name, unit in cont.conv.named_children()]
    residual = None
    if isinstance(cont.residual, torch.nn.Conv3d):
        residual = rt.make_conv(prefix + ".residual", cont.residual)
    return ResidualUnitBlock(rt, units, residual)
