"""Thin, typed Python wrappers over the C ABI (one function per entry point of include/mmtta.h).

Activations are channels-last views: torch tensors of logical shape [N, D, H, W, C] with unit
stride along C (possibly a channel slice of a wider buffer, which is how ``torch.cat`` of the
reference - src/models/unet_multimodal_midfusion.py:135, monai SkipConnection - disappears:
producers write straight into their slice).  Everything runs on torch's current CUDA stream, so
the calls can be captured into a graph.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (BF16, CONV_DGRAD, CONV_FWD, CONVT_DGRAD, CONVT_FWD, F32, NORM_BATCH, NORM_GROUP, NORM_INSTANCE,
                   ConvDesc, ConvEpilogue, ConvPlan, MmttaError, ParamSets, check, desc_cl, desc_ncdhw, ptr, stream_ptr)

NORM_KINDS = {"INSTANCE": NORM_INSTANCE, "BATCH": NORM_BATCH, "GROUP": NORM_GROUP}
PRECISIONS = {"fp32": F32, "f32": F32, "bf16": BF16}


def _require_cuda() -> None:
    if not torch.cuda.is_available():
        raise MmttaError("the adaptation path needs an MI355X (gfx950) device; no CPU fallback exists")


# ----------------------------------------------------------------------------- workspace
class _WorkspaceSelectors(type):
    """`Workspace.slot / lane / frozen` select the scratch buffer of the launches that follow; they are per THREAD (a second
    Python thread driving its own lane does not disturb another thread's selection), with the old class-attribute syntax."""
    _tls = threading.local()

    def _get(cls, name, default):
        return getattr(_WorkspaceSelectors._tls, name, default)

    slot = property(lambda cls: cls._get("slot", 0), lambda cls, v: setattr(_WorkspaceSelectors._tls, "slot", int(v)))
    lane = property(lambda cls: cls._get("lane", 0), lambda cls, v: setattr(_WorkspaceSelectors._tls, "lane", int(v)))
    frozen = property(lambda cls: cls._get("frozen", False), lambda cls, v: setattr(_WorkspaceSelectors._tls, "frozen", bool(v)))


class Workspace(metaclass=_WorkspaceSelectors):
    """One grow-only scratch buffer per device, shared by split-K convs and weight gradients
    (stream ordered).  It must reach its final size before a graph capture starts.

    A buffer that has to grow is REPLACED, never freed: hipGraphs captured for an earlier (smaller) input shape have
    the old address baked into their kernel nodes and stay replayable (the adaptation plugin keeps one graph per input
    shape); handing the old block back to the caching allocator would let those replays scribble over whatever tensor
    received it next."""

    _buffers: Dict[Tuple[int, int, int], torch.Tensor] = {}
    _retired: list = []     # superseded buffers, kept alive for the graphs that reference them
    captured: set = set()   # (device index, lane) pairs on which a hipGraph has been captured (the plugin records them)
    min_bytes = 1 << 20     # first allocation of a slot (tests lower it to provoke growth with small shapes)
    # frozen / slot / lane: thread-local selectors (metaclass above).  slot 0: main launch sequence, 1..: the side streams of
    # the weight gradients (engine.ConvLayer.wgrad); lane: runtimes that run concurrently on different streams keep separate scratch

    @classmethod
    def get(cls, nbytes: int, device: torch.device) -> torch.Tensor:
        """The scratch buffer of the current slot: kernels of the two streams run concurrently and must not share
        scratch."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, cls.slot, cls.lane)
        buf = cls._buffers.get(key)
        if buf is None or buf.numel() < nbytes:
            if cls.frozen:
                raise MmttaError("workspace would have to grow during graph capture; run one eager warm-up step first")
            if buf is not None and (idx, cls.lane) in cls.captured:
                cls._retired.append(buf)        # a captured graph may replay with the old address: keep the block alive
            # grow geometrically: a run over volumes of many sizes retires O(log) buffers, not one per new size; a buffer no
            # graph can reference is simply dropped (the allocator reuses it)
            grown = 0 if buf is None else buf.numel() + buf.numel() // 2
            buf = torch.empty(max(nbytes, grown, cls.min_bytes), dtype=torch.uint8, device=device)
            cls._buffers[key] = buf
        return buf


# ----------------------------------------------------------------------------- streams / hardware queues
_HELPER_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def helper_stream(device) -> "torch.cuda.Stream":
    """One reusable non-default stream per device (graph capture from the default stream)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _HELPER_STREAMS:
        _HELPER_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _HELPER_STREAMS[idx]


_QUEUE_WARNED = False


def hw_queue_status() -> dict:
    """How many hardware queues the HIP runtime of this process multiplexes its streams over, as far as the package can
    know: GPU_MAX_HW_QUEUES is read when HIP starts, so a host that initialised the GPU BEFORE importing this package runs
    with whatever the environment held then (the runtime's default is 4)."""
    import multimodal_tta_amd as pkg
    raw = pkg.HW_QUEUES_AT_HIP_START
    try:
        queues = int(raw) if raw is not None else 4
    except ValueError:
        queues = 4
    return {"queues": queues, "hip_started_before_import": bool(pkg.HIP_STARTED_BEFORE_IMPORT), "env_at_hip_start": raw}


def lanes_effective(count: int) -> int:
    """Lanes that can be expected to run on their own hardware queue: with the runtime's default of 4 queues the lanes beyond
    the second have been measured sharing a queue with an earlier stream (profiles/r02_hw_queues.txt)."""
    q = hw_queue_status()["queues"]
    return int(count) if q >= 8 else min(int(count), 2)


def lane_streams(count: int, device) -> list:
    """`count` streams for volumes adapted concurrently on one GPU, each bound to its OWN hardware queue.

    The HIP runtime gives a stream its hardware queue at the stream's first command and has GPU_MAX_HW_QUEUES of them
    (8: set by the package at import, read at HIP start-up); once they are used up, later streams share the queue of an
    existing one, and two lanes on one queue serialise (measured, 4 lanes: 36 vs 53 volumes/s for the same code,
    depending only on which streams had been touched first).  So the lanes' streams are created AND given their first
    command here, before any other stream of the process does work."""
    global _QUEUE_WARNED
    st = hw_queue_status()
    if int(count) > 2 and st["queues"] < 8 and not _QUEUE_WARNED:
        _QUEUE_WARNED = True
        import warnings
        warnings.warn(
            f"{count} lanes requested but the HIP runtime of this process started with GPU_MAX_HW_QUEUES="
            f"{st['env_at_hip_start'] or 'unset (4 queues)'}"
            + (" (the GPU was initialised before multimodal_tta_amd was imported, so the package could not set it)"
               if st["hip_started_before_import"] else "")
            + ": lanes beyond the second share a hardware queue and serialise - measured 42 instead of 50 volumes/s for four "
              "lanes (profiles/r02_hw_queues.txt).  Export GPU_MAX_HW_QUEUES=8 before the first CUDA call, or use "
              "method.group (volumes batched into one launch sequence) with method.lanes <= 2.", RuntimeWarning)
    device = torch.device(device)
    streams = [torch.cuda.Stream(device=device) for _ in range(int(count))]
    for s in streams:
        with torch.cuda.stream(s):
            torch.zeros(1, device=device)
    torch.cuda.synchronize(device)
    return streams


# ----------------------------------------------------------------------------- library options
_OPTION_EPOCH = 0


def set_option(key: int, value: int) -> int:
    """``mmtta_set_option`` + invalidation of every cached launch plan: the launch-geometry knobs (split-K thresholds,
    weight-gradient slab counts) change workspace sizes and the number of statistics rows a convolution writes, so a
    plan made under the old value must not size buffers for a launch made under the new one.  Set options BEFORE the
    first adaptation step of a plugin: graphs already captured keep the geometry they were captured with."""
    global _OPTION_EPOCH
    prev = int(_lib.load().mmtta_set_option(int(key), int(value)))
    _OPTION_EPOCH += 1
    return prev


# launch-geometry knobs (include/mmtta.h: per batch item) measured best for 4 volumes in flight on one GPU; the optimum
# scales inversely with the volumes in flight (lanes x group): one volume alone wants 384 / 512 / 512 / 1024 (round 1), sixteen
# want 24 / 32 / 32 / 64 (profiles/r03_tuning_sweep.txt) - the launches of all volumes together should fill the chip about once
TUNE_AT_4 = {2: 96, 3: 128, 4: 128, 5: 256, 12: 128}      # SPLITK_BELOW, SPLITK_TARGET, WGRAD_WORKGROUPS, WGRAD_THIN_SLABS, CLASS_FUSED_MIN_WORKGROUPS
if os.environ.get("MMTTA_SPLITK_AT4"):                    # measurement switch: "below,target" at 4 volumes in flight
    TUNE_AT_4[2], TUNE_AT_4[3] = (int(v) for v in os.environ["MMTTA_SPLITK_AT4"].split(","))
_TUNED_FOR: Optional[int] = None


def tune_for_volumes_in_flight(volumes: int) -> Dict[int, int]:
    """Set the four launch-geometry knobs for `volumes` volumes in flight on this GPU (method.lanes x method.group).  Process
    wide, like the knobs themselves; call before the first adaptation step (cached plans are dropped, captured graphs keep the
    geometry they were captured with).  MMTTA_NO_AUTOTUNE=1 leaves the knobs alone (sweeps set them by hand)."""
    global _TUNED_FOR
    volumes = max(1, int(volumes))
    vals = {k: max(1, (v * 4 + volumes - 1) // volumes) for k, v in TUNE_AT_4.items()}
    if os.environ.get("MMTTA_NO_AUTOTUNE", "0") == "1" or _TUNED_FOR == volumes:
        return vals
    for k, v in vals.items():
        set_option(k, v)
    _TUNED_FOR = volumes
    return vals


# ----------------------------------------------------------------------------- norm on load
@dataclass
class NL:
    """Pending normalisation(+ReLU) of a raw conv output; consumers apply it when they read."""
    mean: torch.Tensor
    rstd: torch.Tensor
    gamma: Optional[torch.Tensor] = None
    beta: Optional[torch.Tensor] = None
    relu: bool = True
    scale: Optional[torch.Tensor] = None     # precombined rstd*gamma and beta - mean*rstd*gamma (fast consumer path)
    shift: Optional[torch.Tensor] = None

    def struct(self) -> _lib.NormOnLoad:
        return _lib.norm_on_load(self.mean, self.rstd, self.gamma, self.beta, self.relu, self.scale, self.shift)


def _nl_ref(nl: Optional[NL]):
    if nl is None:
        return None, None
    s = nl.struct()
    return s, C.byref(s)


# ----------------------------------------------------------------------------- layout
def row_pad(c: int, dtype: torch.dtype = torch.float32) -> int:
    """Padded channel count of a voxel row: 16-byte rows for fp32 (4 channels), and for bf16 rows of more than 4 channels
    (8 channels: the implicit GEMM stages 8 channels per 16-byte load); a bf16 row of <= 4 channels is 8 bytes."""
    if dtype == torch.bfloat16 and c > 4:
        return (c + 7) // 8 * 8
    return (c + 3) // 4 * 4


def new_cl(n: int, d: int, h: int, w: int, c: int, device, ldc: Optional[int] = None, zero: bool = False,
           dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Allocate a channels-last buffer [n,d,h,w,ldc] and return the [.., :c] view."""
    ldc = c if ldc is None else ldc
    buf = (torch.zeros if zero else torch.empty)((n, d, h, w, ldc), dtype=dtype, device=device)
    if ldc == c:
        return buf
    view = buf[..., :c]
    view._mmtta_owns_pad = True     # this view owns the pad lanes of its rows (any further slice does not)
    return view


def to_cl(x: torch.Tensor, out: Optional[torch.Tensor] = None, ldc: Optional[int] = None) -> torch.Tensor:
    """NCDHW -> channels-last (reference tensor contract: src/datasets/brats.py:343-347)."""
    _require_cuda()
    n, c, d, h, w = x.shape
    if out is None:
        out = new_cl(n, d, h, w, c, x.device, ldc if ldc is not None else (c + 3) // 4 * 4, zero=True)
    src = desc_ncdhw(x)
    dst = desc_cl(out)
    check(_lib.load().mmtta_copy_strided(C.byref(src), C.byref(dst), stream_ptr()), "copy_strided")
    return out


def from_cl(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """channels-last -> contiguous NCDHW (what reference src/evaluation/seg_eval.py:300 expects back)."""
    n, d, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=x.device)
    src = desc_cl(x)
    # describe `out` with the logical (n,c,d,h,w) of src
    dst = desc_ncdhw(out)
    check(_lib.load().mmtta_copy_strided(C.byref(src), C.byref(dst), stream_ptr()), "copy_strided")
    return out


# ----------------------------------------------------------------------------- profiling hook
IGEMM_KERNELS = ["igemm_f32_kernel<1,4,8,8,8,8>", "igemm_f32_kernel<2,4,4,8,8,16>", "igemm_f32_kernel<4,4,4,4,8,32>",
                 "igemm_f32_kernel<1,1,4,4,8,8>", "igemm_f32_kernel<2,2,4,4,8,8>", "igemm_f32_kernel<4,4,4,4,8,8>",
                 "direct_conv_kernel",
                 "igemm_kernel<1,4,8,8,8,16,bf16>", "igemm_kernel<2,4,4,8,8,32,bf16>", "igemm_kernel<4,4,4,4,8,32,bf16>",
                 "igemm_kernel<1,1,4,4,8,16,bf16>", "igemm_kernel<2,2,4,4,8,16,bf16>", "igemm_kernel<4,4,4,4,8,16,bf16>",
                 "direct_chan_kernel", "igemm_kernel<1,2,4,8,8,16,bf16>", "igemm_cls8_kernel<16,bf16>"]


WGRAD_KERNELS = ["wgrad_f32_kernel<4,4,8,7>", "wgrad_f32_kernel<2,2,8,7>", "wgrad_f32_kernel<4,4,8,1>", "wgrad_small_kernel",
                 "wgrad_bf16_kernel<4,4,1>", "wgrad_bf16_kernel<2,4,2>", "wgrad_tiny_kernel",
                 "wgrad_tr_kernel<4,8,1>", "wgrad_tr_kernel<2,4,2>", "wgrad_tr1_kernel", "wgrad_thin_tr_kernel"]


class KernelProfiler:
    """Brackets every conv launch with events on the launch stream and books its algorithmic FLOPs
    (bench.py roofline: FLOPs per launch / measured duration).  Off unless installed in ops.PROFILER."""

    def __init__(self, reps: int = 1, what_if: Sequence[str] = ()):
        self.records = []   # (kernel name, launches, flops, start event, end event)
        self.what_if = tuple(what_if)       # "no_norm_on_load", "no_stats": see ConvOp._run
        # reps > 1: every profiled call is issued `reps` more times back to back between the two events, so the
        # queue stays ahead of the GPU and the host's launch latency does not leak into the measured duration
        # (outputs of accumulating calls are then wrong: use only in a throw-away pass)
        self.reps = max(1, int(reps))

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name: str, launches: int, flops: float, e0, detail: str = "", nbytes: float = 0.0) -> None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((name, launches, flops, e0, e1, detail, nbytes))

    def by_layer(self):
        """Per (kernel, layer shape) totals: scripts/layer_times.py prints them."""
        torch.cuda.synchronize()
        out = {}
        for name, launches, flops, e0, e1, detail, nbytes in self.records:
            d = out.setdefault((name, detail), {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            d["launches"] += launches
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, launches, flops, e0, e1, _detail, nbytes in self.records:
            d = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0, "calls": 0, "bytes": 0.0})
            d["launches"] += launches
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
        return out


PROFILER: Optional[KernelProfiler] = None


# ----------------------------------------------------------------------------- convolution
class ConvOp:
    """One Conv3d / ConvTranspose3d module's kernels: forward, input gradient, weight gradient.

    Holds the two packed weight images (forward and input-gradient orientation), refreshed by
    ``pack`` after every optimizer step.
    """

    def __init__(self, cin: int, cout: int, ksize: int, stride: int, transposed: bool, device, dtype: int = F32,
                 n_sets: int = 1):
        """``n_sets`` > 1: the op holds that many packed weight images back to back - one per parameter set of a group of
        volumes (x the members of a set family, e.g. the M modality encoders); which batch item reads which image is said by
        ``set_param_sets`` (mmtta_param_sets)."""
        _require_cuda()
        self.cin, self.cout, self.k, self.stride, self.transposed = cin, cout, ksize, stride, transposed
        self.device = torch.device(device)
        fwd_op, dg_op = (CONVT_FWD, CONVT_DGRAD) if transposed else (CONV_FWD, CONV_DGRAD)
        self.d_fwd = ConvDesc(fwd_op, ksize, stride, cin, cout, dtype)
        self.d_dgrad = ConvDesc(dg_op, ksize, stride, cin, cout, dtype)
        lib = _lib.load()
        nb_f = lib.mmtta_conv_packed_bytes(C.byref(self.d_fwd))
        nb_d = lib.mmtta_conv_packed_bytes(C.byref(self.d_dgrad))
        if nb_f < 0 or nb_d < 0:
            check(-2, "conv_packed_bytes")
        self.n_sets = max(1, int(n_sets))
        self.nb_fwd, self.nb_dgrad = int(nb_f), int(nb_d)        # bytes of ONE image (multiples of 16)
        # zero-filled once: the pack kernels write only the entries a weight owns (padding and, for the gather-GEMM
        # up-convolution image, the slots no tap maps to stay zero)
        self.packed_fwd = torch.zeros(nb_f * self.n_sets, dtype=torch.uint8, device=self.device)
        self.packed_dgrad = torch.zeros(nb_d * self.n_sets, dtype=torch.uint8, device=self.device)
        self._plans: Dict[Tuple, ConvPlan] = {}
        self.need_dgrad = True
        # per-item parameter sets (None: one set for the whole batch).  `sets_ctl.use_sets` (the owning runtime) switches
        # them on for the launches of a volume group and off for the plain batched forward of the nn.Module facade
        self.sets_grouped: Optional[Tuple[ParamSets, ParamSets]] = None      # (forward image, input-gradient image)
        self.sets_plain: Optional[Tuple[ParamSets, ParamSets]] = None        # a family of modules inside ONE weight set
        self.sets_ctl = None

    def set_param_sets(self, items_per_set: int, inner: int, weight_outer: int, weight_inner: int, bias_outer: int,
                       bias_inner: int, ctl) -> None:
        """Batch item n uses set q = n // items_per_set, stored (q // inner) outer + (q % inner) inner strides behind set 0;
        the packed images of the op are laid out [outer][inner]."""
        if self.n_sets % inner:
            raise MmttaError(f"{self.n_sets} packed images cannot hold families of {inner} sets")
        mk = lambda nb, outer: ParamSets(int(items_per_set), int(inner), nb * int(inner) * outer, nb, int(weight_outer) * outer,
                                         int(weight_inner), int(bias_outer) * outer, int(bias_inner))
        self.sets_grouped = (mk(self.nb_fwd, 1), mk(self.nb_dgrad, 1))
        # without a volume group (the nn.Module facade: one weight set for the whole batch) a family of modules - the M
        # modality encoders as the batch items of one launch - still selects its member's weights: outer strides 0
        self.sets_plain = (mk(self.nb_fwd, 0), mk(self.nb_dgrad, 0)) if inner > 1 else None
        self.sets_ctl = ctl

    def _sets(self, desc) -> Optional[ParamSets]:
        grouped = self.sets_ctl is not None and getattr(self.sets_ctl, "use_sets", False)
        pair = self.sets_grouped if grouped else self.sets_plain
        if pair is None:
            return None
        return pair[1] if int(desc.op) in (CONV_DGRAD, CONVT_DGRAD) else pair[0]

    def packed_image(self, dgrad: bool, index: int) -> torch.Tensor:
        """The packed image of parameter set ``index`` (a view)."""
        nb, buf = (self.nb_dgrad, self.packed_dgrad) if dgrad else (self.nb_fwd, self.packed_fwd)
        return buf[index * nb:(index + 1) * nb]

    def weight_shape(self) -> Tuple[int, ...]:
        k = self.k
        return (self.cin, self.cout, k, k, k) if self.transposed else (self.cout, self.cin, k, k, k)

    def pack(self, weight: torch.Tensor, index: int = 0) -> None:
        """Refresh the packed images of parameter set ``index`` from its master weight."""
        if tuple(weight.shape) != self.weight_shape() or weight.dtype != torch.float32 or not weight.is_contiguous():
            raise MmttaError(f"weight must be contiguous fp32 {self.weight_shape()}, got {tuple(weight.shape)}")
        lib = _lib.load()
        s = stream_ptr()
        check(lib.mmtta_conv_pack_weights(C.byref(self.d_fwd), ptr(weight), ptr(self.packed_image(False, index)), s), "pack fwd")
        if self.need_dgrad:
            check(lib.mmtta_conv_pack_weights(C.byref(self.d_dgrad), ptr(weight), ptr(self.packed_image(True, index)), s),
                  "pack dgrad")

    def out_shape(self, x: torch.Tensor) -> Tuple[int, int, int, int, int]:
        n, d, h, w, _ = x.shape
        if self.transposed:
            return (n, d * 2, h * 2, w * 2, self.cout)
        if self.stride == 1:
            return (n, d, h, w, self.cout)
        return (n, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2, self.cout)

    def plan(self, desc: ConvDesc, x: torch.Tensor, y: torch.Tensor) -> ConvPlan:
        # the planned geometry (tile form, statistics rows) also depends on storage type, strides and pointer alignment
        # (class-fused / lean / row-loader gates): all of it is part of the key
        key = (desc.op, tuple(x.shape), tuple(y.shape), x.dtype, y.dtype, x.stride(), y.stride(), x.data_ptr() % 16,
               y.data_ptr() % 16, _OPTION_EPOCH)
        p = self._plans.get(key)
        if p is None:
            p = ConvPlan()
            dx, dy = desc_cl(x), desc_cl(y)
            check(_lib.load().mmtta_conv_plan(C.byref(desc), C.byref(dx), C.byref(dy), C.byref(p)), "conv_plan")
            self._plans[key] = p
        return p

    def stats_rows(self, x: torch.Tensor, y: torch.Tensor) -> int:
        return int(self.plan(self.d_fwd, x, y).stats_rows)

    def _run(self, desc, packed, x, x_nl, bias, y, accumulate, stats, add, add_nl):
        p = self.plan(desc, x, y)
        ws = Workspace.get(int(p.workspace_bytes), x.device) if p.workspace_bytes > 0 else None
        dx, dy = desc_cl(x), desc_cl(y)
        nls, nlr = _nl_ref(x_nl)
        epi = None
        keep = None
        if add is not None:
            keep = desc_cl(add)
            epi = ConvEpilogue(C.pointer(keep), add_nl.struct() if add_nl is not None else _lib.norm_on_load())
        sets = self._sets(desc)
        def launch(nlr=nlr, stats=stats):
            check(
                _lib.load().mmtta_conv_run_sets(
                    C.byref(desc), C.byref(dx), nlr, ptr(packed), ptr(bias), C.byref(epi) if epi is not None else None,
                    C.byref(dy), 1 if accumulate else 0, ptr(stats), ptr(ws), int(p.workspace_bytes),
                    C.byref(sets) if sets is not None else None, stream_ptr()),
                "conv_run")

        launch()
        if PROFILER is not None:
            # what-if timings (scripts/layer_times.py --what-if): the TIMED repeats alone drop the norm-on-load of the input
            # or the statistics rows of the output; the real launch above already produced the step's values
            kw = {}
            if "no_norm_on_load" in PROFILER.what_if:
                kw["nlr"] = None
            if "no_stats" in PROFILER.what_if:
                kw["stats"] = None
            e0 = PROFILER.begin()
            for _ in range(PROFILER.reps):
                launch(**kw)
            PROFILER.end(self._kernel_name(p.config, desc, x, y), PROFILER.reps,
                         self.flops(x, y, desc) * PROFILER.reps, e0, self._detail(desc.op, x),
                         self.io_bytes(x, y, packed) * PROFILER.reps)

    def _kernel_name(self, config: int, desc, x: torch.Tensor, y: torch.Tensor) -> str:
        """Profiler label of the kernel a conv_run call dispatches to (mirrors csrc: the thin layers have matrix-core
        variants in bf16 mode; 'bf16' in the label selects the bf16 MFMA peak in bench.py)."""
        name = IGEMM_KERNELS[config]
        if int(desc.dtype) == BF16:
            if config == 13 and (self.cin_of(desc) >= 2 or self.cout_of(desc) > 32):
                return "chan_mfma_kernel<bf16>"
            if config == 6 and int(desc.op) == CONVT_FWD and self.k == 3 and self.stride == 2 and self.cin in (32, 64):
                return "upconv8_kernel<bf16>"
        return name

    def cin_of(self, desc) -> int:
        """Channels the op READS (dgrad ops run the layer backwards)."""
        return self.cout if int(desc.op) in (CONV_DGRAD, CONVT_DGRAD) else self.cin

    def cout_of(self, desc) -> int:
        return self.cin if int(desc.op) in (CONV_DGRAD, CONVT_DGRAD) else self.cout

    def _detail(self, op: int, x: torch.Tensor) -> str:
        kind = {CONV_FWD: "fwd", CONV_DGRAD: "dgrad", CONVT_FWD: "fwdT", CONVT_DGRAD: "dgradT", -1: "wgrad"}[op]
        return (f"{kind} {'convT' if self.transposed else 'conv'} {self.cin}->{self.cout} k{self.k}s{self.stride} "
                f"read {x.shape[1]}x{x.shape[2]}x{x.shape[3]}")

    def flops(self, x: torch.Tensor, y: torch.Tensor, desc=None) -> float:
        """Algorithmic FLOPs (2 x MACs) of one forward / input-gradient / weight-gradient of this module for
        the tensor pair (read, produced): MACs = coarse-grid voxels x Cin x Cout x taps."""
        coarse = min(x.shape[1] * x.shape[2] * x.shape[3], y.shape[1] * y.shape[2] * y.shape[3]) * x.shape[0]
        return 2.0 * coarse * self.cin * self.cout * self.k ** 3

    @staticmethod
    def io_bytes(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor) -> float:
        """Algorithmic bytes of one call (SURVEY.md section 8d convention): each activation tensor of the call once
        (logical channels, storage element size) plus the weight image / weight gradient once."""
        return float(a.numel() * a.element_size() + b.numel() * b.element_size() + w.numel() * w.element_size())

    def forward(self, x: torch.Tensor, x_nl: Optional[NL], bias: Optional[torch.Tensor], y: torch.Tensor,
                stats: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
                add_nl: Optional[NL] = None, accumulate: bool = False) -> None:
        self._run(self.d_fwd, self.packed_fwd, x, x_nl, bias, y, accumulate, stats, add, add_nl)

    def dgrad(self, dy: torch.Tensor, dx: torch.Tensor, accumulate: bool = False, add: Optional[torch.Tensor] = None) -> None:
        """dx (+)= conv^T(dy) [+ add]: ``add`` is a second gradient term of the same tensor (the identity branch of a
        ResidualUnit) summed in the epilogue instead of by a separate pass over dx."""
        self._run(self.d_dgrad, self.packed_dgrad, dy, None, None, dx, accumulate, None, add, None)

    def wgrad(self, x: torch.Tensor, x_nl: Optional[NL], dy: torch.Tensor, dw: torch.Tensor,
              db: Optional[torch.Tensor], accumulate: bool = False) -> None:
        lib = _lib.load()
        tx, tdy = desc_cl(x), desc_cl(dy)
        sets = self._sets(self.d_fwd)
        sref = C.byref(sets) if sets is not None else None
        need = lib.mmtta_conv_wgrad_workspace_bytes_sets(C.byref(self.d_fwd), C.byref(tx), C.byref(tdy), sref)
        if need < 0:
            check(-1, "conv_wgrad_workspace_bytes")
        ws = Workspace.get(int(need), x.device)
        nls, nlr = _nl_ref(x_nl)
        def launch():
            check(lib.mmtta_conv_wgrad_sets(C.byref(self.d_fwd), C.byref(tx), nlr, C.byref(tdy), ptr(dw), ptr(db),
                                            1 if accumulate else 0, ptr(ws), int(need), sref, stream_ptr()), "conv_wgrad")

        launch()
        if PROFILER is not None:
            e0 = PROFILER.begin()
            for _ in range(PROFILER.reps):
                launch()
            kid = lib.mmtta_conv_wgrad_kernel(C.byref(self.d_fwd), C.byref(tx), C.byref(tdy))
            nsets = x.shape[0] // int(sets.items_per_set) if sets is not None else 1
            PROFILER.end(WGRAD_KERNELS[kid], PROFILER.reps, self.flops(x, dy) * PROFILER.reps, e0, self._detail(-1, x),
                         (self.io_bytes(x, dy, dw) + (nsets - 1) * dw.numel() * 4.0) * PROFILER.reps)



class BatchedPacker:
    """Every packed weight image of a model refreshed by ONE kernel launch (the table is built once: parameter
    and image addresses are stable while the arena lives)."""

    def __init__(self, items, device):
        """items: list of (ConvDesc, master weight tensor, packed image tensor)."""
        lib = _lib.load()
        n = len(items)
        arr = (_lib.PackItem * n)()
        for i, (desc, w, packed) in enumerate(items):
            arr[i].desc = desc
            arr[i].w_master = ptr(w)
            arr[i].packed = ptr(packed)
        nbytes = int(lib.mmtta_conv_pack_table_bytes(n))
        host = (C.c_uint8 * nbytes)()
        total = C.c_int64(0)
        check(lib.mmtta_conv_pack_table_build(arr, n, host, C.byref(total)), "conv_pack_table_build")
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        self.count, self.total = n, int(total.value)
        self._keep = [t for it in items for t in it[1:]]

    def run(self) -> None:
        check(_lib.load().mmtta_conv_pack_batched(ptr(self.table), self.count, self.total, stream_ptr()),
              "conv_pack_batched")


# ----------------------------------------------------------------------------- normalisation
def reduce_rows_per_n(x: torch.Tensor) -> int:
    t = desc_cl(x)
    return int(_lib.load().mmtta_reduce_rows_per_n(C.byref(t)))


def channel_stats(x: torch.Tensor, part: torch.Tensor) -> None:
    t = desc_cl(x)
    check(_lib.load().mmtta_channel_stats(C.byref(t), ptr(part), stream_ptr()), "channel_stats")


def norm_stats_finalize(kind: int, groups: int, part: Optional[torch.Tensor], rows_per_n: int, n: int, c: int,
                        count: int, eps: float, training: bool, running_mean, running_var, momentum: float,
                        mean: torch.Tensor, rstd: torch.Tensor, scratch: torch.Tensor, gamma=None, beta=None,
                        scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None) -> None:
    check(_lib.load().mmtta_norm_stats_finalize(kind, groups, ptr(part), rows_per_n, n, c, count, eps,
                                                1 if training else 0, ptr(running_mean), ptr(running_var),
                                                momentum, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(scale),
                                                ptr(shift), ptr(scratch), stream_ptr()),
          "norm_stats_finalize")


def combine(a: torch.Tensor, a_nl: Optional[NL], b: Optional[torch.Tensor], b_nl: Optional[NL], out: torch.Tensor) -> None:
    ta, to = desc_cl(a), desc_cl(out)
    tb = desc_cl(b) if b is not None else None
    sa, ra = _nl_ref(a_nl)
    sb, rb = _nl_ref(b_nl)
    check(_lib.load().mmtta_combine(C.byref(ta), ra, C.byref(tb) if tb is not None else None, rb, C.byref(to),
                                    stream_ptr()), "combine")


def norm_bwd_reduce(dout: torch.Tensor, y: torch.Tensor, nl: NL, part: torch.Tensor) -> None:
    td, ty = desc_cl(dout), desc_cl(y)
    s, r = _nl_ref(nl)
    check(_lib.load().mmtta_norm_bwd_reduce(C.byref(td), C.byref(ty), r, ptr(part), stream_ptr()), "norm_bwd_reduce")


def norm_bwd_finalize(kind: int, groups: int, part: torch.Tensor, rows_per_n: int, n: int, c: int, count: int,
                      gamma, training: bool, m1: torch.Tensor, m2: torch.Tensor, dgamma, dbeta, accumulate: bool,
                      scratch: torch.Tensor) -> None:
    check(_lib.load().mmtta_norm_bwd_finalize(kind, groups, ptr(part), rows_per_n, n, c, count, ptr(gamma),
                                              1 if training else 0, ptr(m1), ptr(m2), ptr(dgamma), ptr(dbeta),
                                              1 if accumulate else 0, ptr(scratch), stream_ptr()),
          "norm_bwd_finalize")


def norm_bwd_apply(dout: torch.Tensor, y: torch.Tensor, nl: NL, m1: torch.Tensor, m2: torch.Tensor,
                   dy: torch.Tensor) -> None:
    td, ty, to = desc_cl(dout), desc_cl(y), desc_cl(dy)
    s, r = _nl_ref(nl)
    check(_lib.load().mmtta_norm_bwd_apply(C.byref(td), C.byref(ty), r, ptr(m1), ptr(m2), C.byref(to), stream_ptr()),
          "norm_bwd_apply")


def small_norm_backward_max() -> int:
    """Largest voxel count per batch item that takes the one-launch norm backward (A/B aid: MMTTA_NORM_SMALL=<voxels>, 0 =
    three passes at every level)."""
    return int(os.environ.get("MMTTA_NORM_SMALL", "512"))


def norm_bwd_small_ok(dout: torch.Tensor, y: torch.Tensor, nl: NL, dy: torch.Tensor) -> bool:
    """True when mmtta_norm_bwd_small (instance-norm backward of a small tensor in one launch) takes these tensors."""
    td, ty, to = desc_cl(dout), desc_cl(y), desc_cl(dy)
    s, r = _nl_ref(nl)
    return bool(_lib.load().mmtta_norm_bwd_small_ok(C.byref(td), C.byref(ty), r, C.byref(to)))


def norm_bwd_small(dout: torch.Tensor, y: torch.Tensor, nl: NL, count: int, dy: torch.Tensor) -> None:
    td, ty, to = desc_cl(dout), desc_cl(y), desc_cl(dy)
    s, r = _nl_ref(nl)
    check(_lib.load().mmtta_norm_bwd_small(C.byref(td), C.byref(ty), r, int(count), C.byref(to), stream_ptr()),
          "norm_bwd_small")


# ----------------------------------------------------------------------------- resample / glue
def upsample2x_fwd(x: torch.Tensor, y: torch.Tensor) -> None:
    tx, ty = desc_cl(x), desc_cl(y)
    check(_lib.load().mmtta_upsample2x_fwd(C.byref(tx), C.byref(ty), stream_ptr()), "upsample2x_fwd")


def upsample2x_bwd(dy: torch.Tensor, dx: torch.Tensor, accumulate: bool = False) -> None:
    ty, tx = desc_cl(dy), desc_cl(dx)
    check(_lib.load().mmtta_upsample2x_bwd(C.byref(ty), C.byref(tx), 1 if accumulate else 0, stream_ptr()),
          "upsample2x_bwd")


def lincomb(inputs: Sequence[torch.Tensor], weights: Sequence[float], out: torch.Tensor, accumulate: bool = False) -> None:
    k = len(inputs)
    descs = [desc_cl(t) for t in inputs]
    arr = (C.POINTER(_lib.Tensor) * k)(*[C.pointer(d) for d in descs])
    w = (C.c_float * k)(*[float(x) for x in weights])
    to = desc_cl(out)
    check(_lib.load().mmtta_lincomb(k, arr, w, C.byref(to), 1 if accumulate else 0, stream_ptr()), "lincomb")


# ----------------------------------------------------------------------------- loss / optimizer / metric
def entropy_partials(logits: torch.Tensor) -> int:
    t = desc_cl(logits)
    return int(_lib.load().mmtta_entropy_partials(C.byref(t)))


def entropy_loss(logits: torch.Tensor, dlogits: torch.Tensor, partial: torch.Tensor, loss: torch.Tensor,
                 softmax: bool = False) -> None:
    tz, tg = desc_cl(logits), desc_cl(dlogits)
    check(_lib.load().mmtta_entropy_loss(C.byref(tz), 1 if softmax else 0, C.byref(tg), ptr(partial), ptr(loss),
                                         stream_ptr()), "entropy_loss")


def entropy_partials_items(logits: torch.Tensor) -> int:
    t = desc_cl(logits)
    return int(_lib.load().mmtta_entropy_partials_items(C.byref(t)))


def entropy_loss_items(logits: torch.Tensor, dlogits: torch.Tensor, partial: torch.Tensor, loss: torch.Tensor,
                       softmax: bool = False) -> None:
    """The objective of every batch item on its own (N independent volumes in one launch): loss [N]."""
    if loss.numel() < logits.shape[0]:
        raise MmttaError("entropy_loss_items: one loss slot per batch item")
    tz, tg = desc_cl(logits), desc_cl(dlogits)
    check(_lib.load().mmtta_entropy_loss_items(C.byref(tz), 1 if softmax else 0, C.byref(tg), ptr(partial), ptr(loss),
                                               stream_ptr()), "entropy_loss_items")


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, n_decay: int, lr: float,
              beta1: float, beta2: float, eps: float, weight_decay: float, step: torch.Tensor) -> None:
    n = p.numel()
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise MmttaError("adam: p, g, m, v must be contiguous fp32 of equal length")
    if step.dtype != torch.int32:
        raise MmttaError("adam: step must be a device int32 scalar")
    check(_lib.load().mmtta_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), n, int(n_decay), lr, beta1, beta2, eps,
                                      weight_decay, ptr(step), stream_ptr()), "adam_step")


OPTIMIZERS = {"adam": _lib.OPTIM_ADAM, "adamw": _lib.OPTIM_ADAMW, "sgd": _lib.OPTIM_SGD}


@dataclass
class OptimSpec:
    """Hyper-parameters of the fused arena optimizer, read the way the reference's factory reads them (reference
    src/core/experiment_manager.py:199-237: `training.optimizer` names the class, `training.optimizers.<name>` holds
    its keyword arguments, `training.learning_rate / weight_decay / momentum` are the fall-backs)."""
    name: str = "adam"
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    momentum: float = 0.0
    dampening: float = 0.0
    nesterov: bool = False

    def struct(self) -> _lib.OptimDesc:
        return _lib.OptimDesc(OPTIMIZERS[self.name], self.lr, self.beta1, self.beta2, self.eps, self.weight_decay,
                              self.momentum, self.dampening, 1 if self.nesterov else 0)


def optim_step(spec: OptimSpec, p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: Optional[torch.Tensor],
               n_decay: int, step: torch.Tensor) -> None:
    """One step of torch.optim.{Adam, AdamW, SGD} over the flat arena ([0, n_decay) decays)."""
    n = p.numel()
    for t in (p, g, m) + ((v,) if v is not None else ()):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise MmttaError("optimizer: p, g, m, v must be contiguous fp32 of equal length")
    if step.dtype != torch.int32:
        raise MmttaError("optimizer: step must be a device int32 scalar")
    d = spec.struct()
    check(_lib.load().mmtta_optim_step(C.byref(d), ptr(p), ptr(g), ptr(m), ptr(v), n, int(n_decay), ptr(step),
                                       stream_ptr()), "optim_step")


def optim_step_sets(spec: OptimSpec, p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: Optional[torch.Tensor], n: int,
                    n_decay: int, sets: int, step: torch.Tensor) -> None:
    """One optimizer step over the first ``sets`` replicas of the arena ([replica][total], the first ``n`` elements of a
    replica train, its first ``n_decay`` decay) in one launch; one shared step counter."""
    for t in (p, g, m) + ((v,) if v is not None else ()):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.shape != p.shape:
            raise MmttaError("optimizer: p, g, m, v must be contiguous fp32 [replicas, total] of equal shape")
    if not (1 <= sets <= p.shape[0] and 0 <= n_decay <= n <= p.shape[1]):
        raise MmttaError(f"optimizer: {sets} sets of {n} ({n_decay} decaying) elements do not fit {tuple(p.shape)}")
    if step.dtype != torch.int32:
        raise MmttaError("optimizer: step must be a device int32 scalar")
    d = spec.struct()
    check(_lib.load().mmtta_optim_step_sets(C.byref(d), ptr(p), ptr(g), ptr(m), ptr(v), int(n), int(n_decay), int(sets),
                                            int(p.shape[1]), ptr(step), stream_ptr()), "optim_step_sets")


def _desc_any(t: torch.Tensor, channels_last: bool):
    return desc_cl(t) if channels_last else desc_ncdhw(t)


def mask_dice_counts(logits: torch.Tensor, label_ncdhw: torch.Tensor, threshold: float, counts: torch.Tensor,
                     mask: Optional[torch.Tensor] = None, logits_channels_last: bool = True) -> None:
    tz = _desc_any(logits, logits_channels_last)
    tl = desc_ncdhw(label_ncdhw)
    check(_lib.load().mmtta_mask_dice_counts(C.byref(tz), C.byref(tl), float(threshold), ptr(counts), ptr(mask),
                                             stream_ptr()), "mask_dice_counts")


_DCE_SCRATCH: Dict[Tuple, torch.Tensor] = {}     # block partials of dice_ce_sums, per (device, size)


def dice_ce_sums(logits: torch.Tensor, label_ncdhw: torch.Tensor, weight: Optional[torch.Tensor], squared_pred: bool,
                 out: torch.Tensor, logits_channels_last: bool = True, softmax: bool = False) -> None:
    tz = _desc_any(logits, logits_channels_last)
    tl = desc_ncdhw(label_ncdhw)
    lib = _lib.load()
    nbytes = int(lib.mmtta_dice_ce_scratch_bytes(C.byref(tz)))
    if nbytes < 0:
        check(-2, "dice_ce_scratch_bytes")
    key = (logits.device.index, nbytes, int(torch.cuda.current_stream().cuda_stream))   # per stream: lanes run concurrently
    scratch = _DCE_SCRATCH.get(key)
    if scratch is None:
        scratch = _DCE_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=logits.device)
    check(lib.mmtta_dice_ce_sums(C.byref(tz), C.byref(tl), ptr(weight), 1 if squared_pred else 0, 1 if softmax else 0, ptr(out),
                                 ptr(scratch), stream_ptr()), "dice_ce_sums")


def dice_ce_grad(logits: torch.Tensor, label_ncdhw: torch.Tensor, weight: Optional[torch.Tensor], squared_pred: bool,
                 jaccard: bool, include_background: bool, lambda_dice: float, lambda_ce: float, sums: torch.Tensor,
                 dlogits: torch.Tensor, logits_channels_last: bool = True, smooth_nr: float = 1e-5,
                 smooth_dr: float = 1e-5, softmax: bool = False) -> None:
    """d DiceCE / d logits from the device-resident sums of ``dice_ce_sums`` (same layouts as there)."""
    tz = _desc_any(logits, logits_channels_last)
    tg = _desc_any(dlogits, logits_channels_last)
    tl = desc_ncdhw(label_ncdhw)
    check(_lib.load().mmtta_dice_ce_grad(C.byref(tz), C.byref(tl), ptr(weight), 1 if squared_pred else 0, 1 if softmax else 0,
                                         1 if jaccard else 0, 1 if include_background else 0, float(lambda_dice),
                                         float(lambda_ce), float(smooth_nr), float(smooth_dr), ptr(sums), C.byref(tg),
                                         stream_ptr()), "dice_ce_grad")


_SURF_SCRATCH: Dict[Tuple, torch.Tensor] = {}    # working set of surface_distances, per (device, size)


def surface_distances(pred_mask: torch.Tensor, label_ncdhw: torch.Tensor, spacing: Sequence[float],
                      percentile: float = 95.0, asd_symmetric: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """pred_mask uint8 [B,R,D,H,W] (dense; from ``mask_dice_counts``) + labels -> (hd, asd) fp32 [B,R] on the device,
    as MONAI returns them (reference src/evaluation/seg_eval.py:327-341); the evaluator applies the fix-ups."""
    if pred_mask.dtype != torch.uint8 or not pred_mask.is_contiguous() or pred_mask.shape != label_ncdhw.shape:
        raise MmttaError(f"surface_distances: mask must be dense uint8 of the label's shape, got {pred_mask.dtype} "
                         f"{tuple(pred_mask.shape)} vs {tuple(label_ncdhw.shape)}")
    B, R, D, H, W = (int(v) for v in pred_mask.shape)
    lib = _lib.load()
    nbytes = int(lib.mmtta_surface_scratch_bytes(B * R, D, H, W))
    if nbytes < 0:
        raise MmttaError(f"surface_distances: extent {(D, H, W)} unsupported (at most 1024 per axis)")
    key = (pred_mask.device.index, nbytes, int(torch.cuda.current_stream().cuda_stream))  # per stream: lanes run concurrently
    scratch = _SURF_SCRATCH.get(key)
    if scratch is None:
        for k in [k for k in _SURF_SCRATCH if k[2] == key[2]]:      # a new shape on this stream replaces its old working set
            del _SURF_SCRATCH[k]
        scratch = _SURF_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=pred_mask.device)
    hd = torch.empty((B, R), dtype=torch.float32, device=pred_mask.device)
    asd = torch.empty((B, R), dtype=torch.float32, device=pred_mask.device)
    sp = (C.c_double * 3)(*[float(v) for v in spacing])
    tl = desc_ncdhw(label_ncdhw)
    check(lib.mmtta_surface_distances(ptr(pred_mask), C.byref(tl), sp, float(percentile), 1 if asd_symmetric else 0,
                                      ptr(hd), ptr(asd), ptr(scratch), stream_ptr()), "surface_distances")
    return hd, asd
