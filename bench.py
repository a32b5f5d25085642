#!/usr/bin/env python
"""Benchmark of the per-test-volume adaptation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = ONE ADAPTED VOLUME of BASELINE.json's headline configuration (configs[1]): a synthetic
BraTS-shaped 4x128^3 fp32 volume, S=10 entropy-minimisation steps of the registered `unet`
(forward + backward of every layer + fused Adam over all 19.22 M parameters), one final forward and
the mask/Dice tail.  Inputs are resident in HBM before the timed region.  Ranks adapt their own
volumes (weak scaling); the only collective is the all_gather of the per-volume Dice table.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     - the dominant conv kernel (by measured time; events on the launch stream).  bf16 mode is HBM-bound
                 (model AI ~ 109 flop/B < ridge ~ 310): `achieved` = that kernel's ALGORITHMIC
                 bytes per launch (input + weight image + output, each once) / its measured launch duration against
                 8 TB/s, `traffic` = its HBM bytes per launch from the committed rocprofv3 PMC pass of this build;
                 `mfma` carries the same kernel's FLOP/s against the dense MFMA peak of its operand type and
                 `whole_volume` the algorithmic bytes and FLOPs of a whole adapted volume over its wall time.
                 fp32 mode (--precision fp32) is MFMA-bound: the two blocks swap places.
  cpu_baseline - the oracle (torch CPU restatement of the reference path) timed on this host on a
                 bounded sample of the same workload.
  parity_full_size - the GPU path against that same CPU step on volume 0 (loss, logits, masks, Dice).
  variants     - (1 GPU) the same workload with one volume in flight, and in fp32 mode (the reference's arithmetic).
  ranks        - per-rank volumes/s (min / max over ranks) and the time of the gather.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The lanes of one GPU (volumes adapted concurrently, one stream each) only overlap when their streams sit on DIFFERENT
# hardware queues; the HIP runtime multiplexes all streams of a process over GPU_MAX_HW_QUEUES (default 4) queues.
# Measured (unet 4x128^3, S=10): 4 lanes 42.0 volumes/s with 4 queues, 50.1 with 8, 36.1 with 16.  Read at HIP start-up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA": ~2.5 PF dense (not the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0


def mfma_peak(kernel_name: str) -> float:
    """Dense MFMA peak of the operand type the named kernel feeds the matrix cores with."""
    bf = "bf16" in kernel_name or kernel_name.startswith(("wgrad_tr", "wgrad_thin_tr"))     # the transposed-read kernels are bf16-only
    return PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48, help="timed adapted volumes per rank (default: two rounds of 3 lanes x 8)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="unet", choices=["unet", "unet_multimodal_deepfusion"])
    ap.add_argument("--task", default="brats", choices=["brats", "hecktor21"])
    ap.add_argument("--method", default="tta_entmin", choices=["tta_entmin", "tta_moddrop"],
                    help="tta_moddrop = BASELINE configs[4]: a modality missing for good (t1c) + seeded modality dropout per step "
                         "(configs/method/tta_moddrop.yaml); its line carries no cpu_baseline / parity block (the oracle "
                         "comparison of that configuration is tests/test_hip_fullsize.py::test_config5_*)")
    ap.add_argument("--tta-steps", type=int, default=10)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="D H W (default: 128^3 brats, 48x144x144 hecktor)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="bf16 (BASELINE configs[1]): bf16 MFMA operands, fp32 accumulate/storage; fp32: exact fp32 MFMA")
    ap.add_argument("--lanes", type=int, default=None,
                    help="volumes adapted concurrently per GPU, each with its own weights, buffers, graph and stream "
                         "(episodic adaptation has no cross-volume state; one volume alone leaves most CUs waiting)")
    ap.add_argument("--group", type=int, default=None,
                    help="volumes adapted TOGETHER as the batch items of one launch sequence, each on its own replica of the "
                         "weights and optimizer state (method.group); lanes x group volumes are in flight per GPU")
    ap.add_argument("--tune-volumes", type=int, default=None,
                    help="launch geometry for this many volumes in flight (method.tune_volumes; default lanes x group) - e.g. a "
                         "one-lane counter run under the headline's geometry")
    ap.add_argument("--side-streams", type=int, default=None, help="side streams for the weight gradients (default: config)")
    ap.add_argument("--grad-storage", default=None, choices=["bf16", "fp32"],
                    help="bf16 precision: storage of the wide activation GRADIENTS (default: the method config, bf16)")
    ap.add_argument("--storage", default=None, choices=["bf16", "fp32"],
                    help="bf16 precision: storage of the wide forward activations (default: the method config, bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the one-lane and fp32-mode runs of the same workload")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="measurement aid, never the headline: the volumes start in pinned HOST memory and every group's "
                         "host-to-device copy is inside the timed region (on the lane's stream): the PCIe-inclusive rate")
    return ap.parse_args()


def build_cfg(args):
    from multimodal_tta_amd.config import compose
    ov = [f"task={args.task}", f"dataset={args.task}", f"model={args.model}", f"method={getattr(args, 'method', 'tta_entmin')}",
          f"method.steps={args.tta_steps}", f"method.precision={args.precision}"]
    if args.task == "hecktor21" and args.model != "unet":
        ov += ["model.num_modalities=2", "model.num_classes=1"]
    if args.no_graph:
        ov += ["method.use_graph=false"]
    if getattr(args, "side_streams", None) is not None:
        ov += [f"method.side_streams={args.side_streams}"]
    if getattr(args, "group", None) is not None:
        ov += [f"method.group={args.group}"]
    if getattr(args, "lanes", None) is not None:
        ov += [f"method.lanes={args.lanes}"]
    if getattr(args, "tune_volumes", None) is not None:
        ov += [f"method.tune_volumes={args.tune_volumes}"]
    if getattr(args, "grad_storage", None):
        ov += [f"method.grad_storage={args.grad_storage}"]
    if getattr(args, "storage", None) or os.environ.get("MMTTA_STORAGE"):
        ov += [f"method.storage={getattr(args, 'storage', None) or os.environ['MMTTA_STORAGE']}"]
    cfg = compose(overrides=ov)
    shape = args.shape or ([128, 128, 128] if args.task == "brats" else [48, 144, 144])
    cfg["dataset"]["synthetic"]["shape"] = list(shape)
    return cfg, tuple(shape)


def cpu_baseline(cfg, shape, tta_steps, timed_steps=2):
    """Oracle on the host cores, a bounded sample of the same workload (a whole 10-step volume is ~1 min of CPU work):
    one DISCARDED warm-up step (first call of the process: allocator, oneDNN primitive caches), then `timed_steps`
    adaptation steps timed per phase (forward / backward / optimizer) and one timed final forward + masks + Dice,
    extrapolated to S steps.  Also returns what the FIRST step produced (loss before the update, logits after it) so that
    the GPU path can be checked against it at full size."""
    import oracle
    from multimodal_tta_amd.synth import synth_volume

    torch.manual_seed(42)
    model = oracle.MODELS[cfg["model"]["name"]](cfg["model"])
    state = {k: v.clone() for k, v in model.state_dict().items()}
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    R = int(cfg["model"]["num_classes"])
    thr = float(cfg["evaluation"]["seg"]["threshold"])
    v = synth_volume(0, C, shape, R)
    x = v["image"].unsqueeze(0)
    opt = oracle.build_optimizer(list(model.named_parameters()), cfg["training"])

    def step():
        model.train()
        t = [time.perf_counter()]
        opt.zero_grad()
        loss = oracle.entropy_loss(model(x))
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        return float(loss.item()), (t[1] - t[0], t[2] - t[1], t[3] - t[2])

    def evaluate():
        model.eval()
        t0 = time.perf_counter()
        with torch.no_grad():
            z = model(x)
            pred, gt = oracle.masks_from_logits(z, v["label"].unsqueeze(0), thr)
            dice, _, _ = oracle.binary_dice_iou(pred, gt)
        return z, pred, dice, time.perf_counter() - t0

    loss0, warm = step()                               # discarded timing; its results are the parity reference
    z, pred, dice, _ = evaluate()                      # logits after exactly one step (also warms the eval path)
    phases = [step()[1] for _ in range(timed_steps)]
    _, _, _, t_eval = evaluate()
    t_fwd, t_bwd, t_opt = (sum(p[i] for p in phases) / timed_steps for i in range(3))
    t_step = t_fwd + t_bwd + t_opt
    per_volume = tta_steps * t_step + t_eval
    base = {
        "value": 1.0 / per_volume, "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
        "phases_s": {"forward": t_fwd, "backward": t_bwd, "optimizer": t_opt, "eval_forward_dice": t_eval,
                     "warmup_step_discarded": sum(warm)},
        "sample": f"1 discarded warm-up step ({sum(warm):.2f} s), then {timed_steps} timed adaptation steps (mean {t_step:.2f} s = "
                  f"forward {t_fwd:.2f} + backward {t_bwd:.2f} + optimizer {t_opt:.2f}) and 1 timed final forward+masks+Dice "
                  f"({t_eval:.2f} s) of one {C}x{shape[0]}x{shape[1]}x{shape[2]} volume, extrapolated to S={tta_steps} steps",
    }
    return base, {"state": state, "x": x, "label": v["label"].unsqueeze(0), "loss0": loss0, "logits": z,
                  "pred": pred, "dice": dice, "thr": thr}


def parity_full_size(cfg, ref, device, precision):
    """The GPU path on the volume, weights and single step the CPU baseline just ran: |loss| relative error, logits
    error relative to max|logits|, fraction of mask voxels that differ, largest per-region Dice difference."""
    import oracle
    from multimodal_tta_amd.registry import get_model, get_plugin
    model = get_model(cfg["model"]["name"])(cfg["model"])
    model.load_state_dict(ref["state"])
    plug = get_plugin("entmin_tta")(cfg)
    plug.lane = 5
    plug.setup(model, device)
    res = plug.adapt_volume(ref["x"].to(device), steps=1)
    z = plug.logits(res).cpu()
    loss0 = float(res["losses"].cpu()[0])
    scale = ref["logits"].abs().max().item()
    pred = (torch.sigmoid(z) >= ref["thr"]).to(torch.uint8)
    dice, _, _ = oracle.binary_dice_iou(pred, (ref["label"] > 0.5).to(torch.uint8))
    tol = {"bf16": dict(loss=1e-2, logits=3e-2, dice=2e-2), "fp32": dict(loss=1e-4, logits=2e-3, dice=1e-3)}[precision]
    out = {
        "against": "cpu_baseline step (oracle, fp32, same weights / volume 0 / 1 step + final forward)",
        "loss_rel_err": abs(loss0 - ref["loss0"]) / abs(ref["loss0"]),
        "logits_err_over_max": (z - ref["logits"]).abs().max().item() / scale,
        "mask_mismatch_fraction": (pred != ref["pred"]).float().mean().item(),
        "dice_max_abs_diff": (dice - ref["dice"]).abs().max().item(),
        "dice_gpu": [float(v) for v in dice.view(-1)], "dice_cpu": [float(v) for v in ref["dice"].view(-1)],
        "tolerance": tol,
    }
    out["within_tolerance"] = bool(out["loss_rel_err"] <= tol["loss"] and out["logits_err_over_max"] <= tol["logits"]
                                   and out["dice_max_abs_diff"] <= tol["dice"])
    del plug, model
    return out


def bench_rows(counts_cpu: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's rows of the per-volume table: global volume index i*world + rank (round-robin shard), domain 0,
    loss 0, then dice / iou / valid from the exact counts.  counts_cpu: int64 [n_local, R, 3]."""
    from multimodal_tta_amd.evaluation import dice_iou_from_counts
    n = counts_cpu.shape[0]
    dice, iou, valid = dice_iou_from_counts(counts_cpu)
    return torch.cat([torch.arange(n, dtype=torch.float64).view(-1, 1) * world + rank,
                      torch.zeros(n, 2, dtype=torch.float64), dice.double(), iou.double(), valid.double()], dim=1)


def post_dice_from_table(table: torch.Tensor, R: int) -> float:
    vmask = table[:, 3 + 2 * R:3 + 3 * R] > 0.5
    dvals = table[:, 3:3 + R]
    return float((dvals * vmask).sum().item() / max(1.0, float(vmask.sum().item())))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the adaptation path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK / WORLD_SIZE exported) the RCCL group is formed even for one rank, so a
    # single-GPU box rehearses exactly the calls the N-GPU run makes (init, barrier, all_gather, all_reduce, destroy)
    dist_on = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner to STDOUT when the communicator comes up; stdout carries exactly one JSON line,
        # so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    import multimodal_tta_amd  # noqa: F401
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.evaluation import dice_iou_from_counts, gather_table, table_width
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume

    cfg, shape = build_cfg(args)
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    R = int(cfg["model"]["num_classes"])
    thr = float(cfg["evaluation"]["seg"]["threshold"])
    nvol = args.steps + args.warmup
    lanes = max(1, int(args.lanes if args.lanes is not None else cfg["method"].get("lanes", 1)))
    group = max(1, int(cfg["method"].get("group", 1)))
    lane_streams = ops.lane_streams(lanes, device)      # first GPU work of the process: one hardware queue per lane
    vols = []
    for i in range(nvol):
        v = synth_volume(rank * nvol + i, C, shape, R)
        vols.append((v["image"].unsqueeze(0).to(device), v["label"].unsqueeze(0).to(device)))

    def schedule(first, last, group_, lanes_=None):
        """[(k, [volume indices])]: the volumes [first, last) as lanes x rounds groups of (nearly) EQUAL size <= `group_`,
        dealt round-robin over the lanes - so every lane gets the same number of groups whatever the volume count (20 volumes,
        2 lanes, groups of up to 8: four groups of 5, not 8 + 8 + 4 with one lane idle for the third)."""
        lanes_ = lanes if lanes_ is None else lanes_
        nv_ = last - first
        if nv_ <= 0:
            return []
        rounds = -(-nv_ // (lanes_ * group_))
        ngroups = min(nv_, lanes_ * rounds)
        base, extra = divmod(nv_, ngroups)
        out_, a = [], first
        for k in range(ngroups):
            size = base + (1 if k < extra else 0)
            out_.append((k, list(range(a, a + size))))
            a += size
        return out_

    # the timed volumes as resident group tensors (inputs are in HBM before the timed region starts)
    _grouped = {}

    def group_tensors(idx):
        key = tuple(idx)
        if key not in _grouped:
            _grouped[key] = (torch.cat([vols[i % nvol][0] for i in idx]), torch.cat([vols[i % nvol][1] for i in idx])) \
                if len(idx) > 1 else vols[idx[0] % nvol]
        return _grouped[key]

    def make_lanes(cfg_, lanes_, lane0=0, streams_=None):
        """`lanes_` plugins with the same seeded source weights (episodic: restored per volume), a stream each."""
        torch.manual_seed(42)
        model = get_model(cfg_["model"]["name"])(cfg_["model"])
        plugs_ = []
        streams_ = list(streams_[:lanes_]) if streams_ is not None else ops.lane_streams(lanes_, device)
        for lane in range(lanes_):
            m = model if lane == 0 else get_model(cfg_["model"]["name"])(cfg_["model"])
            if lane:
                m.load_state_dict(model.state_dict())
            p_ = get_plugin("entmin_tta")(cfg_)
            p_.lane = lane0 + lane
            plugs_.append(p_.setup(m, device))
        return plugs_, streams_

    _host = {}

    def host_tensors(idx):
        key = tuple(idx)
        if key not in _host:
            x, y = group_tensors(idx)
            _host[key] = (x.cpu().pin_memory(), y.cpu().pin_memory())
        return _host[key]

    def run_volumes(plugs_, streams_, first, last, counts_, group_=None):
        group_ = group if group_ is None else group_
        for k, idx in schedule(first, last, group_, len(plugs_)):
            lane = k % len(plugs_)
            x, y = group_tensors(idx)
            with torch.cuda.stream(streams_[lane]):
                if args.from_host:        # image and label cross PCIe inside the timed region
                    hx, hy = host_tensors(idx)
                    x, y = hx.to(device, non_blocking=True), hy.to(device, non_blocking=True)
                res = plugs_[lane].adapt_volume(x)
                ops.mask_dice_counts(res["logits_cl"], y, thr, counts_[idx[0] % nvol:idx[0] % nvol + len(idx)], None)

    plugs, streams = make_lanes(cfg, lanes, streams_=lane_streams)
    plug = plugs[0]
    counts = torch.zeros((nvol, R, 3), dtype=torch.int64, device=device)

    # untimed: the W warm-up volumes, then one dry run of the timed schedule itself, so that every lane has captured the
    # graph of every group size it will see (a partial last group included) and every group tensor is resident
    run_volumes(plugs, streams, 0, args.warmup, counts)
    run_volumes(plugs, streams, args.warmup, nvol, counts)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_volumes(plugs, streams, args.warmup, nvol, counts)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    rows = bench_rows(counts[args.warmup:].cpu(), rank, world)
    tg0 = time.perf_counter()
    table = gather_table(rows.to(device), args.steps * world, world)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - tg0
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, t_local, -t_local, t_gather], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, t_local_max, t_local_min, t_gather = float(t[0]), float(t[1]), -float(t[2]), float(t[3])
    assert table.shape == (args.steps * world, table_width(R))
    post_dice = post_dice_from_table(table, R)

    out = {
        "metric": "adapted volumes/sec", "value": args.steps * world / elapsed, "unit": "volumes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
        "data": "synthetic" + (" (volumes start in pinned host memory: PCIe-inclusive, not the headline)" if args.from_host else ""),
        "post_tta_dice": post_dice,
        "config": {
            "workload": f"{cfg['model']['name']} {C}x{shape[0]}x{shape[1]}x{shape[2]} {args.task}-shaped volume: "
                        f"S={args.tta_steps} entropy-min steps (fwd+bwd+Adam, all parameters) + final forward + Dice"
                        + (f"; modalities {list(cfg['method'].get('missing_modalities', []))} missing, modality dropout p = "
                           f"{cfg['method']['moddrop']['p']} per step (seeded)" if args.method == "tta_moddrop" else ""),
            "tta_steps": args.tta_steps, "volume": [C, *shape], "adapted_params": str(cfg["method"]["params"]),
            "precision": ("bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulate, for forward, input-gradient "
                          "and 27-tap weight-gradient convs; forward activations with >= 32 channels stored as "
                          f"{getattr(plug, 'storage', None) or 'fp32'}, their gradients as "
                          f"{'bf16' if getattr(plug.rt, 'grad_bf16', False) else 'fp32'}; statistics / norms / loss / reduced weight "
                          "gradients / 1x1 and thin weight gradients / optimizer / master weights fp32") if args.precision == "bf16"
            else "fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32)", "weights": "seeded default init (no checkpoint offline)",
            "parallelism": f"{world} rank(s), one per GPU, volumes sharded round-robin, no data-path collective; the "
                           f"per-volume Dice table is merged by one all_gather; per GPU {lanes} lane(s) (own stream = "
                           f"hardware queue, own graph) x {group} volume(s) per launch sequence, every volume on its own "
                           "replica of the weights and optimizer state",
            "graph": bool(plug.use_graph), "lanes": lanes, "group": group,
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
        },
        # how far two runs of this line differ (measured): the timed region is short
        "repeatability": {"same_box_rel": 0.005, "cross_box_rel": 0.03,
                          "timed_region_s": elapsed, "note": "same command twice on one box / on different boxes of the pool"},
        "ranks": {"volumes_per_s_min": args.steps / t_local_max, "volumes_per_s_max": args.steps / t_local_min,
                  "gather_ms": 1000.0 * t_gather},
    }

    if rank == 0 and not args.no_profile_pass:
        first_group = schedule(args.warmup, nvol, group)[0][1]
        out["roofline"] = roofline_block(args, cfg, plug, group_tensors(first_group)[0], elapsed / args.steps)
        out["config"]["algorithmic_conv_tflop_per_volume"] = out["roofline"]["whole_volume"]["algorithmic_tflop"]
    if rank == 0 and world == 1 and not args.no_variants:
        # the same workload (a) one volume at a time (one lane, group 1: the single-volume latency - and the check that the
        # grouped, multi-lane headline run produced the SAME results for the same volumes), (b) in fp32 mode: the reference's
        # own arithmetic
        variants = {}
        # (the equality check runs one volume at a time under the HEADLINE's launch geometry - the geometry fixes the summation
        # order, method.tune_volumes -; the latency figure under the geometry tuned for one volume in flight)
        for name, prec, nl, ng, tune in (("one_volume_same_geometry", args.precision, 1, 1, lanes * group),
                                         ("one_volume", args.precision, 1, 1, None),
                                         ("one_lane", args.precision, 1, group, None),      # one launch sequence of `group` volumes
                                         ("fp32", "fp32", lanes, group, None)):
            if (prec, nl, ng) == (args.precision, lanes, group):
                continue
            a2 = argparse.Namespace(**vars(args))
            a2.precision, a2.group, a2.lanes = prec, ng, nl
            cfg2, _ = build_cfg(a2)
            if tune is not None:
                cfg2["method"]["tune_volumes"] = tune
            pl2, st2 = make_lanes(cfg2, nl, lane0=8 if name == "fp32" else 6, streams_=streams)      # the same queues
            c2 = torch.zeros_like(counts)
            nv = min(args.steps, 4 if tune is not None else max(8, 2 * nl * ng))   # steady state: two rounds of every lane
            run_volumes(pl2, st2, args.warmup, args.warmup + nv, c2, group_=ng)      # untimed: graphs of this schedule
            torch.cuda.synchronize()
            tv = time.perf_counter()
            run_volumes(pl2, st2, args.warmup, args.warmup + nv, c2, group_=ng)
            torch.cuda.synchronize()
            tv = time.perf_counter() - tv
            variants[name] = {"value": nv / tv, "unit": "volumes/s", "precision": prec, "lanes": nl, "group": ng, "volumes": nv}
            if name == "one_volume_same_geometry":
                # Dice counts of the grouped / multi-lane headline run == the same volumes one at a time, exactly
                out["lanes_equal"] = bool(torch.equal(c2[args.warmup:args.warmup + nv], counts[args.warmup:args.warmup + nv]))
            if name == "one_volume":
                variants[name]["latency_ms"] = 1000.0 * tv / nv
            del pl2, st2
        out["variants"] = variants
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.method == "tta_entmin":
        out["cpu_baseline"], ref = cpu_baseline(cfg, shape, args.tta_steps)
        out["parity_full_size"] = parity_full_size(cfg, ref, device, args.precision)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()            # rank 0 may still have been in its profile pass
        dist.destroy_process_group()


def roofline_block(args, cfg, plug, x0, s_per_volume):
    """Instrumented eager pass: events around every conv launch, on the launch stream (multi-kernel entry points launch
    only their main kernel here, so a call's duration IS the duration of the kernel rocprofv3 lists under that name; the
    outputs of this pass are thrown away)."""
    from multimodal_tta_amd import _lib, ops
    prof = ops.KernelProfiler(reps=4)
    ops.PROFILER = prof
    saved = plug.use_graph
    plug.use_graph = False
    _lib.load().mmtta_set_option(1, 1)
    B = int(x0.shape[0])        # the volumes of one group: every figure below is divided down to ONE volume
    try:
        plug.adapt_volume(x0, steps=2)
        torch.cuda.synchronize()
    finally:
        _lib.load().mmtta_set_option(1, 0)
        plug.use_graph = saved
        ops.PROFILER = None
    summ = prof.summary()
    total_ms = sum(d["ms"] for d in summ.values())
    # algorithmic conv FLOPs and bytes of one adapted volume from the same pass (2 steps + 1 final forward recorded)
    per_kind = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for rname, _launches, rflops, _e0, _e1, detail, rbytes in prof.records:
        kind = detail.split(" ")[0]
        kind = "fwd" if kind.startswith("fwd") else ("dgrad" if kind.startswith("dgrad") else "wgrad")
        per_kind[kind][0] += rflops / prof.reps / B
        per_kind[kind][1] += rbytes / prof.reps / B
    S = args.tta_steps
    f_fwd, b_fwd = per_kind["fwd"][0] / 3.0, per_kind["fwd"][1] / 3.0
    f_vol = S * (f_fwd + per_kind["dgrad"][0] / 2.0 + per_kind["wgrad"][0] / 2.0) + f_fwd
    n_params = plug.rt.arena.n_train
    R = int(cfg["model"]["num_classes"])
    nvox = int(x0.shape[2] * x0.shape[3] * x0.shape[4])
    # B = B_fwd * (3S + 1) + 28 B * P * S (optimizer) + 2 * R * DHW (mask + Dice), fp32 tensors
    b_vol_survey = b_fwd * (3 * S + 1) + 28.0 * n_params * S + 2.0 * R * nvox
    b_vol_layers = S * (b_fwd + per_kind["dgrad"][1] / 2.0 + per_kind["wgrad"][1] / 2.0) + b_fwd + 28.0 * n_params * S \
        + 2.0 * R * nvox
    bf = args.precision == "bf16"
    whole = {"algorithmic_tflop": f_vol / 1e12, "effective_tflops_per_gpu": f_vol / s_per_volume / 1e12,
             "frac_of_mfma_peak": f_vol / s_per_volume / 1e12 / (PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS),
             "algorithmic_gb": b_vol_survey / 1e9, "algorithmic_gb_per_layer_io": b_vol_layers / 1e9,
             "achieved_gbps": b_vol_survey / s_per_volume / 1e9, "frac_of_hbm_peak": b_vol_survey / s_per_volume / 1e9 / PEAK_HBM_GBPS,
             "b_fwd_gb": b_fwd / 1e9}
    name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    sec = d["ms"] * 1e-3
    tflops = d["flops"] / sec / 1e12
    gbps = d["bytes"] / sec / 1e9
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of this build
    # (scripts/pmc_layers.sh -> scripts/pmc_summary.py --json; FETCH_SIZE x2 on gfx950 + WRITE_SIZE), else null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath)).get(name)
        if t:
            traffic = t["fetch_bytes"] + t["write_bytes"]
    # the same kernel's average launch duration in the committed rocprofv3 kernel trace of this build's bench command
    # (scripts/trace_summary.py --json -> profiles/trace_avg_us.json): in situ, lanes interleaving - beside the event figure
    trace_us = None
    kpath = os.path.join(ROOT, "profiles", "trace_avg_us.json")
    if os.path.exists(kpath):
        tr = json.load(open(kpath))
        hits = [v for k, v in tr.get("kernels", {}).items() if k.replace(" ", "").startswith(name.replace("bf16", "true").replace(" ", "").split(">")[0])]
        if hits:
            trace_us = sum(h["total_us"] for h in hits) / max(1, sum(h["calls"] for h in hits))
    hbm = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
           "bytes_per_launch": d["bytes"] / d["launches"]}
    mfma = {"bound": "mfma", "achieved": tflops, "peak": mfma_peak(name), "unit": "TFLOP/s", "frac": tflops / mfma_peak(name),
            "flops_per_launch": d["flops"] / d["launches"]}
    first, second = (hbm, mfma) if bf else (mfma, hbm)
    block = dict(first)
    block.update({
        "kernel": name, "traffic": traffic, "avg_launch_us": 1000.0 * d["ms"] / d["launches"], "launches": d["launches"],
        "trace_avg_launch_us": trace_us, "volumes_per_launch": B,
        "share_of_conv_time": d["ms"] / total_ms,
        ("mfma" if bf else "hbm"): {k: v for k, v in second.items() if k != "bound"},
        "all_conv_kernels": {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                 "mfma_frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak(k),
                                 "gbps": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                 "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, "ms": v["ms"],
                                 "launches": v["launches"]} for k, v in sorted(summ.items())},
        "conv_tflops_overall": sum(v["flops"] for v in summ.values()) / (total_ms * 1e-3) / 1e12,
        "whole_volume": whole,
    })
    return block


if __name__ == "__main__":
    main()
