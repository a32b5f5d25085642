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
  roofline     - the dominant kernel (by measured time), algorithmic FLOPs per launch over its
                 measured launch duration (events on the launch stream), against the dense MFMA peak of
                 the operand type that kernel uses (bf16 2.5 PF, fp32 157.3 TF)
  cpu_baseline - the oracle (torch CPU restatement of the reference path) timed on this host on a
                 bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA": ~2.5 PF dense (not the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0


def mfma_peak(kernel_name: str) -> float:
    """Dense MFMA peak of the operand type the named kernel feeds the matrix cores with."""
    return PEAK_BF16_MFMA_TFLOPS if "bf16" in kernel_name else PEAK_FP32_MFMA_TFLOPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed adapted volumes per rank")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="unet", choices=["unet", "unet_multimodal_deepfusion"])
    ap.add_argument("--task", default="brats", choices=["brats", "hecktor21"])
    ap.add_argument("--tta-steps", type=int, default=10)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="D H W (default: 128^3 brats, 48x144x144 hecktor)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="bf16 (BASELINE configs[1]): bf16 MFMA operands, fp32 accumulate/storage; fp32: exact fp32 MFMA")
    ap.add_argument("--lanes", type=int, default=None,
                    help="volumes adapted concurrently per GPU, each with its own weights, buffers, graph and stream "
                         "(episodic adaptation has no cross-volume state; one volume alone leaves most CUs waiting)")
    ap.add_argument("--side-streams", type=int, default=None, help="side streams for the weight gradients (default: config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def build_cfg(args):
    from multimodal_tta_amd.config import compose
    ov = [f"task={args.task}", f"dataset={args.task}", f"model={args.model}", "method=tta_entmin",
          f"method.steps={args.tta_steps}", f"method.precision={args.precision}"]
    if args.task == "hecktor21" and args.model != "unet":
        ov += ["model.num_modalities=2", "model.num_classes=1"]
    if args.no_graph:
        ov += ["method.use_graph=false"]
    if getattr(args, "side_streams", None) is not None:
        ov += [f"method.side_streams={args.side_streams}"]
    cfg = compose(overrides=ov)
    shape = args.shape or ([128, 128, 128] if args.task == "brats" else [48, 144, 144])
    cfg["dataset"]["synthetic"]["shape"] = list(shape)
    return cfg, tuple(shape)


def cpu_baseline(cfg, shape, tta_steps):
    """Oracle on the host cores: 1 adaptation step + 1 final forward of one full-size volume, extrapolated to
    S steps (a bounded sample: a whole 10-step volume is ~1 min of CPU work)."""
    import oracle
    from multimodal_tta_amd.synth import synth_volume

    torch.manual_seed(42)
    model = oracle.MODELS[cfg["model"]["name"]](cfg["model"])
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    R = int(cfg["model"]["num_classes"])
    v = synth_volume(0, C, shape, R)
    x = v["image"].unsqueeze(0)
    opt = oracle.build_adam(list(model.named_parameters()), cfg["training"])
    model.train()
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = oracle.entropy_loss(model(x))
    loss.backward()
    opt.step()
    t_step = time.perf_counter() - t0
    model.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        z = model(x)
        pred, gt = oracle.masks_from_logits(z, v["label"].unsqueeze(0), 0.5)
        oracle.binary_dice_iou(pred, gt)
    t_fwd = time.perf_counter() - t0
    per_volume = tta_steps * t_step + t_fwd
    return {
        "value": 1.0 / per_volume, "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"1 adaptation step ({t_step:.2f} s) + 1 final forward+Dice ({t_fwd:.2f} s) of one "
                  f"{C}x{shape[0]}x{shape[1]}x{shape[2]} volume, extrapolated to S={tta_steps} steps",
    }


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
    torch.manual_seed(42)
    model = get_model(cfg["model"]["name"])(cfg["model"])
    plug = get_plugin("entmin_tta")(cfg).setup(model, device)
    lanes = max(1, int(args.lanes if args.lanes is not None else cfg["method"].get("lanes", 1)))
    plugs, streams = [plug], [torch.cuda.Stream(device=device)]
    for lane in range(1, lanes):          # same source weights in every lane (episodic: restored per volume)
        m2 = get_model(cfg["model"]["name"])(cfg["model"])
        m2.load_state_dict(model.state_dict())
        p2 = get_plugin("entmin_tta")(cfg)
        p2.lane = lane
        plugs.append(p2.setup(m2, device))
        streams.append(torch.cuda.Stream(device=device))
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    R = int(cfg["model"]["num_classes"])
    thr = float(cfg["evaluation"]["seg"]["threshold"])
    nvol = args.steps + args.warmup
    vols = []
    for i in range(nvol):
        v = synth_volume(rank * nvol + i, C, shape, R)
        vols.append((v["image"].unsqueeze(0).to(device), v["label"].unsqueeze(0).to(device)))
    counts = torch.zeros((nvol, R, 3), dtype=torch.int64, device=device)

    def one_volume(i):
        lane = i % lanes
        x, y = vols[i]
        with torch.cuda.stream(streams[lane]):
            res = plugs[lane].adapt_volume(x)
            ops.mask_dice_counts(res["logits_cl"], y, thr, counts[i:i + 1], None)

    for i in range(max(args.warmup, lanes)):      # every lane captures its graph before the timed region
        one_volume(i % nvol)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, nvol):
        one_volume(i)
    torch.cuda.synchronize()
    dice, iou, valid = dice_iou_from_counts(counts[args.warmup:].cpu())
    rows = torch.cat([torch.arange(args.steps, dtype=torch.float64).view(-1, 1) * world + rank,
                      torch.zeros(args.steps, 2, dtype=torch.float64), dice.double(), iou.double(), valid.double()], dim=1)
    table = gather_table(rows.to(device), args.steps * world, world)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert table.shape == (args.steps * world, table_width(R))
    vmask = table[:, 3 + 2 * R:3 + 3 * R] > 0.5
    dvals = table[:, 3:3 + R]
    post_dice = float((dvals * vmask).sum().item() / max(1.0, float(vmask.sum().item())))

    out = {
        "metric": "adapted volumes/sec", "value": args.steps * world / elapsed, "unit": "volumes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
        "post_tta_dice": post_dice,
        "config": {
            "workload": f"{cfg['model']['name']} {C}x{shape[0]}x{shape[1]}x{shape[2]} {args.task}-shaped volume: "
                        f"S={args.tta_steps} entropy-min steps (fwd+bwd+Adam, all parameters) + final forward + Dice",
            "tta_steps": args.tta_steps, "volume": [C, *shape], "adapted_params": str(cfg["method"]["params"]),
            "precision": ("bf16 MFMA operands (v_mfma_f32_32x32x16_bf16) for forward/input-gradient convs, fp32 accumulate; "
                          "fp32 storage, norms, loss, weight gradients, Adam") if args.precision == "bf16"
            else "fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32)", "weights": "seeded default init (no checkpoint offline)",
            "parallelism": f"{world} rank(s), one per GPU, volumes sharded round-robin, one all_gather of the Dice table; "
                           f"{lanes} volume(s) in flight per GPU on separate streams",
            "graph": bool(plug.use_graph), "lanes": lanes,
        },
    }

    if rank == 0 and not args.no_profile_pass:
        # instrumented eager pass: events around every conv launch, on the launch stream
        # (multi-kernel entry points launch only their main kernel here, so a call's duration IS the duration of the
        # kernel rocprofv3 lists under that name; the outputs of this pass are thrown away)
        from multimodal_tta_amd import _lib
        prof = ops.KernelProfiler(reps=4)
        ops.PROFILER = prof
        saved = plug.use_graph
        plug.use_graph = False
        _lib.load().mmtta_set_option(1, 1)
        try:
            plug.adapt_volume(vols[0][0], steps=2)
            torch.cuda.synchronize()
        finally:
            _lib.load().mmtta_set_option(1, 0)
        plug.use_graph = saved
        ops.PROFILER = None
        summ = prof.summary()
        total_ms = sum(d["ms"] for d in summ.values())
        # algorithmic conv FLOPs of one adapted volume from the same pass (2 steps + 1 final forward were recorded)
        per_kind = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
        for rname, _launches, rflops, _e0, _e1, detail in prof.records:
            kind = detail.split(" ")[0]
            kind = "fwd" if kind.startswith("fwd") else ("dgrad" if kind.startswith("dgrad") else "wgrad")
            per_kind[kind] += rflops / prof.reps
        f_fwd = per_kind["fwd"] / 3.0
        f_vol = args.tta_steps * (f_fwd + per_kind["dgrad"] / 2.0 + per_kind["wgrad"] / 2.0) + f_fwd
        eff = f_vol / (elapsed / (args.steps * world) * world) / 1e12     # per GPU: one volume's FLOPs / its wall time
        out["config"]["algorithmic_conv_tflop_per_volume"] = f_vol / 1e12
        whole = {"effective_tflops_per_gpu": eff,
                 "frac_of_mfma_peak": eff / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_FP32_MFMA_TFLOPS)}
        name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        peak = mfma_peak(name)
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of this build
        # (scripts/pmc_layers.sh -> scripts/pmc_summary.py --json; FETCH_SIZE x2 on gfx950 + WRITE_SIZE), else null
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get(name)
            if t:
                traffic = t["fetch_bytes"] + t["write_bytes"]
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic,
            "avg_launch_us": 1000.0 * d["ms"] / d["launches"], "launches": d["launches"],
            "flops_per_launch": d["flops"] / d["launches"],
            "share_of_conv_time": d["ms"] / total_ms,
            "all_conv_kernels": {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                     "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak(k), "ms": v["ms"],
                                     "launches": v["launches"]} for k, v in sorted(summ.items())},
            "conv_tflops_overall": sum(v["flops"] for v in summ.values()) / (total_ms * 1e-3) / 1e12,
            "whole_volume": whole,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, shape, args.tta_steps)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()            # rank 0 may still have been in its profile pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
