#!/usr/bin/env python
"""Per-layer conv kernel times of one adaptation step (instrumented eager pass, events on the launch stream).

    python scripts/layer_times.py [--model unet] [--precision bf16] > gpurun_out/layers.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="unet")
    ap.add_argument("--task", default="brats")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    import bench
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume
    ns = argparse.Namespace(task=args.task, model=args.model, tta_steps=10, precision=args.precision, no_graph=True, shape=None)
    cfg, shape = bench.build_cfg(ns)
    torch.manual_seed(42)
    model = get_model(cfg["model"]["name"])(cfg["model"])
    plug = get_plugin("entmin_tta")(cfg).setup(model, "cuda")
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    x = synth_volume(0, C, shape, int(cfg["model"]["num_classes"]))["image"].unsqueeze(0).cuda()
    plug.adapt_volume(x, steps=1)
    prof = ops.KernelProfiler(reps=args.reps)
    ops.PROFILER = prof
    plug.adapt_volume(x, steps=1)
    ops.PROFILER = None
    rows = prof.by_layer()
    total = sum(d["ms"] for d in rows.values()) / args.reps
    print(f"conv time of 1 step + 1 final forward: {total:.3f} ms (per-call times are averages over {args.reps} back-to-back reps)")
    print(f"{'us/call':>9} {'calls':>5} {'TFLOP/s':>8}  kernel | layer")
    for (name, detail), d in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        ncall = sum(1 for r in prof.records if r[0] == name and r[5] == detail)
        us = 1000.0 * d["ms"] / (ncall * args.reps)
        print(f"{us:9.1f} {ncall:5d} {d['flops'] / (d['ms'] * 1e-3) / 1e12:8.1f}  {name} | {detail}")


if __name__ == "__main__":
    main()
