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
    ap.add_argument("--group", type=int, default=8, help="volumes per launch sequence (method.group)")
    ap.add_argument("--tune-volumes", type=int, default=16, help="launch geometry of the headline arrangement")
    ap.add_argument("--what-if", default="", help="comma list of no_norm_on_load, no_stats: the timed repeats of the forward / "
                    "input-gradient convolutions drop that part of their work (where do a kernel's microseconds go)")
    args = ap.parse_args()
    import bench
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume
    ns = argparse.Namespace(task=args.task, model=args.model, tta_steps=10, precision=args.precision, no_graph=True, shape=None,
                            group=args.group, lanes=1, tune_volumes=args.tune_volumes)
    cfg, shape = bench.build_cfg(ns)
    torch.manual_seed(42)
    model = get_model(cfg["model"]["name"])(cfg["model"])
    plug = get_plugin("entmin_tta")(cfg).setup(model, "cuda")
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    G = int(plug.group)
    x = torch.stack([synth_volume(i, C, shape, int(cfg["model"]["num_classes"]))["image"] for i in range(G)]).cuda()
    plug.adapt_volume(x, steps=1)
    prof = ops.KernelProfiler(reps=args.reps, what_if=[w for w in args.what_if.split(",") if w])
    ops.PROFILER = prof
    plug.adapt_volume(x, steps=1)
    ops.PROFILER = None
    rows = prof.by_layer()
    total = sum(d["ms"] for d in rows.values()) / args.reps
    print(f"conv time of 1 step + 1 final forward of a group of {G} volumes: {total:.3f} ms = {total / G:.3f} ms per volume (per-call "
          f"times are averages over {args.reps} back-to-back reps)")
    print(f"{'us/call':>9} {'calls':>5} {'TFLOP/s':>8} {'TB/s':>6} {'us at 6.3 TB/s':>14}  kernel | layer")
    for (name, detail), d in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        ncall = sum(1 for r in prof.records if r[0] == name and r[5] == detail)
        us = 1000.0 * d["ms"] / (ncall * args.reps)
        sec = d["ms"] * 1e-3
        print(f"{us:9.1f} {ncall:5d} {d['flops'] / sec / 1e12:8.1f} {d['bytes'] / sec / 1e12:6.2f} {d['bytes'] / (ncall * args.reps) / 6.3e6:14.1f}"
              f"  {name} | {detail}")


if __name__ == "__main__":
    main()
