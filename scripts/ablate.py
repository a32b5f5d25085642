#!/usr/bin/env python
"""Marginal cost of kernel families in the real (two-lane, graph-replayed) adaptation loop: the same bench loop with one
family of launches skipped (RESULTS ARE WRONG in every ablated arm - this measures time only).  Decides which fusion
is worth building: a family whose removal does not move volumes/s is hidden behind the other lane.

    python scripts/ablate.py [--lanes 2] [--volumes 6] > gpurun_out/ablate.txt
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per lane (see bench.py)
import torch  # noqa: E402


LANE_STREAMS = {}


def run(arm, lanes, volumes):
    import multimodal_tta_amd  # noqa: F401
    from multimodal_tta_amd import _lib, engine, ops
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume

    saved = {}

    def patch(obj, name, fn):
        saved[(obj, name)] = getattr(obj, name)
        setattr(obj, name, fn)

    lib = _lib.load()
    lib.mmtta_set_option(1, 1 if arm == "main_kernels_only" else 0)
    if arm == "no_norm_bwd":          # dy := dT (no reduce / finalize / apply)
        def bwd(self, pool, key, dT, y, nl, dy, training, accumulate=False):
            ops.lincomb([dT], [1.0], dy)
        patch(engine.NormLayer, "backward", bwd)
    if arm == "no_norm_bwd_at_all":   # not even the copy
        patch(engine.NormLayer, "backward", lambda self, pool, key, dT, y, nl, dy, training, accumulate=False: None)
    if arm == "no_stats_finalize":
        orig = engine.NormLayer.finalize
        cache = {}
        def fin(self, pool, key, part, rows_per_n, n, count, training):
            k = (id(self), key)
            if k not in cache:
                cache[k] = orig(self, pool, key, part, rows_per_n, n, count, training)
            return cache[k]
        patch(engine.NormLayer, "finalize", fin)
    if arm == "no_pack":
        orig_pack = engine.Runtime.pack_all
        done = set()
        def pack(self):
            if id(self) not in done or len(done) < 0:
                done.add(id(self))
                return orig_pack(self)
            if not hasattr(self, "_npack"):
                self._npack = 0
            self._npack += 1
            if self._npack < 3:
                return orig_pack(self)
        patch(engine.Runtime, "pack_all", pack)
    if arm == "no_wgrad":
        patch(engine.ConvLayer, "wgrad", lambda self, x, x_nl, dy, accumulate=False: None)
    if arm == "no_dgrad":
        patch(ops.ConvOp, "dgrad", lambda self, dy, dx, accumulate=False: None)
    if arm == "no_optimizer":
        from multimodal_tta_amd import tta
        patch(tta.EntropyMinimizationTTA, "optimizer_step", lambda self: None)

    device = torch.device("cuda", 0)
    cfg = compose(overrides=["task=brats", "dataset=brats", "model=unet", "method=tta_entmin", "method.steps=10",
                             "method.precision=bf16"])
    torch.manual_seed(42)
    model = get_model("unet")(cfg["model"])
    plugs = []
    if "pool" not in LANE_STREAMS:          # created and touched once, before any other stream: one hardware queue each
        LANE_STREAMS["pool"] = ops.lane_streams(6, device)
    streams = LANE_STREAMS["pool"][:lanes]
    for lane in range(lanes):
        m = model if lane == 0 else get_model("unet")(cfg["model"])
        if lane:
            m.load_state_dict(model.state_dict())
        p = get_plugin("entmin_tta")(cfg)
        p.lane = 20 + lane
        plugs.append(p.setup(m, device))
    vols = [synth_volume(i, 4, (128, 128, 128), 3)["image"].unsqueeze(0).to(device) for i in range(volumes + lanes)]

    def one(i):
        lane = i % lanes
        with torch.cuda.stream(streams[lane]):
            plugs[lane].adapt_volume(vols[i])

    for i in range(lanes):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(lanes, lanes + volumes):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for (obj, name), fn in saved.items():
        setattr(obj, name, fn)
    lib.mmtta_set_option(1, 0)
    del plugs
    torch.cuda.empty_cache()
    return volumes / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, nargs="+", default=[2])
    ap.add_argument("--volumes", type=int, default=6)
    ap.add_argument("--arms", nargs="+", default=["baseline", "main_kernels_only", "no_norm_bwd", "no_norm_bwd_at_all",
                                                  "no_stats_finalize", "no_pack", "no_wgrad", "no_dgrad", "no_optimizer",
                                                  "baseline"])
    args = ap.parse_args()
    for lanes in args.lanes:
        for arm in args.arms:
            v = run(arm, lanes, args.volumes)
            print(f"lanes {lanes}  {arm:22s} {v:7.2f} volumes/s   {1000.0 / v:7.2f} ms/volume", flush=True)


if __name__ == "__main__":
    main()
