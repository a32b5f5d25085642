"""Sweep the launch-geometry knobs of libmmtta.so (mmtta_set_option) inside ONE process on one GPU, so that settings are
compared on the same box: adapted volumes/s of the bench workload (unet 4x128^3, S = 10, bf16) per setting.

usage: python scripts/sweep_tuning.py [--lanes 2] [--volumes 8] [--repeat 2]
"""
import argparse
import gc
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per lane (see bench.py)
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_tta_amd import _lib  # noqa: E402

KEYS = {"splitk_below": 2, "splitk_target": 3, "wgrad_workgroups": 4, "wgrad_thin_slabs": 5}
SETTINGS = [dict(splitk_below=b, splitk_target=t, wgrad_workgroups=w, wgrad_thin_slabs=th)
            for (b, t) in ((192, 256), (96, 128), (48, 64)) for (w, th) in ((256, 256), (128, 128), (128, 256), (64, 128))]


LANE_STREAMS = {}


def run(setting, lanes, volumes):
    import multimodal_tta_amd  # noqa: F401
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume
    lib = _lib.load()
    for k, v in setting.items():
        assert ops.set_option(KEYS[k], int(v)) > 0      # also drops cached launch plans
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
        p.lane = lane
        plugs.append(p.setup(m, device))
    vols = []
    for i in range(volumes + lanes):
        v = synth_volume(i, 4, (128, 128, 128), 3)
        vols.append((v["image"].unsqueeze(0).to(device), v["label"].unsqueeze(0).to(device)))
    counts = torch.zeros((len(vols), 3, 3), dtype=torch.int64, device=device)

    def one(i):
        lane = i % lanes
        with torch.cuda.stream(streams[lane]):
            res = plugs[lane].adapt_volume(vols[i][0])
            ops.mask_dice_counts(res["logits_cl"], vols[i][1], 0.5, counts[i:i + 1], None)

    for i in range(lanes):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(lanes, lanes + volumes):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del plugs, streams, vols, model
    gc.collect()
    torch.cuda.empty_cache()
    return volumes / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, nargs="+", default=[2])
    ap.add_argument("--volumes", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--quick", action="store_true", help="the neighbourhood of the current defaults only")
    ap.add_argument("--around", action="store_true", help="the current defaults and one knob moved at a time")
    a = ap.parse_args()
    if a.quick:
        SETTINGS[:] = [dict(splitk_below=b, splitk_target=t, wgrad_workgroups=w, wgrad_thin_slabs=th)
                       for (b, t) in ((96, 128), (192, 256)) for (w, th) in ((128, 256), (256, 256), (192, 256), (128, 128), (96, 256))]
    if a.around:
        base = dict(splitk_below=96, splitk_target=128, wgrad_workgroups=128, wgrad_thin_slabs=256)
        SETTINGS[:] = [base] + [dict(base, **d) for d in (
            dict(wgrad_thin_slabs=384), dict(wgrad_thin_slabs=512), dict(wgrad_workgroups=96), dict(wgrad_workgroups=160),
            dict(splitk_below=64, splitk_target=96), dict(splitk_below=128, splitk_target=160))]
    for rep in range(a.repeat):
        for lanes in a.lanes:
            for s in SETTINGS:
                print(f"pass {rep} lanes {lanes} {s}: {run(s, lanes, max(a.volumes, 2 * lanes)):.2f} volumes/s", flush=True)


if __name__ == "__main__":
    main()
