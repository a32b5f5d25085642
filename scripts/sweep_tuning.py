"""Sweep volumes in flight (lanes x group) and the launch-geometry knobs of libmmtta.so (mmtta_set_option) inside ONE
process on one GPU, so that settings are compared on the same box: adapted volumes/s of the bench workload (unet
4x128^3, S = 10, bf16) per setting.

  lanes  independent launch sequences on their own streams / hardware queues (method.lanes)
  group  volumes per launch sequence, each on its own replica of the weights (method.group, mmtta_param_sets)

The geometry knobs are PER BATCH ITEM since round 3 (every volume of a group is computed as if launched alone), so the
workgroups of a launch are the knob value x group.

usage: python scripts/sweep_tuning.py [--combos 4x1 2x4 2x8] [--knobs default|splitk|wgrad|all] [--volumes 32] [--repeat 2]
"""
import argparse
import gc
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per lane (see bench.py)
os.environ["MMTTA_NO_AUTOTUNE"] = "1"                # the knobs are set by hand here
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_tta_amd import _lib  # noqa: E402

KEYS = {"splitk_below": 2, "splitk_target": 3, "wgrad_workgroups": 4, "wgrad_thin_slabs": 5, "cls_fused_min": 12}
BASE = dict(splitk_below=96, splitk_target=128, wgrad_workgroups=128, wgrad_thin_slabs=256)
KNOBS = {
    "default": [BASE],
    "splitk": [dict(BASE, splitk_below=b, splitk_target=t) for (b, t) in ((96, 128), (48, 64), (24, 32), (1, 1))],
    "wgrad": [dict(BASE, wgrad_workgroups=w, wgrad_thin_slabs=th) for (w, th) in ((128, 256), (64, 128), (32, 64), (64, 256), (32, 32))],
}
KNOBS["fine"] = [dict(splitk_below=b, splitk_target=t, wgrad_workgroups=w, wgrad_thin_slabs=th)
                 for (b, t) in ((24, 32), (12, 16)) for (w, th) in ((32, 64), (16, 32), (16, 64), (48, 96))]
KNOBS["scaled"] = ["scaled"]        # the package's own rule (ops.tune_for_volumes_in_flight)
# one knob at a time around the scaled rule's values for 24 volumes in flight (3 x 8)
_V24 = dict(splitk_below=16, splitk_target=22, wgrad_workgroups=22, wgrad_thin_slabs=43, cls_fused_min=22)
KNOBS["v24"] = [_V24] + [dict(_V24, splitk_below=b, splitk_target=t) for (b, t) in ((8, 11), (32, 43), (1, 1))] + \
               [dict(_V24, wgrad_workgroups=w) for w in (11, 43)] + [dict(_V24, wgrad_thin_slabs=t) for t in (21, 86)] + \
               [dict(_V24, cls_fused_min=c) for c in (11, 43)]
KNOBS["v24w"] = [dict(_V24, wgrad_workgroups=w) for w in (22, 43, 64, 96, 128)]
KNOBS["v24p"] = [dict(_V24, wgrad_workgroups=w) for w in (22, 16, 32, 12)]          # around 22 once the column pairs took a CU each
KNOBS["all"] = KNOBS["splitk"] + KNOBS["wgrad"][1:] + [dict(splitk_below=24, splitk_target=32, wgrad_workgroups=32, wgrad_thin_slabs=64),
                                                       dict(splitk_below=48, splitk_target=64, wgrad_workgroups=64, wgrad_thin_slabs=128)]

LANE_STREAMS = {}


def run(setting, lanes, group, volumes, model_name="unet", shape=(128, 128, 128), task="brats"):
    import multimodal_tta_amd  # noqa: F401
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume
    _lib.load()
    if setting == "scaled":                 # the values of the package's own rule (set by hand: autotune is off in this script)
        vals = ops.tune_for_volumes_in_flight(lanes * group)
        setting = {name: vals[key] for name, key in KEYS.items()}
    for k, v in setting.items():
        assert ops.set_option(KEYS[k], max(1, int(v))) >= 0      # also drops cached launch plans
    device = torch.device("cuda", 0)
    ov = [f"task={task}", f"dataset={task}", f"model={model_name}", "method=tta_entmin", "method.steps=10", "method.precision=bf16",
          f"method.group={group}"]
    if task == "hecktor21" and model_name != "unet":
        ov += ["model.num_modalities=2", "model.num_classes=1"]
    cfg = compose(overrides=ov)
    C = int(cfg["model"].get("in_channels", cfg["model"].get("num_modalities", 4)))
    R = int(cfg["model"]["num_classes"])
    torch.manual_seed(42)
    model = get_model(model_name)(cfg["model"])
    plugs = []
    if "pool" not in LANE_STREAMS:          # created and touched once, before any other stream: one hardware queue each
        LANE_STREAMS["pool"] = ops.lane_streams(6, device)
    streams = LANE_STREAMS["pool"][:lanes]
    for lane in range(lanes):
        m = model if lane == 0 else get_model(model_name)(cfg["model"])
        if lane:
            m.load_state_dict(model.state_dict())
        p = get_plugin("entmin_tta")(cfg)
        p.lane = lane
        plugs.append(p.setup(m, device))
    per_round = lanes * group
    rounds = max(2, (volumes + per_round - 1) // per_round)
    xs, ys = [], []
    for lane in range(lanes):
        vs = [synth_volume(lane * group + i, C, shape, R) for i in range(group)]
        xs.append(torch.stack([v["image"] for v in vs]).to(device))
        ys.append(torch.stack([v["label"] for v in vs]).to(device))
    counts = torch.zeros((lanes, group, R, 3), dtype=torch.int64, device=device)

    def one(lane):
        with torch.cuda.stream(streams[lane]):
            res = plugs[lane].adapt_volume(xs[lane])
            ops.mask_dice_counts(res["logits_cl"], ys[lane], 0.5, counts[lane], None)

    for lane in range(lanes):
        one(lane)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        for lane in range(lanes):
            one(lane)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del plugs, streams, xs, ys, model
    gc.collect()
    torch.cuda.empty_cache()
    return rounds * per_round / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--combos", nargs="+", default=["4x1", "1x4", "2x4", "1x8", "2x8", "3x8", "2x16"], help="LANESxGROUP")
    ap.add_argument("--knobs", default="default", choices=sorted(KNOBS))
    ap.add_argument("--volumes", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--model", default="unet")
    ap.add_argument("--task", default="brats")
    ap.add_argument("--shape", type=int, nargs=3, default=[128, 128, 128])
    a = ap.parse_args()
    for rep in range(a.repeat):
        for combo in a.combos:
            lanes, group = (int(v) for v in combo.split("x"))
            for s in KNOBS[a.knobs]:
                v = run(s, lanes, group, a.volumes, a.model, tuple(a.shape), a.task)
                print(f"pass {rep} lanes {lanes} group {group} {s}: {v:.2f} volumes/s", flush=True)


if __name__ == "__main__":
    main()
