#!/usr/bin/env python
"""Do the four lanes run in lock-step?  Throughput of N volumes over 4 lanes when every lane starts its first volume at the
same time against lanes started a quarter of a volume apart (a GPU-side spin on the lane's stream before its first volume)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import multimodal_tta_amd  # noqa: F401
    from multimodal_tta_amd import ops
    from multimodal_tta_amd.config import compose
    from multimodal_tta_amd.registry import get_model, get_plugin
    from multimodal_tta_amd.synth import synth_volume
    lanes, volumes = 4, 48
    device = torch.device("cuda", 0)
    cfg = compose(overrides=["task=brats", "dataset=brats", "model=unet", "method=tta_entmin", "method.steps=10",
                             "method.precision=bf16"])
    torch.manual_seed(42)
    model = get_model("unet")(cfg["model"])
    streams = ops.lane_streams(6, device)[:lanes]
    plugs = []
    for lane in range(lanes):
        m = model if lane == 0 else get_model("unet")(cfg["model"])
        if lane:
            m.load_state_dict(model.state_dict())
        p = get_plugin("entmin_tta")(cfg)
        p.lane = lane
        plugs.append(p.setup(m, device))
    vols = []
    for i in range(8):
        v = synth_volume(i, 4, (128, 128, 128), 3)
        vols.append(v["image"].unsqueeze(0).to(device))

    def one(i):
        lane = i % lanes
        with torch.cuda.stream(streams[lane]):
            plugs[lane].adapt_volume(vols[i % len(vols)])

    for i in range(lanes):
        one(i)
    torch.cuda.synchronize()
    # cycles of one lane-volume (about 60 ms at ~2.1 GHz of the spin kernel's clock)
    for rep in range(2):
        for frac in (0.0, 0.25, 0.125):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for lane in range(lanes):
                if frac > 0 and lane > 0:
                    with torch.cuda.stream(streams[lane]):
                        torch.cuda._sleep(int(frac * lane * 0.060 * 2.1e9))
            for i in range(volumes):
                one(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"rep {rep} stagger {frac:5.3f} of a lane-volume per lane: {volumes / dt:6.2f} volumes/s ({1e3 * dt / volumes:.2f} ms per volume, "
                  f"stagger included)", flush=True)


if __name__ == "__main__":
    main()
