#!/usr/bin/env python
"""How far ahead of the GPU is the host?  Enqueue time of N adapted volumes over 4 lanes (time until the last launch call
returns) against the time until the GPU has finished them, and the same with the lanes' work enqueued from one thread."""
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
    lanes, volumes = 4, 24
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
    for i in range(volumes + lanes):
        v = synth_volume(i, 4, (128, 128, 128), 3)
        vols.append(v["image"].unsqueeze(0).to(device))

    def one(i):
        lane = i % lanes
        with torch.cuda.stream(streams[lane]):
            plugs[lane].adapt_volume(vols[i])

    for i in range(lanes):
        one(i)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        per = []
        for i in range(lanes, lanes + volumes):
            a = time.perf_counter()
            one(i)
            per.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per.sort()
        print(f"rep {rep}: enqueue {1e3 * (t1 - t0) / volumes:.2f} ms/volume (median call {1e3 * per[len(per) // 2]:.2f}, min {1e3 * per[0]:.2f}, max {1e3 * per[-1]:.2f}), "
              f"finished {1e3 * (t2 - t0) / volumes:.2f} ms/volume", flush=True)


if __name__ == "__main__":
    main()
