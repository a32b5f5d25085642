#!/usr/bin/env python
"""Round-3 probe: how much would launching B volumes as ONE batch per kernel buy (timing only)?

The existing kernels already take a batch dimension (weights shared over the batch, which is NOT per-volume
adaptation: the numbers of this script are timings of the launch shape, not results).  Prints volumes/s for
(lanes, batch) combinations and split-K settings, same workload as bench.py (unet 4x128^3, S=10, bf16)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import multimodal_tta_amd  # noqa: F401,E402
from multimodal_tta_amd import ops  # noqa: E402
from multimodal_tta_amd.config import compose  # noqa: E402
from multimodal_tta_amd.registry import get_model, get_plugin  # noqa: E402
from multimodal_tta_amd.synth import synth_volume  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = compose(overrides=["task=brats", "dataset=brats", "model=unet", "method=tta_entmin", "method.steps=10",
                             "method.precision=bf16"])
    combos = [(4, 1), (1, 4), (2, 4), (1, 8), (2, 8), (1, 16), (2, 2)]
    if len(sys.argv) > 1:
        combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    streams = ops.lane_streams(4, dev)
    vol = synth_volume(0, 4, (128, 128, 128), 3)["image"].unsqueeze(0).to(dev)
    lane0 = 0
    for lanes, batch in combos:
        for below, target in ((96, 128), (256, 512)):
            ops.set_option(2, below)
            ops.set_option(3, target)
            torch.manual_seed(42)
            plugs = []
            for lane in range(lanes):
                m = get_model("unet")(cfg["model"])
                p = get_plugin("entmin_tta")(cfg)
                p.lane = lane0
                lane0 += 1
                plugs.append(p.setup(m, dev))
            x = vol.expand(batch, -1, -1, -1, -1).contiguous()
            rounds = max(2, 24 // (lanes * batch))
            for lane in range(lanes):
                with torch.cuda.stream(streams[lane]):
                    plugs[lane].adapt_volume(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                for lane in range(lanes):
                    with torch.cuda.stream(streams[lane]):
                        plugs[lane].adapt_volume(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nv = rounds * lanes * batch
            print(f"lanes {lanes} batch {batch} splitk {below}/{target}: {nv / dt:7.2f} volumes/s  ({1000 * dt / nv:.2f} ms/volume)",
                  flush=True)
            del plugs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
