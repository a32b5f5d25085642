#!/usr/bin/env python
"""HBM roofline of the input pre-pass and the DiceCE kernels (algorithmic bytes / measured time, events on the stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multimodal_tta_amd import ops  # noqa: E402
from multimodal_tta_amd.transforms import normalize_image  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    pol = {"enabled": True, "channels": {str(c): {"clip": [-3.0, 3.0], "zscore": {"masked": True, "mask_gt": -2.5}} for c in range(4)}}
    for shape in [(4, 128, 128, 128), (2, 48, 144, 144), (4, 160, 192, 160)]:
        x = torch.randn(shape, device="cuda")
        nb = x.numel() * 4
        t = timed(lambda: normalize_image(x, intensity_policy=pol))
        print(f"intensity policy {shape}: {t * 1e6:8.1f} us  {3 * nb / t / 1e9:8.1f} GB/s (12 B per voxel and channel; includes the scratch allocation of the wrapper)")
        t = timed(lambda: normalize_image(x, mean=[0.1] * shape[0], std=[2.0] * shape[0]))
        print(f"mean/std         {shape}: {t * 1e6:8.1f} us  {2 * nb / t / 1e9:8.1f} GB/s (8 B per voxel and channel)")
    for R, shp in [(3, (128, 128, 128)), (1, (48, 144, 144))]:
        z = ops.new_cl(1, *shp, R, "cuda", (R + 3) // 4 * 4)
        z.normal_()
        dz = ops.new_cl(1, *shp, R, "cuda", (R + 3) // 4 * 4)
        y = (torch.rand(1, R, *shp, device="cuda") > 0.7).float()
        sums = torch.zeros(R * 3 + 1, dtype=torch.float64, device="cuda")
        nvox = shp[0] * shp[1] * shp[2]
        t1 = timed(lambda: ops.dice_ce_sums(z, y, None, False, sums))
        t2 = timed(lambda: ops.dice_ce_grad(z, y, None, False, False, True, 1.0, 1.0, sums, dz))
        print(f"dice_ce sums R={R} {shp}: {t1 * 1e6:8.1f} us  {8 * R * nvox / t1 / 1e9:8.1f} GB/s ; grad {t2 * 1e6:8.1f} us  {12 * R * nvox / t2 / 1e9:8.1f} GB/s")


if __name__ == "__main__":
    main()
