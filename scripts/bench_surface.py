"""Time mmtta_surface_distances (HD95 / ASD) per kernel and against the scipy restatement.

usage: python scripts/bench_surface.py [D H W] [R]      (default 128 128 128, R = 3)
Prints the event-timed duration of the whole call, the bytes it has to move at least, and the host time of
oracle.hd_asd for ONE region (the reference runs MONAI/scipy per region on the host)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_tta_amd import ops  # noqa: E402


def blobs(seed, shape, n=4, rmax=30):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    z, y, x = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    m = torch.zeros(shape, dtype=torch.bool)
    for _ in range(n):
        c = [float(torch.rand((), generator=g)) * s for s in shape]
        r = [4.0 + float(torch.rand((), generator=g)) * rmax for _ in range(3)]
        m |= (((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2) <= 1.0
    return m


def main():
    a = [int(v) for v in sys.argv[1:]]
    shape = tuple(a[:3]) if len(a) >= 3 else (128, 128, 128)
    R = a[3] if len(a) >= 4 else 3
    p = torch.stack([blobs(40 + r, shape) for r in range(R)])[None]
    g = torch.stack([blobs(60 + r, shape) for r in range(R)])[None]
    pm, gl = p.to(torch.uint8).cuda().contiguous(), g.float().cuda()
    for _ in range(3):
        hd, asd = ops.surface_distances(pm, gl, (1.0, 1.0, 1.0), 95.0, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        hd, asd = ops.surface_distances(pm, gl, (1.0, 1.0, 1.0), 95.0, False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    V = shape[0] * shape[1] * shape[2]
    # per mask: label 4 B + mask 1 B read, edges 2 B written; scan 2 B read + 4 B written; pass H 4 B read + 16 B
    # written; pass D reads the edge bytes (2 B) and P2 columns only where edges are
    min_bytes = R * V * (5 + 2 + 6 + 20 + 2)
    print(f"shape {shape} R={R}: {ms:.3f} ms per call, >= {min_bytes / 1e6:.1f} MB moved -> {min_bytes / ms / 1e6:.1f} GB/s;"
          f" hd {hd.cpu().tolist()} asd {asd.cpu().tolist()}")
    import oracle
    t = time.time()
    hv, av = oracle.hd_asd(p[0, 0].numpy(), g[0, 0].numpy(), (1.0, 1.0, 1.0), 95.0, False)
    dt = time.time() - t
    print(f"scipy restatement, one region on the host: {dt * 1e3:.1f} ms (hd {hv:.4f}, asd {av:.4f}); "
          f"x{R} regions = {dt * R * 1e3:.1f} ms vs {ms:.3f} ms on the GPU")


if __name__ == "__main__":
    main()
