#!/usr/bin/env python
"""3 -> 3 convolution at 128^3: conv3_mfma4_kernel (MMTTA_OPT_THIN_MFMA) against direct_row_kernel, timed from a captured
graph of 20 launches.  (The first version of the kernel had switches for its phases here: 44 us all phases, 35 without
the MFMA loop, 34 without the staging loads, 26 with neither loads, MFMAs nor stores - i.e. index arithmetic.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def timed(op, x, y, reps=20):
    """us per launch from a captured graph of `reps` launches (host launch cost out of the picture)."""
    for _ in range(3):
        op.forward(x, None, None, y)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                op.forward(x, None, None, y)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


def main():
    from multimodal_tta_amd import ops
    ops.set_option(13, 1)
    x = ops.new_cl(1, 128, 128, 128, 3, "cuda", ldc=4, zero=True)
    x.normal_()
    y = ops.new_cl(1, 128, 128, 128, 3, "cuda", ldc=4, zero=True)
    for diag, name in ((1, "conv3_mfma4_kernel"),):
        ops.set_option(6, diag)
        op = ops.ConvOp(3, 3, 3, 1, False, "cuda", dtype=ops.BF16)
        op.pack(torch.randn(3, 3, 3, 3, 3, device="cuda"))
        print(f"{name:36s} {timed(op, x, y):8.1f} us")
    ops.set_option(6, 1)
    ops.set_option(13, 0)
    op = ops.ConvOp(3, 3, 3, 1, False, "cuda", dtype=ops.BF16)
    op.pack(torch.randn(3, 3, 3, 3, 3, device="cuda"))
    print(f"{'direct_row_kernel':36s} {timed(op, x, y):8.1f} us")


if __name__ == "__main__":
    main()
