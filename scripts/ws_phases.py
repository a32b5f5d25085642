#!/usr/bin/env python
"""Phase stamps of the producer/consumer implicit GEMM for one convolution (diagnostic; s_memtime ticks = shader cycles).

    python scripts/ws_phases.py --cin 32 --cout 32 --size 64
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--wgs", type=int, default=256)
    ap.add_argument("--knob", type=int, default=1, help="MMTTA_OPT_IGEMM_PIPELINE value: +2 no stores, +4 no epilogue, +8 no MFMA")
    a = ap.parse_args()
    from multimodal_tta_amd import _lib, ops
    lib = C.CDLL(_lib.LIB_PATH)
    _lib.load()
    ops.set_option(8, a.wgs)
    ops.set_option(6, a.knob)
    op = ops.ConvOp(a.cin, a.cout, 3, 1, False, "cuda", dtype=ops.BF16)
    w = torch.randn(a.cout, a.cin, 3, 3, 3, device="cuda") * 0.05
    op.pack(w)
    x = ops.new_cl(1, a.size, a.size, a.size, a.cin, "cuda")
    x.normal_()
    y = ops.new_cl(1, a.size, a.size, a.size, a.cout, "cuda")
    bias = torch.zeros(a.cout, device="cuda")
    for _ in range(3):
        op.forward(x, None, bias, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        op.forward(x, None, bias, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"conv {a.cin}->{a.cout} @ {a.size}^3: {e0.elapsed_time(e1) * 100:.1f} us per call ({a.wgs} workgroups)")
    dbg = torch.zeros(a.wgs * 8, dtype=torch.int64, device="cuda")
    lib.mmtta_debug_set_buffer.argtypes = [C.c_void_p]
    lib.mmtta_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
    op.forward(x, None, bias, y)
    torch.cuda.synchronize()
    lib.mmtta_debug_set_buffer(None)
    d = dbg.view(a.wgs, 8).double().cpu()
    names = ["loader staging", "loader barrier wait", "steps", "consumer barrier wait", "consumer MFMA phases", "consumer epilogue"]
    for i, nme in enumerate(names):
        col = d[:, i]
        print(f"  {nme:24s} median {col.median().item():10.0f}  min {col.min().item():10.0f}  max {col.max().item():10.0f}   (cycles per workgroup)")


if __name__ == "__main__":
    main()
