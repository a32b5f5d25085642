#!/usr/bin/env python
"""Timing ablations of the transposed-read weight gradient (MMTTA_OPT_IGEMM_PIPELINE bits 1-3 switch phases off; the
results of those runs are invalid, only the times mean something).

    python scripts/wgrad_phases.py [cin cout stride size]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from multimodal_tta_amd import ops
    cin, cout, stride, size = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 32, 1, 64)))
    ops.set_option(11, int(os.environ.get("WG", "3")))
    x = torch.randn(1, size, size, size, cin, device="cuda").to(torch.bfloat16)
    so = size // stride
    gy = torch.randn(1, so, so, so, cout, device="cuda")
    mu = torch.zeros(cin, device="cuda"); rs = torch.ones(cin, device="cuda")
    nl = ops.NL(mu, rs, relu=True)
    for diag, name in ((1, "all phases"), (3, "no MFMA loop"), (5, "no box commit"), (9, "no loads"), (7, "loads only (+dense commit)"),
                       (13, "MFMA + dense commit only"), (15, "dense commit only")):
        ops.set_option(6, diag)
        op = ops.ConvOp(cin, cout, 3, stride, False, "cuda", dtype=ops.BF16)
        op.pack(torch.randn(cout, cin, 3, 3, 3, device="cuda"))
        dw = torch.empty(cout, cin, 3, 3, 3, device="cuda")
        db = torch.empty(cout, device="cuda")
        for _ in range(3):
            op.wgrad(x, nl, gy, dw, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            op.wgrad(x, nl, gy, dw, db)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:32s} {e0.elapsed_time(e1) / 20 * 1000:8.1f} us per wgrad (main kernel + reduces)")
    ops.set_option(6, 1)


if __name__ == "__main__":
    main()
