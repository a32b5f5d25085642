#!/usr/bin/env python
"""Chip-equivalent time per adapted volume from a timeline summary (scripts/trace_timeline.py): every (kernel, workgroups)
row costs  average duration x launches x min(1, workgroups / slots)  where `slots` is how many of its workgroups the chip
holds at once (512 for the register-heavy convolution kernels: 2 per CU; 2048 for the light streaming kernels).  With
several volumes in flight (one hardware queue each) small launches overlap and only this product is paid: the sum
reproduces the measured milliseconds per volume and says which kernels the throughput is spent on.

    python scripts/chip_equivalent.py profiles/r02i_timeline.md --volumes 12 > profiles/r02i_chip_equivalent.md
"""
import re
import sys

FAT = ("igemm", "wgrad_tr", "wgrad_bf16", "wgrad_small", "wgrad_f32", "upconv", "chan_mfma", "conv3_mfma4")


def family(k):
    if k.startswith("igemm_cls8"):
        return "class-fused stride-2 forms"
    if k.startswith("igemm"):
        return "implicit GEMM (fwd / dgrad)"
    if k.startswith(("wgrad_tr", "wgrad_bf16")):
        return "27-tap / 1x1x1 weight gradient (MFMA)"
    if k.startswith("wgrad"):
        return "thin-layer weight gradients + slab reductions"
    if k.startswith(("direct", "chan_mfma", "upconv", "conv3_mfma4")):
        return "thin full-resolution layers"
    if k.startswith(("splitk", "instance", "channel_reduce", "stats", "db_reduce", "slab")):
        return "split-K finalize, statistics, reductions"
    if k.startswith(("elementwise", "norm_bwd_apply8", "combine8")):
        return "norm backward / residual adds"
    return "optimizer, repack, loss, copies"


def main():
    path = sys.argv[1]
    vols = float(sys.argv[sys.argv.index("--volumes") + 1]) if "--volumes" in sys.argv else 12.0
    rows = []
    for line in open(path):
        m = re.match(r"\| `(.+)` \| (\d+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5))))
    tot, fam, out = 0.0, {}, []
    for name, wgs, calls, ms, us in rows:
        slots = 512 if name.startswith(FAT) else 2048
        f = min(1.0, wgs / slots)
        ce = ms * f / vols
        tot += ce
        fam[family(name)] = fam.get(family(name), 0.0) + ce
        out.append((ce, name, wgs, calls / vols, us, f))
    busy = sum(r[3] for r in rows) / vols
    print(f"# Chip-equivalent time per adapted volume ({path}, {vols:.0f} volumes in the trace)\n")
    print(f"sum of kernel time {busy:.1f} ms per volume (what one lane pays); chip-equivalent {tot:.1f} ms per volume (what method.lanes "
          "volumes in flight pay; compare `ms_per_step` of the bench line of the same build)\n")
    print("| kernel family | ms per volume | share |\n|---|---:|---:|")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {v:.2f} | {100 * v / tot:.1f} % |")
    print("\n| kernel | workgroups | launches per volume | avg us | fraction of the chip | ms per volume |\n|---|---:|---:|---:|---:|---:|")
    for ce, name, wgs, c, us, f in sorted(out, reverse=True)[:30]:
        print(f"| `{name}` | {wgs} | {c:.1f} | {us:.1f} | {f:.2f} | {ce:.3f} |")


if __name__ == "__main__":
    main()
