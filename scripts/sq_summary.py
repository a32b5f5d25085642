#!/usr/bin/env python
"""Issue-port view of one adaptation step from the SQ counter passes (scripts/pmc_layers.sh -> scripts/pmc_summary.py):
per kernel and per kernel family, how busy the vector ALU, the matrix cores and the scalar ALU are, and what that means
for several volumes in flight.

    python scripts/sq_summary.py <pmc_summary.txt> [--volumes V --ms-lanes A --ms-one-lane B] > profiles/r02c_sq_counters.md

--volumes: adapted volumes the counted run processed (scripts/pmc_bench.sh counts `bench.py --lanes 1`: whole volumes,
10 steps + final forward each); --ms-lanes / --ms-one-lane: measured ms per volume of the unprofiled bench with
method.lanes volumes in flight / with one.

Units (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (x4 = cycles), SQ_VALU_MFMA_BUSY_CYCLES
counts cycles; every counter is summed over the chip's 1024 SIMDs.  Clock taken as 2.0 GHz (under load 1.9-2.1).
"""
import sys

SIMDS, GHZ = 1024, 2.0


def fam(k):
    if k.startswith("igemm"):
        return "implicit GEMM (fwd / dgrad)"
    if k.startswith(("wgrad_bf16", "wgrad_tr")):
        return "27-tap weight gradient (MFMA)"
    if k.startswith("wgrad"):
        return "other weight gradients + their reductions"
    if k.startswith(("direct", "chan_mfma", "upconv")):
        return "thin full-resolution layers (<= 4 channels on one side)"
    if k.startswith(("splitk", "instance_stats", "channel_reduce", "stats_", "bwd_finalize", "rows_reduce")):
        return "split-K finalize, statistics, reductions"
    if k.startswith(("elementwise", "norm_bwd_apply8", "combine8")):
        return "norm backward / residual adds (elementwise)"
    return "optimizer, repack, loss, copies"


def main():
    path = sys.argv[1]
    def opt(name):
        return float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else None
    volumes, ms_lanes, ms_one = opt("--volumes"), opt("--ms-lanes"), opt("--ms-one-lane")
    rows = [line.rstrip("\n").split(" | ") for line in open(path)]
    hdr = rows[0]
    ix = {h: i for i, h in enumerate(hdr)}

    def val(r, name):
        try:
            return float(r[ix[name]])
        except (ValueError, KeyError, IndexError):
            return 0.0

    per, fams = [], {}
    tot = dict(valu=0.0, mfma=0.0, salu=0.0, wave=0.0, wait=0.0, us=0.0)
    for r in rows[1:]:
        if len(r) < len(hdr):
            continue
        calls, us = int(r[1]), float(r[2])
        d = dict(k=r[0], calls=calls, us=us, valu=val(r, "SQ_ACTIVE_INST_VALU") * 4 * calls, mfma=val(r, "SQ_VALU_MFMA_BUSY_CYCLES") * calls,
                 salu=val(r, "SQ_ACTIVE_INST_SCA") * 4 * calls, wave=val(r, "SQ_WAVE_CYCLES") * 4 * calls,
                 wait=val(r, "SQ_WAIT_ANY") * 4 * calls, nv=val(r, "SQ_INSTS_VALU") * calls, nm=val(r, "SQ_INSTS_MFMA") * calls,
                 conf=val(r, "SQ_LDS_BANK_CONFLICT") * calls, tus=us * calls)
        per.append(d)
        f = fams.setdefault(fam(d["k"]), dict(valu=0.0, mfma=0.0, salu=0.0, tus=0.0))
        for key in ("valu", "mfma", "salu", "tus"):
            f[key] += d[key]
        for key in ("valu", "mfma", "salu", "wave", "wait"):
            tot[key] += d[key]
        tot["us"] += d["tus"]
    cyc = lambda us: us * 1e-6 * GHZ * 1e9 * SIMDS
    print("# SQ counters of the adaptation path (unet 4x128^3, bf16 operands, kernels serialised by the PMC passes)\n")
    print(f"source: `{path}` (scripts/pmc_bench.sh or scripts/pmc_layers.sh: one counter group per pass, kernel trace only); {SIMDS} SIMDs, {GHZ} GHz assumed.\n")
    print("| kernel family | share of VALU-busy cycles | kernel time (us) | VALU busy | MFMA busy | SALU busy |")
    print("|---|---:|---:|---:|---:|---:|")
    for name, f in sorted(fams.items(), key=lambda kv: -kv[1]["valu"]):
        c = cyc(f["tus"])
        print(f"| {name} | {100 * f['valu'] / tot['valu']:.1f} % | {f['tus']:.0f} | {100 * f['valu'] / c:.1f} % | {100 * f['mfma'] / c:.1f} % | {100 * f['salu'] / c:.1f} % |")
    c = cyc(tot["us"])
    print(f"| **all kernels** | 100 % | {tot['us']:.0f} | {100 * tot['valu'] / c:.1f} % | {100 * tot['mfma'] / c:.1f} % | {100 * tot['salu'] / c:.1f} % |")
    print(f"\nWaves resident per SIMD (SQ_WAVE_CYCLES / SIMD-cycles): {tot['wave'] / c:.2f}; of a wave's life {100 * tot['wait'] / tot['wave']:.0f} % is SQ_WAIT_ANY.\n")
    print(f"VALU-busy SIMD-cycles of this pass: {tot['valu']:.3g} -> {tot['valu'] / SIMDS / GHZ / 1e3:.0f} us if every SIMD issued a vector instruction every "
          f"cycle it could; MFMA-busy: {tot['mfma']:.3g} -> {tot['mfma'] / SIMDS / GHZ / 1e3:.0f} us.  Serialised kernel time of the same pass: {tot['us']:.0f} us.")
    if volumes:
        v_ms, m_ms, s_ms = (tot[k] / volumes / SIMDS / GHZ / 1e6 for k in ("valu", "mfma", "salu"))
        print(f"\nPer adapted volume ({volumes:.0f} volumes counted): vector-ALU time {v_ms:.2f} ms, matrix-core time {m_ms:.2f} ms, scalar-ALU time "
              f"{s_ms:.2f} ms (SIMD-cycles / {SIMDS} / {GHZ} GHz); serialised kernel time under the counters {tot['us'] / volumes / 1e3:.1f} ms.")
        for label, ms in (("one volume in flight", ms_one), ("method.lanes volumes in flight", ms_lanes)):
            if ms:
                print(f"* {label}: {ms:.2f} ms per volume measured -> vector ALUs {100 * v_ms / ms:.0f} % busy chip-wide, matrix cores "
                      f"{100 * m_ms / ms:.0f} %, scalar ALUs {100 * s_ms / ms:.0f} %.")
    print("\n| kernel | calls | avg us | VALU busy | MFMA busy | waves / SIMD | wait | VALU instr per MFMA instr | LDS bank-conflict cycles / launch |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for d in sorted(per, key=lambda d: -d["valu"])[:28]:
        c = cyc(d["tus"])
        ratio = f"{d['nv'] / d['nm']:.1f}" if d["nm"] > 0 else "-"
        print(f"| `{d['k']}` | {d['calls']} | {d['us']:.1f} | {100 * d['valu'] / c:.1f} % | {100 * d['mfma'] / c:.1f} % | {d['wave'] / c:.2f} | "
              f"{100 * d['wait'] / max(d['wave'], 1):.0f} % | {ratio} | {d['conf'] / max(d['calls'], 1):.3g} |")


if __name__ == "__main__":
    main()
