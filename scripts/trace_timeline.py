#!/usr/bin/env python
"""Timeline view of a rocprofv3 kernel trace (rocpd database): per (kernel, grid) rows with calls, average duration,
workgroups, and how the wall time of the traced window splits into time with 0 / 1 / 2+ kernels in flight.

    python scripts/trace_timeline.py <dir>/run_results.db [--skip-first-ms 0] > profiles/rNN_timeline.md

Used to decide what bounds a lane: a kernel that fills the chip costs throughput, a 6 us kernel with 64 workgroups
costs only latency (another lane's kernels run beside it)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mmtta::", "")


def main(path, skip_ms=0.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gcol = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
    wcol = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
    sel = "name, start, end" + (", " + gcol[0] if gcol else ", 0") + (", " + wcol[0] if wcol else ", 1")
    extra = [c for c in ("grid_y", "grid_size_y") if c in cols] + [c for c in ("grid_z", "grid_size_z") if c in cols]
    if len(extra) == 2:
        sel += ", " + extra[0] + ", " + extra[1]
    rows = list(db.execute(f"select {sel} from kernels order by start"))
    t0 = rows[0][1] + int(skip_ms * 1e6)
    rows = [r for r in rows if r[1] >= t0]
    agg = defaultdict(lambda: [0, 0])
    events = []
    for r in rows:
        name, s, e, gx, wx = r[:5]
        gy, gz = (r[5], r[6]) if len(r) > 5 else (1, 1)
        wgs = max(1, (gx or 1) // max(1, wx or 1)) * max(1, gy or 1) * max(1, gz or 1) if wx and gx and gx >= wx else (gx or 0) * (gy or 1) * (gz or 1)
        k = agg[(short(name), wgs)]
        k[0] += 1
        k[1] += e - s
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, last, hist = 0, events[0][0], defaultdict(int)
    for t, d in events:
        hist[min(depth, 3)] += t - last
        depth += d
        last = t
    wall = events[-1][0] - events[0][0]
    busy = sum(v[1] for v in agg.values())
    print(f"kernels {len(rows)}  wall {wall / 1e6:.2f} ms  sum of kernel time {busy / 1e6:.2f} ms")
    print("in flight: " + "  ".join(f"{k}{'+' if k == 3 else ''}: {100.0 * v / wall:.1f}%" for k, v in sorted(hist.items())))
    print("\n| kernel | workgroups | calls | total ms | avg us | share of kernel time |")
    print("|---|---:|---:|---:|---:|---:|")
    for (name, wgs), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"| `{name}` | {wgs} | {n} | {ns / 1e6:.3f} | {ns / n / 1e3:.2f} | {100.0 * ns / busy:.1f}% |")


if __name__ == "__main__":
    skip = 0.0
    if "--skip-first-ms" in sys.argv:
        i = sys.argv.index("--skip-first-ms")
        skip = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    main(sys.argv[1], skip)
