#!/usr/bin/env python
"""Where one volume alone spends its time: kernel time against the gaps between kernels, over the last `--ms` of a
rocprofv3 kernel trace of `bench.py --lanes 1 --group 1` (rocpd database).

    python scripts/experiments/one_volume_gaps.py <dir>/run_results.db --ms 60"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).replace("mmtta::", "")


ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--ms", type=float, default=60.0)
a = ap.parse_args()
db = sqlite3.connect(a.db)
rows = list(db.execute("select name, start, end from kernels order by start"))
t1 = rows[-1][2]
rows = [r for r in rows if r[1] >= t1 - int(a.ms * 1e6)]
events = sorted([(s, 1) for _, s, _ in rows] + [(e, -1) for _, _, e in rows])
depth, last, hist = 0, events[0][0], defaultdict(int)
for t, d in events:
    hist[min(depth, 3)] += t - last
    depth += d
    last = t
wall = events[-1][0] - events[0][0]
print(f"window {wall / 1e6:.2f} ms, {len(rows)} kernels, sum of kernel time {sum(e - s for _, s, e in rows) / 1e6:.2f} ms")
print("in flight: " + "  ".join(f"{k}: {100.0 * v / wall:.1f}%" for k, v in sorted(hist.items())))
# gaps of the merged timeline by the kernel that FOLLOWS the gap
gap_by = defaultdict(lambda: [0, 0])
end = rows[0][2]
for name, s, e in rows[1:]:
    if s > end:
        g = gap_by[short(name)]
        g[0] += 1
        g[1] += s - end
    end = max(end, e)
print("\n| kernel after the gap | gaps | total gap ms | avg gap us |\n|---|---:|---:|---:|")
for k, (n, ns) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"| `{k}` | {n} | {ns / 1e6:.3f} | {ns / n / 1e3:.2f} |")
agg = defaultdict(lambda: [0, 0])
for name, s, e in rows:
    k = agg[short(name)]
    k[0] += 1
    k[1] += e - s
print("\n| kernel | calls | total ms | avg us |\n|---|---:|---:|---:|")
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| `{k}` | {n} | {ns / 1e6:.3f} | {ns / n / 1e3:.2f} |")
