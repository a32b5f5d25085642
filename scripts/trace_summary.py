#!/usr/bin/env python
"""Aggregate a rocprofv3 ``*_kernel_trace.csv`` into a per-kernel table (calls, total, average, share).

    python scripts/trace_summary.py gpurun_out/<tag>/prof/run_results.db > profiles/rNN_kernels.md
    python scripts/trace_summary.py <dir>/<pid>_kernel_trace.csv

rocprofv3's own ``--stats`` CSV is kept beside it when present; this table exists because that file
has been seen to cover only part of a long run, and because template arguments matter here (the
same kernel template is several kernels)."""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mmtta::", "")


def rows(path: str):
    """(kernel name, start ns, end ns) from a rocprofv3 kernel trace: CSV, or the rocpd SQLite database that
    ROCm 7 writes by default (its `kernels` view)."""
    if path.endswith(".db"):
        import sqlite3
        db = sqlite3.connect(path)
        yield from db.execute("select name, start, end from kernels")
    else:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])


def main(path: str, as_json: bool = False) -> None:
    tot = defaultdict(lambda: [0, 0])
    t_min, t_max = None, None
    for name, s, e in rows(path):
        k = tot[short(name)]
        k[0] += 1
        k[1] += e - s
        t_min = s if t_min is None else min(t_min, s)
        t_max = e if t_max is None else max(t_max, e)
    busy = sum(v[1] for v in tot.values())
    if as_json:      # per-kernel totals for bench.py's roofline block (profiles/trace_avg_us.json)
        import json
        print(json.dumps({"source": path, "kernels": {name: {"calls": n, "total_us": ns / 1e3, "avg_us": ns / n / 1e3}
                                                      for name, (n, ns) in sorted(tot.items())}}, indent=1))
        return
    print(f"kernels: {sum(v[0] for v in tot.values())}   GPU busy: {busy / 1e6:.2f} ms   "
          f"first start -> last end: {(t_max - t_min) / 1e6:.2f} ms\n")
    print("| kernel | calls | total ms | avg us | share |")
    print("|---|---:|---:|---:|---:|")
    for name, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{name}` | {n} | {ns / 1e6:.3f} | {ns / n / 1e3:.2f} | {100.0 * ns / busy:.1f}% |")


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if a != "--json"][0], "--json" in sys.argv)
