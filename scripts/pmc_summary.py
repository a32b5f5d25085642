#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one database per counter pass).

    python scripts/pmc_summary.py gpurun_out/pmc_lt/p1/run_results.db gpurun_out/pmc_lt/p2/run_results.db ...
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mmtta::", "")


def main(paths):
    vals = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over dispatches (and over SE instances)
    calls = defaultdict(lambda: defaultdict(set))
    dur = defaultdict(lambda: [0, 0])
    for p in paths:
        db = sqlite3.connect(p)
        seen = set()
        for name, cname, value, disp, d in db.execute(
                "select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
            k = short(name)
            vals[k][cname] += value
            calls[k][cname].add(disp)
            if (p, disp) not in seen:
                seen.add((p, disp))
                dur[k][0] += 1
                dur[k][1] += d
    counters = sorted({c for k in vals for c in vals[k]})
    print("kernel | calls | avg us (under PMC) | " + " | ".join(counters))
    for k in sorted(vals, key=lambda k: -dur[k][1]):
        row = [k, str(dur[k][0] // max(1, len(paths))), f"{dur[k][1] / dur[k][0] / 1e3:.1f}"]
        for c in counters:
            n = len(calls[k][c])
            row.append(f"{vals[k][c] / n:.4g}" if n else "-")
        print(" | ".join(row))


if __name__ == "__main__":
    main(sys.argv[1:])
