#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd sqlite database, for the largest dispatch of each kernel.
usage: tools/pmc_summary.py <results.db> [title]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    c = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))
    meta = {}
    for name, disp, grid, cname, val, dur, vg in c.execute(
            "select kernel_name, dispatch_id, grid_size, counter_name, value, duration, vgpr_count from counters_collection"):
        per[(name, disp)][cname] += val
        meta[(name, disp)] = (grid, dur, vg)
    best = {}
    for (name, disp), (grid, dur, vg) in meta.items():
        if name not in best or grid > meta[(name, best[name])][0]:
            best[name] = disp
    counters = sorted({k for v in per.values() for k in v})
    print(f"# {title}\n\nlargest dispatch of each kernel; counter values summed over SEs/XCDs\n")
    print("| kernel | grid | us | vgpr | " + " | ".join(counters) + " |")
    print("|---|---|---|---|" + "---|" * len(counters))
    for name, disp in sorted(best.items(), key=lambda kv: -meta[(kv[0], kv[1])][1]):
        grid, dur, vg = meta[(name, disp)]
        short = name if len(name) < 60 else name[:57] + "..."
        print(f"| `{short}` | {grid} | {dur / 1e3:.1f} | {vg} | " + " | ".join(f"{per[(name, disp)][k]:.4g}" for k in counters) + " |")


if __name__ == "__main__":
    main()
