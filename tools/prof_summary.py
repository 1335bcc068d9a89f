#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) result database into the per-kernel statistics table that is
committed under profiles/.   usage: tools/prof_summary.py <results.db> [title] > profiles/<name>.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("rocprofv3 --kernel-trace --stats; durations in microseconds\n")
    print("| kernel | calls | total us | avg us | min us | max us | % | max grid | wg | vgpr | sgpr | lds B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e3:.1f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
              f"{100 * r[2] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main()
