#!/usr/bin/env python3
"""Sums the counters of a rocprofv3 --pmc database over ALL dispatches of the kernels whose name contains a pattern.
usage: tools/pmc_sum.py <results.db or directory> <pattern> [label]"""
import glob, os, sqlite3, sys
from collections import defaultdict
path = sys.argv[1]
if os.path.isdir(path):
    path = max(glob.glob(os.path.join(path, "**", "*.db"), recursive=True), key=os.path.getmtime)
pat = sys.argv[2]
c = sqlite3.connect(path)
tot = defaultdict(float); dur = {}
for name, d, cn, val, du in c.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    if pat in name:
        tot[cn] += val; dur[d] = du
label = sys.argv[3] if len(sys.argv) > 3 else pat
print(f"{label}: {len(dur)} dispatches, {sum(dur.values()) / 1e3:.1f} us under the profiler")
for k in sorted(tot):
    print(f"   {k:28s} {tot[k]:.4e}")
g = lambda k: tot.get(k, 0.0)
if g("SQ_INSTS_VALU"):
    print(f"   lanes per VALU instruction   {g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU'):.1f} / 64")
if g("SQ_WAVE_CYCLES"):
    print(f"   waiting / wave cycles        {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.2f}")
    print(f"   VALU insts / wave cycle      {g('SQ_INSTS_VALU') / g('SQ_WAVE_CYCLES'):.3f}")
