#!/usr/bin/env python3
"""Assembles the files committed under profiles/ from one run of tools/profile_round.sh (gpurun_out/prof/):
   usage: tools/make_profiles.py <tag>     e.g. r01_c  ->  profiles/r01_c_kernel_stats.md, _pmc.md, _traffic.json, _bench_under_rocprof.json"""
import glob
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = "python3 bench.py --no-cpu-baseline --steps 5 --warmup 2"


def db(sub):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof", sub, "**", "*.db"), recursive=True))
    if not f:
        sys.exit(f"no database under gpurun_out/prof/{sub}")
    return max(f, key=os.path.getmtime)


def run(tool, *a):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *a], check=True, capture_output=True, text=True).stdout


def largest(path, counter):
    """counter value (summed over SEs/XCDs) of the largest dispatch of each kernel"""
    c = sqlite3.connect(path)
    per, grid = defaultdict(float), {}
    for name, disp, g, cname, val in c.execute("select kernel_name, dispatch_id, grid_size, counter_name, value from counters_collection"):
        if cname == counter:
            per[(name, disp)] += val
            grid[(name, disp)] = g
    best = {}
    for (name, disp), g in grid.items():
        if name not in best or g > grid[(name, best[name])]:
            best[name] = disp
    return {name: per[(name, disp)] for name, disp in best.items()}


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "profiles")
    rel = lambda p: os.path.relpath(p, ROOT)
    open(os.path.join(out, f"{tag}_kernel_stats.md"), "w").write(
        run("prof_summary.py", db("stats"), f"{tag}: rocprofv3 --kernel-trace --stats -- {CMD}"))
    pmc = [f"# {tag}: PMC passes of the same command (one rocprofv3 --kernel-trace --pmc run per counter group)\n\n"
           "FETCH_SIZE / WRITE_SIZE are in KiB per dispatch as rocprofv3 reports them (largest dispatch of each kernel).\n"
           "On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced streams; bench.py's\n"
           "roofline.traffic doubles it as the guide prescribes: traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes.\n"]
    for sub in ("fetch", "write", "sq"):
        pmc.append(run("pmc_summary.py", db(sub), rel(db(sub))).replace("# ", "## ", 1))
    open(os.path.join(out, f"{tag}_pmc.md"), "w").write("\n".join(pmc))
    fetch, write = largest(db("fetch"), "FETCH_SIZE"), largest(db("write"), "WRITE_SIZE")
    traffic = {}
    # (the frame kernels by the instantiation the headline workload runs: fp32, culled primitive lists, no photon maps)
    pick = {"k_primary": "k_primary<float, true, false, false>", "k_tail": "k_tail<float, true, false, false, 0>", "k_tail_persistent": "k_tail_persistent<float, true, false, false>",
            "k_gather_knn_wave": "k_gather_knn_wave(", "k_gather_knn_single": "k_gather_knn_single(", "k_emit": "k_emit<float", "k_kd_lds": "k_kd_lds("}
    for key, pat in pick.items():
        f = [v for n, v in fetch.items() if pat in n]
        w = [v for n, v in write.items() if pat in n]
        if f and w:
            traffic[key] = {"fetch_size_kib": max(f), "write_size_kib": max(w), "traffic_bytes_per_launch": int((2 * max(f) + max(w)) * 1024)}
    traffic["_note"] = (f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `{CMD}` (tools/profile_round.sh); largest dispatch "
                        "of each kernel; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled as the gfx950 correction for coalesced streams")
    json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
    src = os.path.join(ROOT, "gpurun_out", "prof", "stats.json")
    if os.path.exists(src):
        open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(open(src).read())
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
