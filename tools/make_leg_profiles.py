#!/usr/bin/env python3
"""Per-leg profile tables for profiles/ from the runs of tools/profile_legs.sh (gpurun_out/prof/<leg>/<pass>/).
   usage: tools/make_leg_profiles.py <tag> <leg> [<leg> ...]      e.g.  r03 frames gather emit

For every leg:  profiles/<tag>_<leg>_kernel_stats.md   kernel-time table of that leg's command alone
                profiles/<tag>_<leg>_pmc.md            the counter passes, per kernel (sums over its dispatches and the
                                                       dispatches of the largest grid, which are the ones a roofline quotes)
and, merged over the legs given (entries of other legs already in the file are kept):
                profiles/<tag>_counters_by_leg.json    what bench.py reads for roofline.traffic and roofline.executed

executed flop of a kernel = (2 FMA + MUL + ADD + TRANS wave-instructions) x the mean number of active lanes of a VALU
instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU), separately for fp32 and fp64.
traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (FETCH_SIZE doubled: on gfx950 it tallies 128-byte requests of wide
coalesced streams at 64 bytes)."""
import glob
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOP32 = ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")
FLOP64 = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")


def db(leg, sub):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", "prof", leg, sub, "**", "*.db"), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def counters(path):
    """{kernel: {dispatch: {"grid": g, counter: value summed over the chip}}}"""
    out = defaultdict(lambda: defaultdict(dict))
    if not path:
        return out
    c = sqlite3.connect(path)
    for name, disp, g, cname, val in c.execute("select kernel_name, dispatch_id, grid_size, counter_name, value from counters_collection"):
        d = out[name][disp]
        d["grid"] = g
        d[cname] = d.get(cname, 0.0) + val
    return out


def durations(path):
    """{kernel: [(grid, duration_ns), ...]} of the kernel-time pass"""
    out = defaultdict(list)
    c = sqlite3.connect(path)
    for name, g, dur in c.execute("select name, grid_x * grid_y * grid_z, duration from kernels"):
        out[name].append((g, dur))
    return out


def fold(dispatches, quoted_only):
    """sums of every counter over the dispatches (all, or those of the largest grid) and their number"""
    if not dispatches:
        return {}, 0
    gmax = max(d["grid"] for d in dispatches.values())
    pick = [d for d in dispatches.values() if (not quoted_only or d["grid"] == gmax)]
    tot = defaultdict(float)
    for d in pick:
        for k, v in d.items():
            if k != "grid":
                tot[k] += v
    return tot, len(pick)


def describe(kernel, passes, durs, quoted_only):
    """one record: counters per launch, lanes, executed flop, traffic, duration of the kernel-time pass"""
    rec = {}
    sums = {}
    n = None
    for sub, table in passes.items():
        tot, cnt = fold(table.get(kernel, {}), quoted_only)
        if cnt:
            sums.update({k: v / cnt for k, v in tot.items()})
            n = cnt if n is None else n
    if not sums:
        return None
    rec["dispatches"] = n
    rec["per_launch"] = {k: sums[k] for k in sorted(sums)}
    iv, tc = sums.get("SQ_INSTS_VALU"), sums.get("SQ_THREAD_CYCLES_VALU")
    lanes = tc / iv if iv and tc else None
    rec["lanes_per_valu_instruction"] = lanes
    if sums.get("SQ_WAVE_CYCLES"):
        rec["waiting_over_wave_cycles"] = sums.get("SQ_WAIT_ANY", 0.0) / sums["SQ_WAVE_CYCLES"]
    w = lambda names: 2 * sums.get(names[0], 0.0) + sums.get(names[1], 0.0) + sums.get(names[2], 0.0) + sums.get(names[3], 0.0)
    if lanes and any(k in sums for k in FLOP32 + FLOP64):
        rec["executed_flop_f32_per_launch"] = w(FLOP32) * lanes
        rec["executed_flop_f64_per_launch"] = w(FLOP64) * lanes
        if iv:
            rec["flop_instruction_share_of_valu"] = (sum(sums.get(k, 0.0) for k in FLOP32 + FLOP64)) / iv
    if "FETCH_SIZE" in sums and "WRITE_SIZE" in sums:
        rec["traffic_bytes_per_launch"] = (2 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024
    dl = durs.get(kernel, [])
    if dl:
        gmax = max(g for g, _ in dl)
        pick = [d for g, d in dl if (not quoted_only or g == gmax)]
        rec["kernel_time_pass"] = {"launches": len(pick), "avg_us": sum(pick) / len(pick) / 1e3, "total_us": sum(pick) / 1e3}
        t = sum(pick) / len(pick) * 1e-9
        if "executed_flop_f32_per_launch" in rec and t > 0:
            rec["executed_TFLOPs_f32_in_that_pass"] = rec["executed_flop_f32_per_launch"] / t / 1e12
            rec["executed_TFLOPs_f64_in_that_pass"] = rec["executed_flop_f64_per_launch"] / t / 1e12
    return rec


def main():
    tag, legs = sys.argv[1], sys.argv[2:]
    committed = os.path.join(ROOT, "profiles")
    out = os.environ.get("PROFILES_OUT", committed)          # (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)
    os.makedirs(out, exist_ok=True)
    merged_path = os.path.join(out, f"{tag}_counters_by_leg.json")
    prior = merged_path if os.path.exists(merged_path) else os.path.join(committed, f"{tag}_counters_by_leg.json")
    merged = json.load(open(prior)) if os.path.exists(prior) else {}
    for leg in legs:
        L = "none" if leg == "frames" else leg
        cmd = f"python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --legs {L}"
        stats = db(leg, "stats")
        if not stats:
            print(f"{leg}: no kernel-time database, skipped")
            continue
        table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), stats, f"{tag}, leg '{leg}': rocprofv3 --kernel-trace --stats -- {cmd}"],
                               check=True, capture_output=True, text=True).stdout
        open(os.path.join(out, f"{tag}_{leg}_kernel_stats.md"), "w").write(table)
        durs = durations(stats)
        passes = {sub: counters(db(leg, sub)) for sub in ("fetch", "write", "sq", "flops")}
        total = sum(d for v in durs.values() for _, d in v) or 1
        kernels = sorted(durs, key=lambda k: -sum(d for _, d in durs[k]))
        md = [f"# {tag}, leg '{leg}': counter passes of `{cmd}`\n",
              "One rocprofv3 --kernel-trace --pmc run per counter group (FETCH_SIZE; WRITE_SIZE; SQ occupancy set; SQ flop-class set), the",
              "kernel times from the --stats run of the same command.  Values are PER LAUNCH (mean over the dispatches named), summed over",
              "the chip.  'largest grid' = the dispatches with the kernel's largest grid in this leg: the ones a roofline figure quotes.",
              "lanes = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU; executed flop = (2 FMA + MUL + ADD + TRANS) x lanes; traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB.\n"]
        leg_json = {"_command": cmd}
        for k in kernels:
            share = sum(d for _, d in durs[k]) / total
            if share < 0.005:
                continue
            every, quoted = describe(k, passes, durs, False), describe(k, passes, durs, True)
            if not every:
                continue
            leg_json[k] = {"share_of_leg_kernel_time": share, "all_dispatches": every, "largest_grid_dispatches": quoted}
            md.append(f"## `{k if len(k) < 110 else k[:107] + '...'}`  ({100 * share:.1f} % of the leg's kernel time)\n")
            md.append("| | all dispatches | largest grid |")
            md.append("|---|---|---|")
            rows = [("dispatches (counter passes)", every["dispatches"], quoted["dispatches"])]
            kt = lambda r, f: r.get("kernel_time_pass", {}).get(f)
            rows.append(("launches / avg us (kernel-time pass)", f"{kt(every, 'launches')} / {kt(every, 'avg_us'):.2f}", f"{kt(quoted, 'launches')} / {kt(quoted, 'avg_us'):.2f}"))
            for key in sorted(set(every["per_launch"]) | set(quoted["per_launch"])):
                rows.append((key, f"{every['per_launch'].get(key, float('nan')):.4e}", f"{quoted['per_launch'].get(key, float('nan')):.4e}"))
            for key, fmt in (("lanes_per_valu_instruction", "{:.1f}"), ("waiting_over_wave_cycles", "{:.2f}"), ("flop_instruction_share_of_valu", "{:.2f}"),
                             ("executed_flop_f32_per_launch", "{:.4e}"), ("executed_flop_f64_per_launch", "{:.4e}"), ("traffic_bytes_per_launch", "{:.4e}"),
                             ("executed_TFLOPs_f32_in_that_pass", "{:.2f}"), ("executed_TFLOPs_f64_in_that_pass", "{:.2f}")):
                if key in every or key in quoted:
                    f = lambda r: fmt.format(r[key]) if r.get(key) is not None else "-"
                    rows.append((key, f(every), f(quoted)))
            md += [f"| {a} | {b} | {c} |" for a, b, c in rows]
            md.append("")
        open(os.path.join(out, f"{tag}_{leg}_pmc.md"), "w").write("\n".join(md))
        merged[leg] = leg_json
        src = os.path.join(ROOT, "gpurun_out", "prof", leg, "stats.json")
        if os.path.exists(src) and os.path.getsize(src):
            open(os.path.join(out, f"{tag}_{leg}_bench_under_rocprof.json"), "w").write(open(src).read())
        print(f"{leg}: {len(leg_json) - 1} kernels")
    json.dump(merged, open(merged_path, "w"), indent=1)


if __name__ == "__main__":
    main()
