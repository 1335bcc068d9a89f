"""prints the headline numbers of bench result files: python tools/show.py name ... (gpurun_out/exp_<name>.json)"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f"gpurun_out/exp_{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    prim = r["kernel_ms_per_step"] * 1000
    tail = (r.get("tail_kernel_ms_per_step") or 0) / d["steps"] * 1000 * (d["steps"] if (r.get("tail_kernel_ms_per_step") or 0) < 1 else 1)
    print(f"{f:12s} {d['value']:9.0f} Mrays/s  {d['ms_per_step']*1000:7.1f} us/step  primary {prim:6.1f} us  tail {tail:6.1f} us  frac {r['frac']:.3f}")
