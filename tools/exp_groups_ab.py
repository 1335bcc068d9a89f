# scan groups on / off: same frames bit for bit?  and the bench frame's time
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ROOT = os.path.join(os.path.dirname(__file__), "..")
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", f"scene_preset{sys.argv[1] if len(sys.argv) > 1 else 10}_flat.json")).with_size(W, H)
ctx = capi.Context(0)
ctx.set_option("scan_groups", 0); sc0 = ctx.upload(flat)
ctx.set_option("scan_groups", 1); sc1 = ctx.upload(flat)
for kw in (dict(use_direct_light=0, use_photon_maps=0), dict(use_photon_maps=0)):
    p = capi.default_params(precision=capi.PRECISION_F32, seed=1, **kw)
    a = sc0.trace_tile(p, want=("pre",)); b = sc1.trace_tile(p, want=("pre",))
    same = np.array_equal(a["pre"].view(np.uint64), b["pre"].view(np.uint64))
    print(kw, "identical:", same, "rays", a["stats"]["rays"], b["stats"]["rays"], flush=True)
    if not same:
        diff = (a["pre"].view(np.uint64) != b["pre"].view(np.uint64)).any(-1)
        print("  pixels differing:", int(diff.sum()), np.argwhere(diff.reshape(H, W))[:5])
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1)
for name, sc in (("groups off", sc0), ("groups on", sc1)):
    for _ in range(3):
        st = sc.trace_tile(p, out_rgba8=outs[0], want=(), stats=True)["stats"]
    ctx.set_frames_in_flight(3)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
    for i in range(n):
        sc.trace_tile(p, out_rgba8=outs[i % 3], want=(), stats=False)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / n
    ctx.set_frames_in_flight(1)
    print(f"{name}: {dt * 1e3:.4f} ms/frame, {st['rays'] / dt / 1e9:.2f} Grays/s", flush=True)
pf = capi.default_params(precision=capi.PRECISION_F32, use_photon_maps=0, seed=1, sub_sample=1)
for name, sc in (("direct light, groups off", sc0), ("direct light, groups on", sc1)):
    for _ in range(2):
        sc.trace_tile(pf, out_rgba8=outs[0], want=(), stats=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for i in range(n):
        sc.trace_tile(pf, out_rgba8=outs[0], want=(), stats=False)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.4f} ms/frame", flush=True)
