# experiment: the bench's headline frames (preset 10, 1920x1080, no direct light, no maps): exclusive kernel times and
# ms per frame with four frames in flight.  Options come from RT355_* environment variables / RT355_LIB.
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ROOT = os.path.join(os.path.dirname(__file__), "..")
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
ctx = capi.Context(0)
sc = ctx.upload(flat)
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1)
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
st = sc.trace_tile(p, out_rgba8=outs[0], want=())["stats"]
ctx.profile_enable(True); ctx.profile_select("primary,tail"); ctx.profile_reset()
for _ in range(64):
    sc.trace_tile(p, out_rgba8=outs[0], want=())
pm, pn = ctx.profile_get("primary"); tm, tn = ctx.profile_get("tail")
ctx.profile_enable(False)
res = []
for nf in (1, 4):
    ctx.set_frames_in_flight(nf)
    for i in range(8):
        sc.trace_tile(p, out_rgba8=outs[i % 4], want=(), stats=False)
    ctx.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for i in range(400):
            sc.trace_tile(p, out_rgba8=outs[i % 4], want=(), stats=False)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t) / 400)
    res.append(best)
    ctx.set_frames_in_flight(1)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT355_"))
print(f"[{tag}] rays {st['rays']} tests/seg {st['prim_tests'] / st['rays']:.2f} | exclusive primary {pm / pn * 1e3:.1f} us, tail {tm / tn * 1e3:.1f} us | "
      f"frame: 1 in flight {res[0] * 1e3:.4f} ms, 4 in flight {res[1] * 1e3:.4f} ms = {st['rays'] / res[1] / 1e9:.2f} Grays/s", flush=True)
