"""Config-4 frame (10 000 primitives, 3840x2160, up to 8 segments) with the hierarchy walked near child first per
direction octant against one fixed child order: ms per frame and primitive tests of both."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, synth

ctx = capi.Context(0)
flat = synth.many_primitive_scene(10000, seed=42, width=3840, height=2160)
frame = torch.empty((2160, 3840, 4), dtype=torch.uint8, device="cuda:0")
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1, max_bounces=8)
out = {}
for name, one in (("near_first", 0), ("one_order", 1), ("near_first_again", 0)):
    with ctx.option("bvh_one_order", one, 0):
        sc = ctx.upload(flat)
    for _ in range(2):
        st = sc.trace_tile(p, None, None, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        st = sc.trace_tile(p, None, None, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    out[name] = {"ms_per_frame": (time.perf_counter() - t) / 5 * 1e3, "prim_tests": st["prim_tests"], "rays": st["rays"]}
print(json.dumps(out))
