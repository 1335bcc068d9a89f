"""config 4: 10 000-primitive scene at 3840x2160, up to 8 segments per path (experiment / report)"""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi, synth
ctx = capi.Context(0)
W, H = 3840, 2160
t0 = time.perf_counter()
flat = synth.many_primitive_scene(10000, seed=42, width=W, height=H)
t1 = time.perf_counter()
sc = ctx.upload(flat)
t2 = time.perf_counter()
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
out = {"n_shapes": int(flat.n_shapes), "scene_build_s": t1 - t0, "upload_and_hierarchy_s": t2 - t1}
for direct in (0, 1):
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=direct, use_photon_maps=0, seed=1, sub_sample=1, max_bounces=8, shadow_bias=1e-4)
    for _ in range(2):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    N = 5
    t = time.perf_counter()
    for _ in range(N):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    dt = (time.perf_counter() - t) / N
    segs = st["rays"] + st["shadow_rays"]
    out["direct_light_%d" % direct] = {"ms_per_frame": dt * 1e3, "Mrays_per_s": st["rays"] / dt / 1e6, "Msegments_incl_shadow_per_s": segs / dt / 1e6,
        "rays": st["rays"], "shadow_rays": st["shadow_rays"], "prim_tests": st["prim_tests"],
        "brute_force_tests": segs * st["prims_per_ray_brute_force"], "tests_avoided_frac": 1 - st["prim_tests"] / (segs * st["prims_per_ray_brute_force"]),
        "longest_path": st["bounces"]}
print(json.dumps(out))
