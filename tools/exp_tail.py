"""tail-kernel time as a function of the bounce cap (experiment; prints one line per cap)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
sc = ctx.upload(flat)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
for mb in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 64):
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1, max_bounces=mb)
    for _ in range(3):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    ctx.profile_enable(True); ctx.profile_select(None); ctx.profile_reset()
    N = 10
    t0 = time.perf_counter()
    for _ in range(N):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / N
    tail, _ = ctx.profile_get("tail"); prim, _ = ctx.profile_get("primary")
    ctx.profile_enable(False)
    print(f"max_bounces {mb:3d} rays {st['rays']:8d} tail_rays {st['tail_rays']:8d} wall {dt*1e3:.3f} ms primary {prim/N*1e3:.1f} us tail {tail/N*1e3:.1f} us", flush=True)
