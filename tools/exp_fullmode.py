"""the reference's default render at 1080p: direct light and both photon maps on (experiment / report)"""
import os, sys, time, json
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
p = capi.default_params(precision=capi.PRECISION_F32, seed=1, sub_sample=1, num_photons_diffuse=20000, num_photons_caustic=50000)
t = time.perf_counter()
em = sc.emit_many(p)
md, mc = ctx.photon_map(em["diffuse"]), ctx.photon_map(em["caustic"])
print("emission + trees %.1f ms, photons %d + %d" % ((time.perf_counter() - t) * 1e3, len(em["diffuse"]), len(em["caustic"])))
for kw in (dict(use_direct_light=1, use_photon_maps=1), dict(use_direct_light=1, use_photon_maps=0), dict(use_direct_light=0, use_photon_maps=1)):
    for k, v in kw.items():
        setattr(p, k, v)
    ctx.profile_enable(True); ctx.profile_select(None)
    for _ in range(2):
        st = sc.trace_tile(p, md, mc, out_rgba8=frame, want=())["stats"]
    ctx.profile_reset()
    N = 5
    t = time.perf_counter()
    for _ in range(N):
        st = sc.trace_tile(p, md, mc, out_rgba8=frame, want=())["stats"]
    ctx.synchronize()
    dt = (time.perf_counter() - t) / N
    pr, _ = ctx.profile_get("primary"); tl, _ = ctx.profile_get("tail")
    ctx.profile_enable(False)
    print(kw, f"{dt*1e3:.3f} ms/frame rays {st['rays']} shadow {st['shadow_rays']} gathers {st['gathers']} primary {pr/N*1e3:.0f} us tail {tl/N*1e3:.0f} us", flush=True)
