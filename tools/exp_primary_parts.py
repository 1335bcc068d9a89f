# experiment: k_primary and the tail of the bench frame, one frame at a time, through the library's own event timing
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ROOT = os.path.join(os.path.dirname(__file__), "..")
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
ctx = capi.Context(0)
out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1)
sc = ctx.upload(flat)
for _ in range(3):
    st = sc.trace_tile(p, out_rgba8=out, want=(), stats=True)["stats"]
ctx.profile_enable(True); ctx.profile_select("primary,tail"); ctx.profile_reset()
for _ in range(20):
    sc.trace_tile(p, out_rgba8=out, want=(), stats=True)
pm, pn = ctx.profile_get("primary"); tm, tn = ctx.profile_get("tail")
print(os.environ.get("RT355_LIB", "default"), f"primary {pm / pn * 1e3:.1f} us, tail {tm / max(tn, 1) * 1e3:.1f} us ({tn} launches), segments {st['rays']}", flush=True)
