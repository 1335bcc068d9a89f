"""wavefront schedule of the bench frame: time of the closest-hit kernel and of the shade kernel per frame (experiment)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RT355_TAIL_THRESHOLD"] = "0"
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
sc = ctx.upload(flat)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
for mb in (2, 3, 0):
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1, max_bounces=mb)
    ctx.profile_enable(True); ctx.profile_select(None)
    for _ in range(2):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    ctx.profile_reset()
    N = 3
    for _ in range(N):
        st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
    out = {k: ctx.profile_get(k) for k in ("primary", "intersect_closest", "shade", "queue_scan", "queue_scatter", "tail")}
    print("max_bounces", mb, "rays", st["rays"], {k: (round(v[0] / N * 1e3, 1), v[1] // N) for k, v in out.items()}, flush=True)
