#!/usr/bin/env python3
"""First pass of the full mode-B frame by parts: kernel time (HIP events) of the first pass with direct light and photon maps
switched on and off (1080p, default scene, 2 M + 2 M photons)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
W, H = 1920, 1080
sc = ctx.upload(FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H))
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=12345)
lists, em = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
md, mc = lists.photon_map(0), lists.photon_map(1)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
for dl in (0, 1):
    for pm in (0, 1):
        pf = capi.default_params(precision=capi.PRECISION_F32, seed=12345, shadow_bias=1e-4, use_direct_light=dl, use_photon_maps=pm)
        for _ in range(2):
            sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(5):
            st = sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())["stats"]
        out = {k: ctx.profile_get(k) for k in ("primary", "tail", "gather_deferred", "gather_items", "gather_sort", "resolve")}
        ctx.profile_enable(False)
        print(f"direct light {dl} photon maps {pm}: " + ", ".join(f"{k} {v[0] / max(v[1], 1) * 1e3:.0f} us" for k, v in out.items()), f"shadow rays {st['shadow_rays']}")
