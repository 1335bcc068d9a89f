"""Kernel times of the full 1080p frame (fp32, 2 M + 2 M emitted photons) with direct light and the photon gathers
switched on one at a time."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
W, H = 1920, 1080
ctx = capi.Context(0)
sc = ctx.upload(FlatScene.from_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scene_preset10_flat.json")).with_size(W, H))
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1)
lists, _ = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
md, mc = lists.photon_map(0), lists.photon_map(1)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
for dl, pm in ((0, 0), (1, 0), (0, 1), (1, 1)):
    pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1, shadow_bias=1e-4,
                             use_direct_light=dl, use_photon_maps=pm)
    sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())
    ctx.synchronize(); ctx.profile_enable(True); ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        st = sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())["stats"]
    ctx.synchronize(); dt = (time.perf_counter() - t) / 3
    kern = {k: round(ctx.profile_get(k)[0] / 3, 3) for k in ("primary", "tail", "gather_sort", "gather_deferred", "gather_items")}
    ctx.profile_enable(False)
    print(json.dumps({"direct_light": dl, "photon_maps": pm, "ms_per_frame": round(dt * 1e3, 3), "kernel_ms": kern,
                      "rays": st["rays"], "shadow_rays": st["shadow_rays"], "gathers": st["gathers"], "bounces": st.get("bounces")}), flush=True)
