"""Which map costs what in the full frame: kernel times with only the diffuse / only the caustic map, by gather mode."""
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
pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1, shadow_bias=1e-4, use_direct_light=0)
names = ("primary", "tail", "gather_sort", "gather_deferred")
for label, maps in (("diffuse only", (md, None)), ("caustic only", (None, mc))):
    for defer, ratio in ((0, 0.0), (0, 8.0), (2, 0.0), (2, 8.0)):
        with ctx.option("wave_gather", ratio, 8), ctx.option("defer_gathers", defer, 1):
            sc.trace_tile(pf, maps[0], maps[1], out_rgba8=frame, want=())
            ctx.synchronize(); ctx.profile_enable(True); ctx.profile_reset()
            t = time.perf_counter()
            sc.trace_tile(pf, maps[0], maps[1], out_rgba8=frame, want=())
            ctx.synchronize(); dt = time.perf_counter() - t
            kern = {k: round(ctx.profile_get(k)[0], 3) for k in names}
            ctx.profile_enable(False)
        print(json.dumps({"maps": label, "defer": defer, "wave_gather": ratio, "ms": round(dt * 1e3, 2), "kernel_ms": kern}), flush=True)
