#!/usr/bin/env python3
"""One full mode-B frame (direct light + both radius gathers, 1080p, 2 M + 2 M photons) with the library's probe switch
on: a probe build (tools/build_variant.sh gprobe -DRT_GATHER_PROBE) prints the gather walks' statistics to stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = capi.Context(0)
W, H = 1920, 1080
sc = ctx.upload(FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H))
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=12345)
lists, em = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
md, mc = lists.photon_map(0), lists.photon_map(1)
pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=12345, shadow_bias=1e-4)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
for _ in range(2):
    st = sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())["stats"]
ctx.set_option("probe", 1)
t = time.perf_counter()
st = sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())["stats"]
print("frame ms", (time.perf_counter() - t) * 1e3, st)
ctx.set_option("probe", 0)
