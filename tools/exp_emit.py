# experiment: emission time of the bench's 2M + 2M photon lists
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ROOT = os.path.join(os.path.dirname(__file__), "..")
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(1920, 1080)
ctx = capi.Context(0)
sc = ctx.upload(flat)
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1)
lists, _ = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
del lists
ctx.profile_enable(True); ctx.profile_select("photon_emit_bounce"); ctx.profile_reset()
ctx.synchronize(); t = time.perf_counter()
lists, em = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
ctx.synchronize(); dt = time.perf_counter() - t
ms, n = ctx.profile_get("photon_emit_bounce")
print(os.environ.get("RT355_LIB", "default"), f"emission {dt * 1e3:.3f} ms, k_emit {ms:.3f} ms in {n} launches, segments {em['segments']}", flush=True)
