"""the closest-hit scan alone: rt_intersect on 2 M random rays against preset 10 (experiment; kernel time from the library's events)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json"))
sc = ctx.upload(flat)
rng = np.random.default_rng(1)
n = 2_000_000
o = rng.uniform(-5, 5, (n, 3)); o[:, 2] = rng.uniform(0.1, 8, n)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
p = capi.default_params(precision=capi.PRECISION_F32)
ctx.profile_enable(True); ctx.profile_select(None)
for _ in range(2):
    sc.intersect(p, capi.SCAN_CLOSEST, o, d)
ctx.profile_reset()
for _ in range(3):
    sc.intersect(p, capi.SCAN_CLOSEST, o, d)
ms, k = ctx.profile_get("scan_closest")
print(f"scan_closest: {ms / k * 1e3:.1f} us per launch of {n} rays x {flat.n_shapes} primitives = {n * flat.n_shapes / (ms / k * 1e-3) / 1e9:.1f} G tests/s, {n / (ms / k * 1e-3) / 1e9:.2f} Grays/s")
