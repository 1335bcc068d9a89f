"""mode A at 1080p and photon emission throughput (experiment)"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
out = {}
W, H = 1920, 1080
flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_rayjs3_flat.json")).with_size(W, H)
sc = ctx.upload(flat)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
p = capi.default_params(precision=capi.PRECISION_F32, mode=capi.MODE_WHITTED, seed=1, max_trace_dist=100.0, max_depth=20, sub_sample=1)
for _ in range(3):
    st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
ctx.synchronize(); N = 10; t = time.perf_counter()
for _ in range(N):
    st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
ctx.synchronize(); dt = (time.perf_counter() - t) / N
out["modeA_rayjs3_1080p"] = {"ms": dt * 1e3, "Mrays_per_s": st["rays"] / dt / 1e6, "rays": st["rays"], "longest": st["bounces"], "n_shapes": int(flat.n_shapes)}
# emission
flat10 = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json"))
sc10 = ctx.upload(flat10)
for nd, nc in ((100_000, 100_000), (2_000_000, 2_000_000)):
    pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=nd, num_photons_caustic=nc, seed=1)
    sc10.emit_many(pe, strict=False)
    t = time.perf_counter(); e = sc10.emit_many(pe, strict=False); dt = time.perf_counter() - t
    s = e["stats"]
    out[f"emit_{nd}"] = {"s": dt, "stored_per_s": (s["n_diffuse"] + s["n_caustic"]) / dt, "attempts": s["attempts_phase1"] + s["attempts_phase2"], "attempts_per_s": (s["attempts_phase1"] + s["attempts_phase2"]) / dt, "segments_per_s": s["segments"] / dt, "capped": s["capped"], "errors": s["errors"]}
print(json.dumps(out))
