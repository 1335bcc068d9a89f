"""4K frames (preset 10 and the 10 000-primitive scene) with the tail threshold at its default and above the frame size (experiment)"""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi, synth
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
W, H = 3840, 2160
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
scenes = {"preset10": FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H),
          "config4": synth.many_primitive_scene(10000, seed=42, width=W, height=H)}
for name, flat in scenes.items():
    sc = ctx.upload(flat)
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1, max_bounces=8 if name == "config4" else 0)
    for thr in ("2097152", "16777216"):
        os.environ["RT355_TAIL_THRESHOLD"] = thr
        for _ in range(2):
            st = sc.trace_tile(p, out_rgba8=frame, want=())["stats"]
        ctx.synchronize()
        N = 5
        t = time.perf_counter()
        for _ in range(N):
            sc.trace_tile(p, out_rgba8=frame, want=(), stats=False)
        ctx.synchronize()
        dt = (time.perf_counter() - t) / N
        print(name, "threshold", thr, f"{dt*1e3:.3f} ms/frame {st['rays']/dt/1e6:.0f} Mrays/s rays {st['rays']}", flush=True)
