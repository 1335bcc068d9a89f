# experiment: how much of a frame's time do the primitives of the default scene cost?  Times the bench frame with
# subsets of the primitives removed (ns per segment is the figure to compare; images differ, of course).
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene

ROOT = os.path.join(os.path.dirname(__file__), "..")
W, H = 1920, 1080
flat0 = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)


def without(flat, drop):
    keep = [i for i in range(flat.n_shapes) if i not in drop]
    remap = {o: k for k, o in enumerate(keep)}
    ls = np.array([remap[int(l)] for l in flat.light_shape], np.uint32)
    return FlatScene(flat.shape_meta[keep].copy(), flat.shape_geom[keep].copy(), flat.shape_colour[keep].copy(), ls, flat.light_data,
                     flat.camera, flat.width, flat.height, flat.materials, None)


ctx = capi.Context(0)
dev = torch.device("cuda", 0)
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(3)]
p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, sub_sample=1)
cases = {"all 33": set(), "box faces 4-8 dropped (28)": {4, 5, 6, 7, 8}, "whole box dropped (27)": {3, 4, 5, 6, 7, 8}}
for name, drop in cases.items():
    sc = ctx.upload(without(flat0, drop))
    for _ in range(3):
        st = sc.trace_tile(p, out_rgba8=outs[0], want=(), stats=True)["stats"]
    ctx.set_frames_in_flight(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for i in range(n):
        sc.trace_tile(p, out_rgba8=outs[i % 3], want=(), stats=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    ctx.set_frames_in_flight(1)
    print(f"{name}: {dt * 1e3:.4f} ms/frame, {st['rays']} segments, {dt / st['rays'] * 1e12:.2f} ps/segment, {st['rays'] / dt / 1e9:.2f} Grays/s", flush=True)
    sc.close()
