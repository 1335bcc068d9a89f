"""Config-5 sizes on one GPU: emission into device lists, both kd-trees, a k-nearest gather, for N photons per list.
usage: python tools/scale_check.py [photons_per_list]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, synth
from raytracer_amd.scene import FlatScene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = capi.Context(0)
flat = FlatScene.from_json("tests/golden/scene_preset10_flat.json").with_size(1920, 1080)
sc = ctx.upload(flat)
p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=n, num_photons_caustic=n, seed=1)
t = time.perf_counter()
lists, st = sc.emit_many_lists(p, capi.SHARD_PACKED, strict=False)
ctx.synchronize()
print(f"emission: {time.perf_counter() - t:.3f} s", {k: st[k] for k in ("n_diffuse", "n_caustic", "attempts_phase1", "attempts_phase2", "segments", "capped", "rc")}, flush=True)
t = time.perf_counter()
md = lists.photon_map(0); ctx.synchronize()
print(f"diffuse map of {n}: {time.perf_counter() - t:.3f} s", flush=True)
t = time.perf_counter()
mc = lists.photon_map(1); ctx.synchronize()
print(f"caustic map of {n}: {time.perf_counter() - t:.3f} s", flush=True)
hp = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1)
hits = torch.from_numpy(synth.first_hit_queries(sc, hp, 1920, 1080)).cuda()
out = torch.empty((hits.shape[0], 3), dtype=torch.float32, device="cuda")
for name, m in (("diffuse", md), ("caustic", mc)):
    m.gather(hits, k=64, out=out); ctx.synchronize()
    t = time.perf_counter()
    m.gather(hits, k=64, out=out); ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"k=64 gather on the {name} map: {hits.shape[0] / dt / 1e6:.1f} M gathers/s, finite {bool(torch.isfinite(out).all())}, mean {out.mean().item():.4g}", flush=True)
free, total = torch.cuda.mem_get_info()
print(f"device memory in use: {(total - free) / 2**30:.1f} GiB")
