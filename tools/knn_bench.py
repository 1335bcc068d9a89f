"""k-nearest gather legs: synthetic map with tile-ordered and raster-ordered queries, a real emitted map at the frame's
first-hit points.  usage: python tools/knn_bench.py [heap]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, synth
from raytracer_amd.scene import FlatScene
ctx = capi.Context(0)
W, H, K = 1920, 1080, 64
if os.environ.get("KNN_VERBOSE") or os.environ.get("RT355_KNN_VERBOSE"):
    ctx.set_option("knn_verbose", 1)
def rate(pm, hits, name, reps=3):
    d = torch.from_numpy(hits).cuda(); out = torch.empty((d.shape[0], 3), dtype=torch.float32, device="cuda")
    pm.gather(d, k=K, out=out); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(reps): pm.gather(d, k=K, out=out)
    ctx.synchronize(); dt = (time.perf_counter() - t) / reps
    ev = {kk: ctx.profile_get(kk) for kk in ("gather_knn", "gather_knn_split", "gather_knn_single", "gather_knn_heap_fallback")}
    ctx.profile_enable(False)
    print(f"{name:34s} {d.shape[0] / dt / 1e6:9.1f} M gathers/s   {dt * 1e3:8.3f} ms   kernels (ms total, launches): " +
          ", ".join(f"{kk.replace('gather_knn', 'g')} {v[0] / reps:.3f}/{v[1] // reps}" for kk, v in ev.items()), flush=True)
    return out.cpu().numpy()
rec = synth.box_photons(1_000_000, seed=7); hits = synth.box_queries(W, H)
pm = ctx.photon_map(torch.from_numpy(rec).cuda())
a = rate(pm, hits, "synthetic 1M, 8x8-tile order")
order = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
rate(pm, np.ascontiguousarray(hits[np.argsort(order)]), "synthetic 1M, raster order")
rng = np.random.default_rng(1)
rate(pm, np.ascontiguousarray(hits[rng.permutation(hits.shape[0])[:200000]]), "synthetic 1M, random order (200k)")
flat = FlatScene.from_json("tests/golden/scene_preset10_flat.json").with_size(W, H)
sc = ctx.upload(flat)
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1)
lists, st = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
md, mc = lists.photon_map(0), lists.photon_map(1)
hp = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1)
rh = synth.first_hit_queries(sc, hp, W, H)
b = rate(mc, rh, "emitted caustic map 2M, first hits")
c = rate(md, rh, "emitted diffuse map 2M, first hits")
if len(sys.argv) > 1 and sys.argv[1] == "heap":
    ctx.set_option("knn_heap", 1)
    sub = np.ascontiguousarray(rh[:: 16])
    rate(mc, sub, "HEAP kernel: caustic map (1/16 of the queries)", reps=1)
    rate(md, sub, "HEAP kernel: diffuse map (1/16)", reps=1)
    rate(pm, np.ascontiguousarray(hits[::16]), "HEAP kernel: synthetic (1/16)", reps=1)
# the same query sets sorted by the Morton code of their positions (what a caller, or the library, could do first)
def morton_order(h, cells=1024):
    p = h[:, :3].astype(np.float64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    c = np.clip(((p - lo) / (hi - lo).max() * cells).astype(np.int64), 0, cells - 1)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return np.argsort(key, kind="stable")
if len(sys.argv) > 1 and sys.argv[1] == "sorted":
    for cells in (256, 1024):
        rate(pm, np.ascontiguousarray(hits[morton_order(hits, cells)]), f"synthetic, Morton order ({cells} cells)")
        rate(mc, np.ascontiguousarray(rh[morton_order(rh, cells)]), f"caustic map, Morton order ({cells} cells)")
        rate(md, np.ascontiguousarray(rh[morton_order(rh, cells)]), f"diffuse map, Morton order ({cells} cells)")
