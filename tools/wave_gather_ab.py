"""The reference's full render (direct light + both radius gathers over 2 M + 2 M emitted photons, 1920x1080, fp32) with
the first hits' photon sums gathered lane by lane (wave_gather 0) and wave by wave (box limits 4 / 8 / 16 / 64 ball
areas): ms per frame, and the frames against the lane-by-lane one."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.scene import FlatScene

W, H = 1920, 1080
ctx = capi.Context(0)
flat = FlatScene.from_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
sc = ctx.upload(flat)
pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1)
lists, _ = sc.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
md, mc = lists.photon_map(0), lists.photon_map(1)
pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1, shadow_bias=1e-4)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
ref = None
combos = [(0, 0.0, 0, 0), (1, 8.0, 1, 4096), (1, 8.0, 1, 4096), (1, 8.0, 1, 2048), (1, 8.0, 1, 8192), (1, 8.0, 0, 0)]
for defer, ratio, srt, split in combos:
    with ctx.option("wave_gather", ratio, 8), ctx.option("defer_gathers", defer, 1), ctx.option("gather_sort", srt, 1), ctx.option("gather_split", split, 4096):
        out = sc.trace_tile(pf, md, mc, want=("pre",))
        ctx.synchronize()
        ctx.profile_enable(True); ctx.profile_reset()
        t = time.perf_counter()
        for _ in range(3):
            st = sc.trace_tile(pf, md, mc, out_rgba8=frame, want=())["stats"]
        ctx.synchronize()
        dt = (time.perf_counter() - t) / 3
        kern = {k: round(ctx.profile_get(k)[0] / 3, 3) for k in ("primary", "tail", "gather_keys", "gather_sort", "gather_deferred", "gather_items", "resolve")}
        ctx.profile_enable(False)
    pre = out["pre"]
    line = {"defer": defer, "wave_gather": ratio, "sort": srt, "split": split, "ms_per_frame": dt * 1e3, "gathers": st["gathers"], "rays": st["rays"], "kernel_ms": kern}
    if ref is None:
        ref = pre
    else:
        d = np.abs(pre[..., :3] - ref[..., :3])
        rel = d / np.maximum(1.0, np.abs(ref[..., :3]))
        line.update(max_abs=float(d.max()), max_rel=float(rel.max()), px_off_1e_3=float((rel.max(axis=-1) > 1e-3).mean()))
    print(json.dumps(line), flush=True)
