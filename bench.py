#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary segments) of the wavefront tracer on the reference's default
scene at 1920x1080, 1 spp, reflection/refraction only -- plus, as extra fields of the same JSON line, photon k-NN
gathers/s on a 1M-photon map, photon emission, the kd-tree build and the other BASELINE configurations.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one full-frame pass of the hot path.
  N = 1   every step renders the frame on the one GPU (frames queued back to back, four in flight).
  N > 1   ONE job over N ranks, the split of the north star: photon emission sharded by attempt range, the photon
          shards all-gathered from and into device memory (RCCL), the list caps applied and the kd-trees built
          identically on every rank, then every step's frame traced as interleaved row bands, one share per rank,
          and gathered to rank 0.  `value` is the frames' segments per second (strong scaling); the emission,
          all-gather and build phases are reported under "phases".
          --independent-frames gives the old mode instead: every rank renders whole frames of its own (weak scaling,
          no collective in the timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic flop per ray/primitive test, by primitive type (add/sub/mul/div/sqrt = 1, FMA = 2,
# compare/abs/neg/select = 0; the path that reaches the first candidate root's validity test)
FLOP_PER_TEST = {0: 14, 1: 20, 2: 34, 3: 54, 4: 46, 5: 45, 6: 30, 7: 32}
PEAK_FP32_VALU_TFLOPS = 157.3      # 256 CU x 4 SIMD x 64 flop/clk (32 lanes x FMA) x 2.4 GHz; packed fp32 issues at half rate, so it is the same peak
PEAK_HBM_GBS = 8000.0
BYTES_PER_GATHER = 64 * 36 + 36 + 12
PROFILE_TAG = "r03"
_COUNTERS = None


def leg_counters(leg, pattern, which="largest_grid_dispatches"):
    """Counter record of the kernel whose name contains `pattern`, from the committed per-leg counter passes of this same
    command (rocprofv3 cannot run inside the benchmark): profiles/<tag>_counters_by_leg.json, written by
    tools/profile_legs.sh + tools/make_leg_profiles.py from runs with --legs <leg> ('frames' = --legs none).
    which: the dispatches with the kernel's largest grid in that leg (the ones a roofline figure quotes) or all_dispatches."""
    global _COUNTERS
    if _COUNTERS is None:
        try:
            with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_counters_by_leg.json")) as f:
                _COUNTERS = json.load(f)
        except (OSError, ValueError):
            _COUNTERS = {}
    for name, rec in _COUNTERS.get(leg, {}).items():
        if not name.startswith("_") and pattern in name:
            return rec.get(which)
    return None


def measured_traffic(leg, pattern, which="largest_grid_dispatches", launches=1.0):
    """HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes) of `launches` launches of the kernel in that leg"""
    rec = leg_counters(leg, pattern, which)
    return int(rec["traffic_bytes_per_launch"] * launches) if rec and rec.get("traffic_bytes_per_launch") is not None else None


def executed_block(leg, pattern, seconds_per_launch, which="largest_grid_dispatches", launches=1.0):
    """What the SQ counters say the kernel executed, against this run's clock: wave-level fp instructions by class
    (2 FMA + MUL + ADD + TRANS) x the mean active lanes of a VALU instruction = flop per launch (counter passes of this
    command's leg), divided by the launch duration measured in this run.  fp64 flop count double in `frac` (the vector
    unit issues them at half the fp32 rate)."""
    rec = leg_counters(leg, pattern, which)
    if not rec or rec.get("executed_flop_f32_per_launch") is None or seconds_per_launch <= 0:
        return None
    f32, f64 = rec["executed_flop_f32_per_launch"] * launches, rec["executed_flop_f64_per_launch"] * launches
    return {"flop_f32_per_launch": f32, "flop_f64_per_launch": f64, "lanes_per_valu_instruction": rec.get("lanes_per_valu_instruction"),
            "waiting_over_wave_cycles": rec.get("waiting_over_wave_cycles"), "fp_share_of_valu_instructions": rec.get("flop_instruction_share_of_valu"),
            "achieved_f32": f32 / seconds_per_launch / 1e12, "achieved_f64": f64 / seconds_per_launch / 1e12, "unit": "TFLOP/s",
            "frac": (f32 + 2 * f64) / seconds_per_launch / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "source": f"profiles/{PROFILE_TAG}_counters_by_leg.json['{leg}'][kernel containing '{pattern}']['{which}'] (tables: profiles/{PROFILE_TAG}_{leg}_pmc.md); "
                      "64 x wave instructions (2 FMA + MUL + ADD + TRANS) x lanes / 64, over this run's launch duration"}


def frame_executed(in_flight, seconds_per_frame):
    """both kernels of a headline frame (the pair the library runs with that many frames in flight) against the frame time"""
    pair = ("k_primary_tiles<", "k_tail_persistent") if in_flight >= 3 else ("k_primary<float", "k_tail<float")
    parts = [executed_block("frames", k, seconds_per_frame) for k in pair]
    if any(x is None for x in parts):
        return None
    f32 = sum(x["flop_f32_per_launch"] for x in parts)
    return {"flop_f32_per_frame": f32, "achieved_f32": f32 / seconds_per_frame / 1e12, "unit": "TFLOP/s", "frac": f32 / seconds_per_frame / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "kernels": list(pair), "source": parts[0]["source"]}


def flop_per_ray_pass(flat):
    return int(sum(FLOP_PER_TEST[int(t)] for t in flat.shape_meta[:, 0]))


# ------------------------------------------------------------------ CPU baselines (the oracle on the host's cores)
def _cpu_band(job):
    """worker: one row band of one frame with the C oracle"""
    from oracle import oracle as ora
    flat, seed, y0, h, full = job
    osc = _cpu_band.cache.get("scene")
    if osc is None:
        osc = _cpu_band.cache["scene"] = ora.Scene(flat)
    op = ora.params(seed=seed, use_direct_light=int(full), use_photon_maps=0)
    _, _, rays, _ = ora.render(osc, op, None, 0, y0, flat.width, h)
    return rays


_cpu_band.cache = {}


def cpu_workers():
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    n = os.cpu_count() or 1
    return max(1, min(n, share if share > 0 else 16))


def cpu_baseline(flat, seed, budget_s=10.0):
    """The CPU oracle (fp64 restatement of the reference; the reference's JavaScript cannot travel to the GPU box) on
    the same workload: whole frames cut into row bands over a pool of worker processes, one per host core of this
    job's share, for about budget_s seconds; then one frame on ONE core for the per-core figure.  Must run before the
    process touches the GPU (the pool forks)."""
    import multiprocessing as mp
    workers = cpu_workers()
    band = 8
    bands = [(y, min(band, flat.height - y)) for y in range(0, flat.height, band)]
    rays_total, frames = 0, 0
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_band, [(flat, seed, y, h, False) for (y, h) in bands[:workers]])     # warm the workers (library load, scene)
        t0 = time.perf_counter()
        while True:
            rays_total += sum(pool.map(_cpu_band, [(flat, seed + frames, y, h, False) for (y, h) in bands], chunksize=2))
            frames += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or frames >= 64:
                break
    t1 = time.perf_counter()
    one = sum(_cpu_band((flat, seed, y, h, False)) for (y, h) in bands[::8])                # every 8th band on one core
    dt1 = time.perf_counter() - t1
    return {"value": rays_total / dt / 1e6, "unit": "Mrays/s", "cores": workers, "kind": "port", "workers": workers,
            "host_cpus": os.cpu_count(), "one_core_value": one / dt1 / 1e6,
            "sample": f"{frames} full {flat.width}x{flat.height} frames of the same workload (seeds {seed}..{seed + frames - 1}) in {band}-row bands over "
                      f"{workers} worker processes, {rays_total} rays in {dt:.2f} s; one core: every 8th band of one frame, {one} rays in {dt1:.2f} s; "
                      "oracle/rt_oracle.c (fp64, bit-exact with the reference JS on its golden frames)"}


_FULL = {}     # the oracle's scene, parameters and photon maps of the full-render leg, set before the pool forks


def _cpu_full_band(job):
    from oracle import oracle as ora
    y0, h = job
    osc, op, maps, w = _FULL["job"]
    return ora.render(osc, op, maps, 0, y0, w, h)[2]


def cpu_full_render(seed, w=320, h=240, photons=10000):
    """The reference's render in full (direct light, both radius gathers) on the CPU oracle: BASELINE config 0's frame
    -- the default scene at 320x240 with 10k + 10k photons -- emission and tree builds on one core (as the reference
    does them), the frame over the worker pool and again on one core.  bench.py times the GPU on the same frame."""
    import multiprocessing as mp
    from oracle import oracle as ora
    from raytracer_amd.scene import FlatScene
    flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(w, h)
    osc = ora.Scene(flat)
    op = ora.params(seed=seed, num_photons_diffuse=photons, num_photons_caustic=photons, sub_sample=1)
    t0 = time.perf_counter()
    em = ora.emit_many(osc, op)
    t_emit = time.perf_counter() - t0
    t0 = time.perf_counter()
    maps = ora.Maps(em["diffuse"], em["caustic"])
    t_build = time.perf_counter() - t0
    _FULL["job"] = (osc, op, maps, w)
    workers = cpu_workers()
    bands = [(y, min(4, h - y)) for y in range(0, h, 4)]
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_full_band, bands[:workers])
        t0 = time.perf_counter()
        rays = sum(pool.map(_cpu_full_band, bands))
        t_pool = time.perf_counter() - t0
    t0 = time.perf_counter()
    rays1 = sum(_cpu_full_band(b) for b in bands)
    t_one = time.perf_counter() - t0
    return {"frame": f"{w}x{h}, {photons} + {photons} photons, 1 sub-sample", "rays": int(rays), "emission_s_one_core": t_emit, "kd_builds_s_one_core": t_build,
            "frame_ms_all_cores": t_pool * 1e3, "frame_ms_one_core": t_one * 1e3, "cores": workers, "kind": "port",
            "Mrays_per_s_all_cores": rays / t_pool / 1e6, "Mrays_per_s_one_core": rays1 / t_one / 1e6}


_KD = {}       # the oracle's tree, set before the pool forks (the workers inherit it instead of unpickling 30 MB per chunk)


def _cpu_gather_chunk(q):
    from oracle import oracle as ora
    pos, perm, dim, r = _KD["tree"]
    found = 0
    for i in range(q.shape[0]):
        found += ora.kd_radius(pos, perm, dim, q[i], r, cap=1024).size
    return found


def cpu_baseline_gather(rec, hits, k, n_queries=1_000_000):
    """CPU side of the photon-map leg: the oracle's kd-tree (the reference's makeTree / nearestNeighbours restated in C)
    built on the same photons (one core: the build is sequential), and its radius query at the radius that holds k
    photons on average -- the reference has no k-nearest query -- for a sample of the same hit points, over the worker pool."""
    import multiprocessing as mp
    from oracle import oracle as ora
    pos = np.ascontiguousarray(rec[:, :3])
    t0 = time.perf_counter()
    perm, dim = ora.kd_build(pos)
    build_s = time.perf_counter() - t0
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    d = hi - lo
    area = 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])
    r = float(np.sqrt(k * area / (np.pi * pos.shape[0])))
    rows = np.random.default_rng(0).choice(hits.shape[0], min(n_queries, hits.shape[0]), replace=False)
    q = hits[rows, :3].astype(np.float64)
    workers = cpu_workers()
    chunks = np.array_split(q, workers * 4)
    _KD["tree"] = (pos, perm, dim, r)
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_gather_chunk, [c[:8] for c in chunks[:workers]])          # warm the workers
        t0 = time.perf_counter()
        found = sum(pool.map(_cpu_gather_chunk, chunks))
        dt = time.perf_counter() - t0
    return {"value": q.shape[0] / dt, "unit": "gathers/s", "cores": workers, "kind": "port", "kd_build_s": build_s, "kd_build_cores": 1,
            "sample": f"{q.shape[0]} of the {hits.shape[0]} hit points, radius query r={r:.4f} ({found / q.shape[0]:.1f} photons per query) "
                      f"on the oracle's kd-tree of the same {pos.shape[0]} photons, {dt:.2f} s over {workers} worker processes"}


# ------------------------------------------------------------------ GPU legs
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--photons", type=int, default=1_000_000)
    ap.add_argument("--knn", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the photon-map (k-NN gather) leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs for the older tracer, the 10k-primitive scene, emission and the full render")
    ap.add_argument("--legs", default="all", help="report-only legs to run, comma-separated: gather (1M-photon build + k-NN gather, tile order), raster, mode_a, config4, "
                                                 "emit, full_b, config0, knn_real, radius_real; 'all' (default) or 'none'.  One leg per run is how the per-leg profiles are made")
    ap.add_argument("--independent-frames", action="store_true", help="N > 1: every rank renders whole frames of its own (weak scaling) instead of the one-job split")
    ap.add_argument("--photons-per-gpu", type=int, default=12_500_000, help="N > 1: stored photons per GPU of the sharded emission (half diffuse, half caustic)")
    ap.add_argument("--band", type=int, default=16, help="N > 1: rows per interleaved band")
    ap.add_argument("--frames-per-gather", type=int, default=8, help="N > 1: frames whose bands travel to rank 0 in one collective")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    from raytracer_amd.scene import FlatScene
    W, H = args.width, args.height
    flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
    seed = 1 + (rank if args.independent_frames else 0)

    # the CPU baselines fork worker processes: they run first, before this process initialises the GPU
    cpu_line, cpu_gather_line, cpu_full_line = None, None, None
    if not args.no_cpu_baseline and world == 1:
        from raytracer_amd import synth as _synth
        cpu_line = cpu_baseline(flat, seed)
        if not args.no_other_configs:
            cpu_full_line = cpu_full_render(seed)
        if not args.no_gather:
            cpu_gather_line = cpu_baseline_gather(_synth.box_photons(args.photons, seed=7), _synth.box_queries(W, H), args.knn)

    import torch
    import torch.distributed as dist
    # RT355_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo -- a dry run of the N > 1 control flow on a
    # one-GPU box (the numbers mean nothing); the real thing is one GPU per rank over RCCL
    rehearsal = os.environ.get("RT355_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearsal else dev          # where the tensors of the collectives live

    from raytracer_amd import capi, parallel, synth

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = capi.Context(local)
    sc = ctx.upload(flat)
    fl = flop_per_ray_pass(flat)
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=seed, sub_sample=1)
    split = world > 1 and not args.independent_frames
    phases = None

    ALL_LEGS = ("gather", "raster", "mode_a", "config4", "emit", "full_b", "config0", "knn_real", "radius_real")
    legs = set(ALL_LEGS) if args.legs == "all" else {x for x in args.legs.split(",") if x and x != "none"}
    if legs - set(ALL_LEGS):
        ap.error(f"unknown leg(s) {sorted(legs - set(ALL_LEGS))}; known: {', '.join(ALL_LEGS)}")
    if args.no_gather:
        legs -= {"gather", "raster", "knn_real", "radius_real"}
    if args.no_other_configs:
        legs -= {"mode_a", "config4", "emit", "full_b", "config0", "knn_real", "radius_real"}
    want = lambda name: name in legs

    def other_legs():
        """the report-only legs (photon maps, the older tracer, the large scene, emission, full renders); each times itself"""
        # ---- photon-map leg: 1M-photon kd-tree build + k-NN radiance gather per pixel ----
        extra = {}
        roof_gather = roof_build = roof_emit = None
        if want("gather") or want("raster"):          # (every rank runs the report-only legs: they hold barriers)
            rec_host, hits_host = synth.box_photons(args.photons, seed=7), synth.box_queries(W, H)
            rec = torch.from_numpy(rec_host).to(dev)
            hits = torch.from_numpy(hits_host).to(dev)
            rgb = torch.empty((hits.shape[0], 3), dtype=torch.float32, device=dev)
            pm = ctx.photon_map(rec)                  # warm-up build (first-use costs of the sort library)
            del pm
            torch.cuda.synchronize()
            builds = []
            for _ in range(3 if want("gather") else 1):
                tb = time.perf_counter()
                pm = ctx.photon_map(rec)
                builds.append(time.perf_counter() - tb)
            build_s = min(builds)
            n_ph = args.photons
            levels = int(np.ceil(np.log2(max(n_ph, 2))))
            build_bytes = n_ph * (12 + 12) * levels + n_ph * 5
            roof_build = {"kernel": "photon map build (all launches of rt_photon_map_build: position extract, level-synchronous kd build, permute)", "bound": "hbm",
                          "achieved": build_bytes / build_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": build_bytes / build_s / 1e9 / PEAK_HBM_GBS,
                          "traffic": None, "algorithmic_bytes": build_bytes, "seconds": build_s, "photons": n_ph,
                          "note": "algorithmic bytes = N x (12 B read + 12 B written) x ceil(log2 N) + 5 N; time = host clock around the whole call, best of 3"}
            if want("gather"):
                pm.gather(hits, k=args.knn, out=rgb)                               # warm-up
                ctx.profile_enable(True)
                ctx.profile_select("gather_knn,gather_knn_split,gather_knn_single,gather_knn_heap_fallback")
                ctx.profile_reset()
                gsteps = max(2, min(args.steps, 5))
                barrier()
                tg = time.perf_counter()
                for _ in range(gsteps):
                    pm.gather(hits, k=args.knn, out=rgb)
                ctx.synchronize()
                barrier()
                gdt = time.perf_counter() - tg
                g_ms, g_launches = ctx.profile_get("gather_knn")
                fb_ms, fb_launches = ctx.profile_get("gather_knn_heap_fallback")
                sp_ms, sp_launches = ctx.profile_get("gather_knn_split")
                sg_ms, sg_launches = ctx.profile_get("gather_knn_single")
                ctx.profile_enable(False)
                gathers = hits.shape[0] * gsteps
                gps = torch.tensor([gathers / gdt], dtype=torch.float64, device=cdev)
                if world > 1:
                    dist.all_reduce(gps, op=dist.ReduceOp.SUM)
                g_achieved = hits.shape[0] * BYTES_PER_GATHER / (g_ms / g_launches * 1e-3) / 1e9
                extra = {"knn_gathers_per_s": float(gps.item()), "knn_k": args.knn, "photons": args.photons,
                         "kd_build_s": build_s, "gather_ms_per_step": gdt / gsteps * 1e3,
                         "split_units_ms_per_step": sp_ms / gsteps, "split_unit_launches": sp_launches,
                         "single_query_ms_per_step": sg_ms / gsteps, "single_query_launches": sg_launches,
                         "heap_fallback_ms_per_step": fb_ms / gsteps, "heap_fallback_launches": fb_launches}
                roof_gather = {"kernel": "k_gather_knn_wave", "bound": "hbm", "achieved": g_achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": g_achieved / PEAK_HBM_GBS, "traffic": measured_traffic("gather", "k_gather_knn_wave"),
                               "executed": executed_block("gather", "k_gather_knn_wave", g_ms / g_launches * 1e-3),
                               "algorithmic_bytes_per_gather": BYTES_PER_GATHER, "avg_launch_ms": g_ms / g_launches,
                               "whole_call": {"ms": gdt / gsteps * 1e3, "frac": hits.shape[0] * BYTES_PER_GATHER / (gdt / gsteps) / 1e9 / PEAK_HBM_GBS,
                                              "note": "every launch of rt_gather_radiance (first pass, the units cut into eighths, single queries) and the host's counter read-backs between them"}}
            # the same gather on less friendly inputs: the queries in raster order instead of 8x8 tiles, and a real emitted map
            # (the default scene's caustic + diffuse photons) queried at the frame's own first-hit points
            order = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1) if (H % 8 == 0 and W % 8 == 0) else None
            if order is not None and want("raster"):
                raster = torch.from_numpy(np.ascontiguousarray(hits_host[np.argsort(order)])).to(dev)
                pm.gather(raster, k=args.knn, out=rgb)
                ctx.synchronize()
                tg = time.perf_counter()
                for _ in range(2):
                    pm.gather(raster, k=args.knn, out=rgb)
                ctx.synchronize()
                extra["knn_raster_order"] = {"gathers_per_s": 2 * hits.shape[0] / (time.perf_counter() - tg),
                                             "note": "same queries, row by row (a wave's 64 queries are a 64-pixel strip instead of an 8x8 patch)"}
                del raster
            del pm, rec

        # ---- short legs for the other BASELINE configs (reported under "extra", never part of `value`) ----
        if legs - {"gather", "raster"}:
            frame = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)

            def timed_frames(scene, params, out, n=5, maps=(None, None)):
                for _ in range(2):
                    stt = scene.trace_tile(params, maps[0], maps[1], out_rgba8=out, want=())["stats"]
                ctx.synchronize()
                t_ = time.perf_counter()
                for _ in range(n):
                    stt = scene.trace_tile(params, maps[0], maps[1], out_rgba8=out, want=())["stats"]
                ctx.synchronize()
                return (time.perf_counter() - t_) / n, stt
            if want("mode_a"):
                # the older tracer (ray.js) on its preset 3 at the same resolution
                flat_a = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_rayjs3_flat.json")).with_size(W, H)
                pa = capi.default_params(precision=capi.PRECISION_F32, mode=capi.MODE_WHITTED, seed=seed, max_trace_dist=100.0, max_depth=20, sub_sample=1)
                dt_a, st_a = timed_frames(ctx.upload(flat_a), pa, frame)
                extra["mode_a_rayjs_preset3"] = {"Mrays_per_s": st_a["rays"] / dt_a / 1e6, "ms_per_frame": dt_a * 1e3, "rays_per_frame": st_a["rays"],
                                                 "primitives": int(flat_a.n_shapes), "resolution": [W, H]}
            if want("config4"):
                # configs[3]: 10 000 primitives at 3840x2160, paths of up to 8 segments, through the hierarchy
                flat4 = synth.many_primitive_scene(10000, seed=42, width=3840, height=2160)
                frame4 = torch.empty((2160, 3840, 4), dtype=torch.uint8, device=dev)
                p4 = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=seed, sub_sample=1, max_bounces=8)
                dt4, st4 = timed_frames(ctx.upload(flat4), p4, frame4, n=3)
                brute = st4["rays"] * st4["prims_per_ray_brute_force"]
                extra["config4_10k_primitives_4k"] = {"Mrays_per_s": st4["rays"] / dt4 / 1e6, "ms_per_frame": dt4 * 1e3, "rays_per_frame": st4["rays"],
                                                      "primitives": int(flat4.n_shapes), "prim_tests": st4["prim_tests"], "brute_force_tests": brute,
                                                      "tests_avoided_frac": 1.0 - st4["prim_tests"] / brute}
                del frame4
            need_maps = want("emit") or want("full_b") or want("knn_real") or want("radius_real")
            if need_maps:
                # photon emission on one GPU (the configs[4] kernel): 2 M diffuse + 2 M caustic photons of the default scene, the lists
                # kept in device memory as packed 36-byte photons, then both kd-trees straight from the lists
                pe1 = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=seed)
                lists, _ = sc.emit_many_lists(pe1, capi.SHARD_PACKED, strict=False)
                del lists
                ctx.profile_enable(True)
                ctx.profile_select("photon_emit_bounce")
                ctx.profile_reset()
                ctx.synchronize()
                t_ = time.perf_counter()
                lists, em = sc.emit_many_lists(pe1, capi.SHARD_PACKED, strict=False)
                ctx.synchronize()
                dte = time.perf_counter() - t_
                e_ms, e_n = ctx.profile_get("photon_emit_bounce")
                ctx.profile_enable(False)
                t_ = time.perf_counter()
                md, mc = lists.photon_map(0), lists.photon_map(1)
                ctx.synchronize()
                dtm = time.perf_counter() - t_
                stored = em["n_diffuse"] + em["n_caustic"]
                extra["photon_emission"] = {"stored_photons_per_s": stored / dte, "segments_per_s": em["segments"] / dte,
                                            "attempts": em["attempts_phase1"] + em["attempts_phase2"], "segments": em["segments"], "seconds": dte,
                                            "emit_kernel_ms": e_ms, "emit_kernel_launches": e_n, "maps_from_lists_s": dtm,
                                            "note": "rt_emit_many_lists: photons stay in device memory (packed 36-byte records), list caps applied on the GPU, "
                                                    "8 words read back per batch; segments = emitPhoton calls of the attempts the sequential loop runs"}
                if e_ms > 0:
                    # (the kernel also runs the attempts of a last batch beyond the one that ends a phase: its own segment count is the rate's numerator)
                    ach = em["segments"] * fl / (e_ms * 1e-3) / 1e12
                    roof_emit = {"kernel": "k_emit<float> (createPhoton + the emitPhoton random walk, one attempt per lane; every segment scans all primitives)",
                                 "bound": "valu", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_VALU_TFLOPS,
                                 "traffic": measured_traffic("emit", "k_emit<float", which="all_dispatches", launches=e_n), "algorithmic_flop_per_segment": fl, "segments": em["segments"], "kernel_ms": e_ms, "launches": e_n,
                                 "executed": executed_block("emit", "k_emit<float", e_ms * 1e-3, which="all_dispatches", launches=e_n),
                                 "note": "segments counted are those of the sequential loop; the launches also trace the surplus attempts of each phase's last batch"}
            if want("full_b"):
                # the reference's actual render (mode B): direct light + both radius gathers on those maps, one 1080p frame
                pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=seed, shadow_bias=1e-4)
                dtf, stf = timed_frames(sc, pf, frame, n=3, maps=(md, mc))
                extra["full_mode_b_frame"] = {"ms_per_frame": dtf * 1e3, "Mrays_per_s": stf["rays"] / dtf / 1e6, "rays_per_frame": stf["rays"],
                                              "shadow_rays_per_frame": stf["shadow_rays"], "gathers_per_frame": stf["gathers"], "gathers_per_s": stf["gathers"] / dtf,
                                              "photons": stored, "note": "direct light, diffuse (r = 0.40) and caustic (r = 0.15) radius gathers: the reference's traceRay in full; the gathers are queued, sorted by position and served 64 at a time (k_gather_deferred, k_gather_items)"}
            if want("config0"):
                # ... and BASELINE config 0's frame (320x240, 10k + 10k photons, everything on): the frame the CPU oracle is timed on
                flat0 = flat.with_size(320, 240)
                sc0 = ctx.upload(flat0)
                p0 = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=10000, num_photons_caustic=10000, seed=seed, shadow_bias=1e-4)
                ctx.synchronize()
                t_ = time.perf_counter()
                lists0, _ = sc0.emit_many_lists(p0, capi.SHARD_PACKED, strict=False)
                md0, mc0 = lists0.photon_map(0), lists0.photon_map(1)
                ctx.synchronize()
                dt_prep = time.perf_counter() - t_
                frame0 = torch.empty((240, 320, 4), dtype=torch.uint8, device=dev)
                dt0, st0 = timed_frames(sc0, p0, frame0, n=20, maps=(md0, mc0))
                extra["full_mode_b_config0"] = {"gpu": {"frame_ms": dt0 * 1e3, "emission_and_kd_builds_ms": dt_prep * 1e3, "rays": st0["rays"], "Mrays_per_s": st0["rays"] / dt0 / 1e6},
                                                "cpu": cpu_full_line,
                                                "note": "BASELINE config 0 (default scene at 320x240, 10k + 10k photons, direct light and both radius gathers): the same frame on "
                                                        "the GPU (fp32) and on the CPU oracle (fp64)"}
                del md0, mc0, lists0, sc0
            # the k-nearest gather on that real map at the frame's own first-hit points
            if want("knn_real") or want("radius_real"):
                hp = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=seed, max_bounces=1)
                real_hits = synth.first_hit_queries(sc, hp, W, H)
                if real_hits.shape[0]:
                    rh = torch.from_numpy(real_hits).to(dev)
                    out_r = torch.empty((rh.shape[0], 3), dtype=torch.float32, device=dev)
                    if want("knn_real"):
                        mc.gather(rh, k=args.knn, out=out_r)
                        ctx.synchronize()
                        t_ = time.perf_counter()
                        for _ in range(2):
                            mc.gather(rh, k=args.knn, out=out_r)
                        ctx.synchronize()
                        extra["knn_real_map"] = {"gathers_per_s": 2 * rh.shape[0] / (time.perf_counter() - t_), "photons": em["n_caustic"], "queries": int(rh.shape[0]),
                                                 "note": "the default scene's emitted caustic map, queried at the frame's first-hit points in 8x8-tile order"}
                    if want("radius_real"):
                        # the reference's own gather (rayphoton.js:1765-1815): fixed radius, at the same points, both maps
                        rad = {}
                        for name, pm_, r_ in (("diffuse_r0.40", md, 0.40), ("caustic_r0.15", mc, 0.15)):
                            pm_.gather(rh, r=r_, out=out_r)
                            ctx.synchronize()
                            t_ = time.perf_counter()
                            for _ in range(2):
                                pm_.gather(rh, r=r_, out=out_r)
                            ctx.synchronize()
                            gps = 2 * rh.shape[0] / (time.perf_counter() - t_)
                            # photons inside the balls, from a sample of the queries (rt_kd_radius counts them)
                            samp = real_hits[:: max(1, real_hits.shape[0] // 4096), :3]
                            cnt, _ = pm_.radius(samp, r_, cap=1)
                            mean_in = float(cnt.mean())
                            # per (query, photon inside the ball): 3 sub + 5 (d^2) + 5 (direction . normal) + 6 (three FMAs) = 19 flop
                            tf = gps * mean_in * 19 / 1e12
                            rad[name] = {"gathers_per_s": gps, "photons_in_ball_mean": mean_in, "photon_tests_per_s": gps * mean_in,
                                         "algorithmic_TFLOP_per_s": tf, "frac_of_fp32_peak": tf / PEAK_FP32_VALU_TFLOPS}
                        with ctx.option("defer_gathers", 0, -1):
                            sub = rh[::16].contiguous()
                            out_s = torch.empty((sub.shape[0], 3), dtype=torch.float32, device=dev)
                            md.gather(sub, r=0.40, out=out_s)
                            ctx.synchronize()
                            t_ = time.perf_counter()
                            md.gather(sub, r=0.40, out=out_s)
                            ctx.synchronize()
                            rad["diffuse_r0.40_lane_by_lane"] = {"gathers_per_s": sub.shape[0] / (time.perf_counter() - t_), "note": "every 16th query, one lane walking the tree per query (k_gather_radius)"}
                        rad["note"] = "rt_gather_radiance with k = 0 on the emitted 2 M-photon maps: queries queued, sorted by position, served 64 at a time, subtrees of <= 4096 photons as work items"
                        extra["radius_gather_real_map"] = rad
            if need_maps:
                del md, mc, lists
        return extra, roof_gather, roof_build, roof_emit

    # N = 1: the other legs run first (seconds of GPU work), so that the headline frames are timed on a chip that is already
    # busy and clocked up -- the driver's run times only a few milliseconds of frames
    if not split:
        extra, roof_gather, roof_build, roof_emit = other_legs()

    if not split:
        # ---- frames of config[1] on this rank's GPU, queued back to back, four in flight (one per hardware queue of the process;
        # five share queues: 25 % slower)
        in_flight = 1 if os.environ.get("RT355_BENCH_SERIAL_FRAMES") else int(os.environ.get("RT355_BENCH_IN_FLIGHT", "4"))
        frames_out = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(in_flight)]   # outputs stay in HBM
        frame = frames_out[0]

        def step(stats=True, i=0):
            return sc.trace_tile(p, out_rgba8=frames_out[i % in_flight], want=(), stats=stats)["stats"]

        timed_every = int(os.environ.get("RT355_BENCH_TIME_EVERY", "8"))
        # warm-up: one frame with statistics (every step renders the same frame, same seed: its ray count is this one's),
        # the rest queued like the timed frames, so that every in-flight slot has its workspaces allocated and its lists of
        # reachable primitives made before the clock starts
        # for the report only, frames one at a time before the timed ones: both kernels with the chip to themselves (in the
        # timed region a kernel shares it with the neighbouring frames' kernels)
        ctx.profile_enable(True)
        ctx.profile_select("primary,tail")
        step()
        ctx.profile_reset()
        for _ in range(64):
            step()
        tx_ms, tx_n = ctx.profile_get("tail")
        px_ms, px_n = ctx.profile_get("primary")
        # (every 8th launch of the dominant kernel carries the timing events: a sample over the whole timed region)
        ctx.profile_enable(timed_every if os.environ.get("RT355_BENCH_NO_EVENTS") is None else 0)
        ctx.profile_select(None if os.environ.get("RT355_BENCH_ALL_EVENTS") else "tail")   # time the dominant kernel only: every event pair costs stream time
        st = step()
        ctx.set_frames_in_flight(in_flight)
        for i in range(max(args.warmup, 1) - 1):
            step(stats=False, i=i)
        ctx.synchronize()
        ctx.profile_reset()                     # (the warm-up's timing events go back to the library's pool)
        # the timed frames are only queued (no statistics read back in between) and the clock stops after the library's
        # streams have drained.  Four frames in flight: consecutive frames rotate through four streams, workspaces and output
        # buffers, so the thinly occupied end of one frame's path tracing overlaps the start of the next; from three frames in
        # flight on the library follows the surviving paths with persistent lanes (fewer instructions per frame, a longer
        # serial chain: it needs the neighbours to fill the gaps).  A kernel's own duration, which the roofline block
        # reports, is then two to three times what it costs the chip: `frame` is the figure that counts.
        ctx.set_frames_in_flight(in_flight)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(stats=False, i=i)
        t_submit = time.perf_counter() - t0    # host time to queue the frames (the GPU is still running)
        ctx.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ctx.set_frames_in_flight(1)
        rays = st["rays"] * args.steps
        bounces = st["bounces"] * args.steps
        tail_ms, tail_launches = ctx.profile_get("tail")          # the dominant kernel of the frame
        ctx.profile_enable(False)
        tot = torch.tensor([float(rays), dt], dtype=torch.float64, device=cdev)
        if world > 1:
            rays_all = tot[0:1].clone()
            dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
            tmax = tot[1:2].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            rays_total, t_total = float(rays_all.item()), float(tmax.item())
            fr = frame.to(cdev)
            frames = [torch.empty_like(fr) for _ in range(world)] if rank == 0 else None
            dist.gather(fr, frames, dst=0)                                     # frames to rank 0 (outside the timed region)
        else:
            rays_total, t_total = float(rays), dt
        tail_rays = (rays - W * H * args.steps) // args.steps
        parallelism = f"{world} independent frame(s), one per GPU"
        scaling = "weak"
    else:
        # ---- ONE job over all ranks: sharded emission -> all-gather -> replicated build -> band-split frames -> gather
        in_flight = 1
        per_gpu = args.photons_per_gpu
        nd = nc = (per_gpu * world) // 2
        pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=nd, num_photons_caustic=nc, seed=seed)
        if rehearsal:
            eng = _RehearsalEngine(sc, pe, device=dev)
        else:
            eng = parallel.DeviceEngine(sc, pe, fmt=capi.SHARD_PACKED, device=dev)
        T = {}
        barrier()
        t0 = time.perf_counter()
        res = parallel.emit_sharded(eng, batch=min(world * (8 << 20), 1 << 30), timings=T)
        barrier()
        t_emit = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.build_maps()
        barrier()
        t_build = time.perf_counter() - t0
        # frames of config[1], every frame's row bands interleaved over the ranks and gathered to rank 0
        reng = parallel.DeviceEngine(sc, p, device=dev, with_lists=False)      # config[1] frames: no maps, no lists
        if rehearsal:
            reng = _RehearsalEngine(sc, p, device=dev, inner=reng)
        st_full = sc.trace_tile(p, want=())["stats"]                     # the whole frame once, for its segment count (every rank, same number)
        # frames go to rank 0 in batches: every rank traces its bands of up to G frames back to back (four in flight), then
        # ONE gather brings them together -- a frame's bands take tens of microseconds, less than a collective's latency
        G = max(1, args.frames_per_gather)

        def split_frames(engine, Wd, Hd, n_frames, group, n_warm):
            """n_frames frames of the engine's scene, each traced as interleaved row bands (one share per rank) and gathered
            to rank 0 in groups of `group`; seconds for all of them, the maximum over the ranks"""
            rows_ = ctx.band_rows(Hd, args.band, world, rank)
            bufs_ = {}
            frames0_ = torch.empty((group, Hd, Wd, 4), dtype=torch.uint8, device=cdev) if rank == 0 else None

            def frame_groups(sizes):
                """the groups one after the other, a group's gather running beside the next group's tracing (two band buffers)"""
                pending = None
                for k, n in enumerate(sizes):
                    if (n, k & 1) not in bufs_:
                        bufs_[(n, k & 1)] = torch.empty((n, rows_, Wd, 4), dtype=torch.uint8, device=dev)
                    engine.out = bufs_[(n, k & 1)]
                    parts = parallel.trace_bands_batch(engine, Wd, Hd, n, band=args.band)
                    if pending is not None:
                        parallel.gather_bands_end(pending[0], Wd, Hd, band=args.band, out=frames0_[:pending[1]] if rank == 0 else None)
                    pending = (parallel.gather_bands_begin(parts, Hd, band=args.band), n)
                if pending is not None:
                    parallel.gather_bands_end(pending[0], Wd, Hd, band=args.band, out=frames0_[:pending[1]] if rank == 0 else None)

            sizes_of = lambda total: [min(group, total - g0) for g0 in range(0, total, group)]
            for k, n in enumerate(sizes_of(n_frames)):  # (every band buffer of the timed run exists before the clock starts)
                if (n, k & 1) not in bufs_:
                    bufs_[(n, k & 1)] = torch.empty((n, rows_, Wd, 4), dtype=torch.uint8, device=dev)
            frame_groups(sizes_of(max(n_warm, 1)))
            barrier()
            t0_ = time.perf_counter()
            frame_groups(sizes_of(n_frames))
            barrier()
            dt_ = time.perf_counter() - t0_
            tm_ = torch.tensor([dt_], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
            return float(tm_.item())

        t_total = split_frames(reng, W, H, args.steps, G, args.warmup)
        dt = t_total
        # ---- the two configurations BASELINE names for 8 GPUs, in the same split job (reported under "phases", never `value`)
        # configs[4]'s full render: direct light and both radius gathers on the all-gathered maps this job has just built
        pf = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=nd, num_photons_caustic=nc, seed=seed, shadow_bias=1e-4)
        feng = parallel.DeviceEngine(sc, pf, device=dev, with_lists=False)
        feng.maps = eng.maps
        st5 = sc.trace_tile(pf, eng.maps[0], eng.maps[1], want=())["stats"]
        if rehearsal:
            feng = _RehearsalEngine(sc, pf, device=dev, inner=feng)
        n5 = max(1, min(args.steps, 4))
        t5 = split_frames(feng, W, H, n5, min(G, n5), 1)
        # configs[3]: the 10 012-primitive scene at 3840x2160, paths of up to 8 segments
        W4, H4 = (3840, 2160) if not rehearsal else (960, 540)
        flat4 = synth.many_primitive_scene(10000, seed=42, width=W4, height=H4)
        sc4 = ctx.upload(flat4)
        p4 = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=seed, sub_sample=1, max_bounces=8)
        eng4 = parallel.DeviceEngine(sc4, p4, device=dev, with_lists=False)
        st4 = sc4.trace_tile(p4, want=())["stats"]
        if rehearsal:
            eng4 = _RehearsalEngine(sc4, p4, device=dev, inner=eng4)
        n4 = max(1, min(args.steps, 8))
        t4 = split_frames(eng4, W4, H4, n4, min(2, n4), 2)
        del eng4, sc4
        rays = st_full["rays"] * args.steps
        rays_total = float(rays)
        bounces = st_full["bounces"] * args.steps
        tail_rays = st_full["tail_rays"]
        tail_ms = tail_launches = tx_ms = tx_n = px_ms = px_n = 0
        t_submit = 0.0
        stored = res["n_diffuse"] + res["n_caustic"]
        ag_bytes = T.get("allgather_bytes", 0)
        phases = {
            "emission": {"seconds": t_emit, "stored_photons": stored, "stored_photons_per_s": stored / t_emit, "attempts": res["attempts1"] + res["attempts2"],
                         "rounds": T.get("rounds"), "emit_kernel_s": T.get("emit_s"), "exchange_s": T.get("exchange_s"), "caps_s": T.get("apply_s"),
                         "record_bytes": 40, "note": "36-byte packed photon + 4-byte tag per stored photon, all-gathered from and into device memory"},
            "allgather_bytes": ag_bytes, "allgather_gbps": (ag_bytes / T["exchange_s"] / 1e9) if T.get("exchange_s") else None,
            "kd_build": {"seconds": t_build, "photons": stored, "note": "both maps, built identically on every rank from its own copy of the lists"},
            "render": {"seconds": dt, "frames": args.steps, "ms_per_frame": dt / args.steps * 1e3, "band_rows": args.band,
                       "gather_bytes_per_frame": W * H * 4, "frames_per_gather": G,
                       "note": "every frame: this rank's interleaved bands in one rt_trace_tile call (queued, four in flight); the bands of frames_per_gather frames gathered to rank 0 by one collective"},
            "config5_render": {"seconds": t5, "frames": n5, "ms_per_frame": t5 / n5 * 1e3, "Mrays_per_s": st5["rays"] * n5 / t5 / 1e6, "rays_per_frame": st5["rays"],
                               "shadow_rays_per_frame": st5["shadow_rays"], "gathers_per_frame": st5["gathers"], "gathers_per_s": st5["gathers"] * n5 / t5,
                               "photons": stored, "resolution": [W, H], "band_rows": args.band, "gather_bytes_per_frame": W * H * 4,
                               "note": "BASELINE configs[4]'s full render on the maps of this job (emitted in shards, all-gathered, built on every rank): direct light, diffuse (r = 0.40) "
                                       "and caustic (r = 0.15) radius gathers, every frame's interleaved row bands traced on their rank and gathered to rank 0"},
            "config4": {"seconds": t4, "frames": n4, "ms_per_frame": t4 / n4 * 1e3, "Mrays_per_s": st4["rays"] * n4 / t4 / 1e6, "rays_per_frame": st4["rays"],
                        "primitives": int(flat4.n_shapes), "resolution": [W4, H4], "band_rows": args.band, "gather_bytes_per_frame": W4 * H4 * 4, "frames_per_gather": min(2, n4),
                        "prim_tests_per_frame": st4["prim_tests"], "tests_avoided_frac": 1.0 - st4["prim_tests"] / max(1, st4["rays"] * st4["prims_per_ray_brute_force"]),
                        "note": f"BASELINE configs[3]: the synthetic 10 012-primitive scene at {W4}x{H4}, paths of up to 8 segments, through the hierarchy; "
                                "interleaved row bands per rank, gathered to rank 0"},
        }
        parallelism = f"one job over {world} GPUs: emission by attempt range, photon all-gather, replicated kd build, frames split into interleaved {args.band}-row bands gathered to rank 0"
        scaling = "strong"

    if split:
        extra, roof_gather, roof_build, roof_emit = other_legs()

    if rank == 0:
        line = {
            "metric": "Mrays/s (primary+secondary)", "value": rays_total / t_total / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_total / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"reference default scene (preset 10, {flat.n_shapes} primitives) at {W}x{H}, 1 spp, "
                                   "primary + reflection/refraction segments only (direct light and photon gathers off)",
                       "parallelism": parallelism, "frames_in_flight_per_gpu": in_flight, "rays_per_step": int(rays // args.steps),
                       "longest_path_segments": int(bounces // args.steps)},
        }
        if not split:
            t_avg = tail_ms / max(1, tail_launches) * 1e-3
            tail_flop = tail_rays * fl
            ach = tail_flop / max(t_avg, 1e-12) / 1e12
            tx_avg, px_avg = tx_ms / max(1, tx_n) * 1e-3, px_ms / max(1, px_n) * 1e-3
            frame_ach = (rays // args.steps) * fl / (t_total / args.steps) / 1e12
            line["roofline"] = {
                "kernel": ("k_tail_persistent<float> (every path that survives its first hit, to its end: closest-hit scan (every primitive, less the groups "
                           "no ray of the wave reaches) + shade per segment; a lane whose path has ended takes the next queued path of its workgroup's chunk)" if in_flight >= 3 else
                           "k_tail<float> (every path that survives its first hit, to its end: closest-hit scan (every primitive, less the groups no ray of the wave reaches) + shade per segment, "
                           "256-lane groups with LDS-resident path state)") + " -- the longer of the frame's two kernels",
                "bound": "valu", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_VALU_TFLOPS,
                "traffic": measured_traffic("frames", "k_tail_persistent" if in_flight >= 3 else "k_tail<float"), "algorithmic_flop_per_ray_pass": fl, "avg_launch_ms": t_avg * 1e3,
                "executed": executed_block("frames", "k_tail_persistent" if in_flight >= 3 else "k_tail<float", t_avg),
                "segments_per_launch": int(tail_rays), "launches_per_step": 1.0, "timed_launches": int(tail_launches),
                "primitive_tests_per_segment_executed": st["prim_tests"] / max(1, st["rays"]),
                "note": "the timed frames overlap (four frames in flight), so a launch shares the chip with the neighbouring frames' kernels and its duration is not its cost; "
                        "`exclusive` has both kernels on frames run one at a time; `frame` is the whole frame against the driver's clock. "
                        "bound 'valu': this path is neither HBM- nor MFMA-bound; peak = fp32 vector FMA rate (the measured sustained v_fma_f32 "
                        "rate on this chip is 106 TFLOP/s, profiles/r02_valu_issue_rate.txt)",
                "exclusive": {"note": "frames one at a time, where the library uses the compacting tail kernel k_tail<float> (it finishes a lone frame sooner)",
                              "tail": {"avg_launch_ms": tx_avg * 1e3, "achieved": tail_flop / max(tx_avg, 1e-12) / 1e12,
                                       "frac": tail_flop / max(tx_avg, 1e-12) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                                       "traffic": measured_traffic("frames", "k_tail<float"), "executed": executed_block("frames", "k_tail<float", tx_avg)},
                              "primary": {"avg_launch_ms": px_avg * 1e3, "rays_per_launch": W * H, "achieved": W * H * fl / max(px_avg, 1e-12) / 1e12,
                                          "frac": W * H * fl / max(px_avg, 1e-12) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                                          "traffic": measured_traffic("frames", "k_primary<float"), "executed": executed_block("frames", "k_primary<float", px_avg)}},
                "frame": {"achieved": frame_ach, "frac": frame_ach / PEAK_FP32_VALU_TFLOPS, "ms": t_total / args.steps * 1e3,
                          "executed": frame_executed(in_flight, t_total / args.steps)},
                "host_submit_ms_per_frame": t_submit / args.steps * 1e3}
        else:
            frame_ach = (rays // args.steps) * fl / (t_total / args.steps) / 1e12
            line["roofline"] = {"kernel": "whole band-split frame (k_primary + k_tail on every rank, plus the gather to rank 0)", "bound": "valu",
                                "achieved": frame_ach, "peak": PEAK_FP32_VALU_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": frame_ach / (PEAK_FP32_VALU_TFLOPS * world), "traffic": None}
            line["phases"] = phases
            line["allgather_gbps"] = phases["allgather_gbps"]
        if roof_gather:
            line["roofline_gather"] = roof_gather
        if roof_build:
            line["roofline_build"] = roof_build
        if roof_emit:
            line["roofline_emit"] = roof_emit
        line["extra"] = extra
        if cpu_line:
            line["cpu_baseline"] = cpu_line
        if cpu_gather_line:
            line["cpu_baseline_gather"] = cpu_gather_line
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


class _RehearsalEngine:
    """rehearsal only (all ranks on one GPU, gloo): the device engine with its shards moved to CPU tensors for the collectives"""

    def __init__(self, scene, params, device, inner=None):
        from raytracer_amd import capi, parallel
        self.inner = inner if inner is not None else parallel.DeviceEngine(scene, params, fmt=capi.SHARD_PACKED, device=device)
        self.device = device
        self.out = None

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def emit_shard(self, phase, a0, a1, tag_base):
        sh, n = self.inner.emit_shard(phase, a0, a1, tag_base)
        self.inner.sync()
        return sh.cpu(), n.cpu()

    def apply(self, phase, tag_base, merged, counts, n_shards, stride):
        return self.inner.apply(phase, tag_base, merged.to(self.device), counts.to(self.device), n_shards, stride)

    def render_bands(self, width, height, band, world, rank):
        self.inner.out = self.out
        return self.inner.render_bands(width, height, band, world, rank).cpu()

    def render_bands_batch(self, width, height, band, world, rank, n_frames):
        self.inner.out = self.out
        return self.inner.render_bands_batch(width, height, band, world, rank, n_frames).cpu()


if __name__ == "__main__":
    main()
