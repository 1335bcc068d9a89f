#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary segments) of the wavefront tracer on the
reference's default scene at 1920x1080, 1 spp, reflection/refraction only -- plus, as extra
fields of the same JSON line, photon k-NN gathers/s on a 1M-photon map.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)

A step is one full-frame pass of the hot path.  With N ranks every rank renders its own frame
(same scene, its own RNG key: independent sub-sample passes), so scaling is weak and no
collective sits on the data path; frames are gathered to rank 0 after the timed region only.
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
PEAK_FP32_VALU_TFLOPS = 157.3      # 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz
PEAK_HBM_GBS = 8000.0
BYTES_PER_GATHER = 64 * 36 + 36 + 12


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (rocprofv3 cannot run
    inside the benchmark): profiles/r01_c_traffic.json, written by tools/profile_round.sh + tools/make_profiles.py."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_c_traffic.json")) as f:
            return json.load(f)[kernel]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def flop_per_ray_pass(flat):
    return int(sum(FLOP_PER_TEST[int(t)] for t in flat.shape_meta[:, 0]))


def cpu_baseline(flat, seed, budget_s=12.0):
    """The CPU oracle (fp64 restatement of the reference) on the same workload, one host core: whole frames with
    successive seeds until about budget_s seconds have been spent."""
    from oracle import oracle as ora
    osc = ora.Scene(flat)
    rays_total, frames = 0, 0
    t0 = time.perf_counter()
    while True:
        op = ora.params(seed=seed + frames, use_direct_light=0, use_photon_maps=0)
        _, _, rays, _ = ora.render(osc, op, None, 0, 0, flat.width, flat.height)
        rays_total += rays
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 64:
            break
    return {"value": rays_total / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{frames} full {flat.width}x{flat.height} frames of the same workload (seeds {seed}..{seed + frames - 1}), "
                      f"{rays_total} rays in {dt:.2f} s, oracle/rt_oracle.c (fp64, bit-exact with the reference JS on its golden frames)"}


def cpu_baseline_gather(rec, hits, k, n_queries=100_000):
    """CPU side of the photon-map leg: the oracle's kd-tree (the reference's makeTree / nearestNeighbours restated in C)
    built on the same photons, and its radius query at the radius that holds k photons on average -- the reference has
    no k-nearest query -- for a sample of the same hit points, one core."""
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
    t0 = time.perf_counter()
    found = 0
    for i in range(q.shape[0]):
        found += ora.kd_radius(pos, perm, dim, q[i], r, cap=1024).size
    dt = time.perf_counter() - t0
    return {"value": q.shape[0] / dt, "unit": "gathers/s", "cores": 1, "kind": "port", "kd_build_s": build_s,
            "sample": f"{q.shape[0]} of the {hits.shape[0]} hit points, radius query r={r:.4f} ({found / q.shape[0]:.1f} photons per query) "
                      f"on the oracle's kd-tree of the same {pos.shape[0]} photons, {dt:.2f} s"}


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
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs for the older tracer, the 10k-primitive scene and photon emission")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    from raytracer_amd import capi, synth
    from raytracer_amd.scene import FlatScene

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = capi.Context(local)
    W, H = args.width, args.height
    flat = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_preset10_flat.json")).with_size(W, H)
    sc = ctx.upload(flat)
    seed = 1 + rank
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=seed, sub_sample=1)
    in_flight = 1 if os.environ.get("RT355_BENCH_SERIAL_FRAMES") else int(os.environ.get("RT355_BENCH_IN_FLIGHT", "2"))
    frames_out = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(in_flight)]   # outputs stay in HBM
    frame = frames_out[0]

    def step(stats=True, i=0):
        return sc.trace_tile(p, out_rgba8=frames_out[i % in_flight], want=(), stats=stats)["stats"]

    ctx.profile_enable(os.environ.get("RT355_BENCH_NO_EVENTS") is None)
    ctx.profile_select(None if os.environ.get("RT355_BENCH_ALL_EVENTS") else "primary")   # time the roofline kernel only: every event pair costs ~8 us of stream time
    for _ in range(max(args.warmup, 1)):
        st = step()                         # every step renders the same frame (same seed): its ray count is this one's
    ctx.profile_reset()                     # (the warm-up's timing events go back to the library's pool)
    # the timed frames are only queued (no statistics read back in between: that would put a host round trip after
    # every frame) and the clock stops after the library's streams have drained
    # Two frames in flight: consecutive frames go to alternate streams, workspaces and output buffers, so the thinly
    # occupied end of one frame's path tracing overlaps the start of the next.
    ctx.set_frames_in_flight(in_flight)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(stats=False, i=i)
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ctx.set_frames_in_flight(1)
    rays = st["rays"] * args.steps
    bounces = st["bounces"] * args.steps
    k_ms, k_launches = ctx.profile_get("primary")        # the fused first pass: the roofline kernel
    # for the report only, a few extra steps outside the timed region, one frame at a time: the tail kernel's duration,
    # and the first pass's when it has the chip to itself (in the timed region it shares it with the previous frame's tail)
    ctx.profile_select("primary,tail")
    ctx.profile_reset()
    for _ in range(3):
        step()
    tail_ms, tail_launches = ctx.profile_get("tail")
    tail_ms = tail_ms / max(1, tail_launches) * args.steps
    kx_ms, kx_launches = ctx.profile_get("primary")
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
        dist.gather(fr, frames, dst=0)                                     # tiles/frames to rank 0 (outside the timed region)
    else:
        rays_total, t_total = float(rays), dt

    # ---- photon-map leg: 1M-photon kd-tree build + k-NN radiance gather per pixel ----
    extra = {}
    roof_gather = None
    if not args.no_gather:                        # (every rank runs the report-only legs: they hold barriers)
        rec_host, hits_host = synth.box_photons(args.photons, seed=7), synth.box_queries(W, H)
        rec = torch.from_numpy(rec_host).to(dev)
        hits = torch.from_numpy(hits_host).to(dev)
        rgb = torch.empty((hits.shape[0], 3), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        pm = ctx.photon_map(rec)
        build_s = time.perf_counter() - tb
        pm.gather(hits, k=args.knn, out=rgb)                               # warm-up
        ctx.profile_enable(True)
        ctx.profile_select("gather_knn,gather_knn_heap_fallback")
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
        ctx.profile_enable(False)
        gathers = hits.shape[0] * gsteps
        gps = torch.tensor([gathers / gdt], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(gps, op=dist.ReduceOp.SUM)
        g_achieved = hits.shape[0] * BYTES_PER_GATHER / (g_ms / g_launches * 1e-3) / 1e9
        extra = {"knn_gathers_per_s": float(gps.item()), "knn_k": args.knn, "photons": args.photons,
                 "kd_build_s": build_s, "gather_ms_per_step": gdt / gsteps * 1e3,
                 "heap_fallback_ms_per_step": fb_ms / gsteps, "heap_fallback_launches": fb_launches}
        roof_gather = {"kernel": "k_gather_knn_wave", "bound": "hbm", "achieved": g_achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": g_achieved / PEAK_HBM_GBS, "traffic": measured_traffic("k_gather_knn_wave"),
                       "algorithmic_bytes_per_gather": BYTES_PER_GATHER, "avg_launch_ms": g_ms / g_launches}

    # ---- short legs for the other BASELINE configs (reported under "extra", never part of `value`) ----
    if not args.no_other_configs:
        def timed_frames(scene, params, out, n=5):
            for _ in range(2):
                stt = scene.trace_tile(params, out_rgba8=out, want=())["stats"]
            ctx.synchronize()
            t_ = time.perf_counter()
            for _ in range(n):
                stt = scene.trace_tile(params, out_rgba8=out, want=())["stats"]
            ctx.synchronize()
            return (time.perf_counter() - t_) / n, stt
        # the older tracer (ray.js) on its preset 3 at the same resolution
        flat_a = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", "scene_rayjs3_flat.json")).with_size(W, H)
        pa = capi.default_params(precision=capi.PRECISION_F32, mode=capi.MODE_WHITTED, seed=seed, max_trace_dist=100.0, max_depth=20, sub_sample=1)
        dt_a, st_a = timed_frames(ctx.upload(flat_a), pa, frame)
        extra["mode_a_rayjs_preset3"] = {"Mrays_per_s": st_a["rays"] / dt_a / 1e6, "ms_per_frame": dt_a * 1e3, "rays_per_frame": st_a["rays"],
                                         "primitives": int(flat_a.n_shapes), "resolution": [W, H]}
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
        # photon emission (the configs[4] kernel on one GPU's share): 2 M diffuse + 2 M caustic photons of the default scene
        pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=seed)
        sc.emit_many(pe, strict=False)
        t_ = time.perf_counter()
        em = sc.emit_many(pe, strict=False)["stats"]
        dte = time.perf_counter() - t_
        extra["photon_emission"] = {"stored_photons_per_s": (em["n_diffuse"] + em["n_caustic"]) / dte, "segments_per_s": em["segments"] / dte,
                                    "attempts": em["attempts_phase1"] + em["attempts_phase2"], "seconds": dte,
                                    "note": "includes copying the 11-double records to host memory"}

    if rank == 0:
        fl = flop_per_ray_pass(flat)
        achieved = (W * H) * fl / ((max(k_ms, 1e-9) / max(1, k_launches)) * 1e-3) / 1e12     # TFLOP/s while the kernel runs
        line = {
            "metric": "Mrays/s (primary+secondary)", "value": rays_total / t_total / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_total / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"reference default scene (preset 10, {flat.n_shapes} primitives) at {W}x{H}, 1 spp, "
                                   "primary + reflection/refraction segments only (direct light and photon gathers off)",
                       "parallelism": f"{world} independent frame(s), one per GPU", "frames_in_flight_per_gpu": in_flight, "rays_per_step": rays // args.steps,
                       "longest_path_segments": bounces // args.steps},
            "roofline": {"kernel": "k_primary<float> (camera ray + closest hit among the primitives its workgroup's rays can reach + shade, one lane per primary ray; algorithmic flops count the all-primitives scan the reference does)", "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": measured_traffic("k_primary"),
                         "algorithmic_flop_per_ray_pass": fl, "avg_launch_ms": k_ms / max(1, k_launches),
                         "launches_per_step": k_launches / args.steps, "kernel_ms_per_step": k_ms / args.steps,
                         "rays_per_launch": W * H, "tail_kernel_ms_per_step": tail_ms / args.steps,
                         "tail_rays_per_step": (rays - W * H * args.steps) // args.steps,
                         "note": "the timed frames overlap in pairs (two frames in flight), so this kernel's launches share the chip "
                                 "with the previous frame's tail kernel; `exclusive` is the same kernel over frames run one at a time",
                         "exclusive": {"avg_launch_ms": kx_ms / max(1, kx_launches),
                                       "achieved": (W * H) * fl / ((max(kx_ms, 1e-9) / max(1, kx_launches)) * 1e-3) / 1e12,
                                       "frac": (W * H) * fl / ((max(kx_ms, 1e-9) / max(1, kx_launches)) * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS}},
        }
        if roof_gather:
            line["roofline_gather"] = roof_gather
        line["extra"] = extra
        if not args.no_cpu_baseline and world == 1:         # (the CPU baseline is reported with the single-GPU line only)
            line["cpu_baseline"] = cpu_baseline(flat, seed)
            if not args.no_gather:
                line["cpu_baseline_gather"] = cpu_baseline_gather(rec_host, hits_host, args.knn)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
