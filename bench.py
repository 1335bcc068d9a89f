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


def flop_per_ray_pass(flat):
    return int(sum(FLOP_PER_TEST[int(t)] for t in flat.shape_meta[:, 0]))


def cpu_baseline(flat, seed):
    """The CPU oracle (fp64 restatement of the reference) on the same frame, one host core."""
    from oracle import oracle as ora
    osc = ora.Scene(flat)
    op = ora.params(seed=seed, use_direct_light=0, use_photon_maps=0)
    rows = flat.height
    t0 = time.perf_counter()
    _, _, rays, _ = ora.render(osc, op, None, 0, 0, flat.width, rows)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"the full {flat.width}x{rows} frame of the same workload, {rays} rays in {dt:.2f} s, "
                      "oracle/rt_oracle.c (fp64, bit-exact with the reference JS on its golden frame)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--photons", type=int, default=1_000_000)
    ap.add_argument("--knn", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the photon-map (k-NN gather) leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

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
    frame = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)          # output stays in HBM

    def step():
        return sc.trace_tile(p, out_rgba8=frame, want=())["stats"]

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(os.environ.get("RT355_BENCH_NO_EVENTS") is None)
    ctx.profile_select(None if os.environ.get("RT355_BENCH_ALL_EVENTS") else "primary,tail")   # time the two trace kernels only
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    bounces = 0
    for _ in range(args.steps):
        st = step()
        rays += st["rays"]
        bounces += st["bounces"]
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    k_ms, k_launches = ctx.profile_get("primary")        # the fused first pass: the roofline kernel
    tail_ms, _ = ctx.profile_get("tail")
    ctx.profile_enable(False)

    tot = torch.tensor([float(rays), dt], dtype=torch.float64, device=dev)
    if world > 1:
        rays_all = tot[0:1].clone()
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        tmax = tot[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays_total, t_total = float(rays_all.item()), float(tmax.item())
        frames = [torch.empty_like(frame) for _ in range(world)] if rank == 0 else None
        dist.gather(frame, frames, dst=0)                                  # tiles/frames to rank 0 (outside the timed region)
    else:
        rays_total, t_total = float(rays), dt

    # ---- photon-map leg: 1M-photon kd-tree build + k-NN radiance gather per pixel ----
    extra = {}
    roof_gather = None
    if not args.no_gather:
        rec = torch.from_numpy(synth.box_photons(args.photons, seed=7)).to(dev)
        hits = torch.from_numpy(synth.box_queries(W, H)).to(dev)
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
        gps = torch.tensor([gathers / gdt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(gps, op=dist.ReduceOp.SUM)
        g_achieved = hits.shape[0] * BYTES_PER_GATHER / (g_ms / g_launches * 1e-3) / 1e9
        extra = {"knn_gathers_per_s": float(gps.item()), "knn_k": args.knn, "photons": args.photons,
                 "kd_build_s": build_s, "gather_ms_per_step": gdt / gsteps * 1e3,
                 "heap_fallback_ms_per_step": fb_ms / gsteps, "heap_fallback_launches": fb_launches}
        roof_gather = {"kernel": "k_gather_knn_wave", "bound": "hbm", "achieved": g_achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": g_achieved / PEAK_HBM_GBS, "traffic": None,
                       "algorithmic_bytes_per_gather": BYTES_PER_GATHER, "avg_launch_ms": g_ms / g_launches}

    if rank == 0:
        fl = flop_per_ray_pass(flat)
        achieved = (W * H) * fl / ((max(k_ms, 1e-9) / max(1, k_launches)) * 1e-3) / 1e12     # TFLOP/s while the kernel runs
        line = {
            "metric": "Mrays/s (primary+secondary)", "value": rays_total / t_total / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_total / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"reference default scene (preset 10, {flat.n_shapes} primitives) at {W}x{H}, 1 spp, "
                                   "primary + reflection/refraction segments only (direct light and photon gathers off)",
                       "parallelism": f"{world} independent frame(s), one per GPU", "rays_per_step": rays // args.steps,
                       "longest_path_segments": bounces // args.steps},
            "roofline": {"kernel": "k_primary<float> (camera ray + closest-hit scan of all primitives + shade, one lane per primary ray)", "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": None,
                         "algorithmic_flop_per_ray_pass": fl, "avg_launch_ms": k_ms / max(1, k_launches),
                         "launches_per_step": k_launches / args.steps, "kernel_ms_per_step": k_ms / args.steps,
                         "rays_per_launch": W * H, "tail_kernel_ms_per_step": tail_ms / args.steps,
                         "tail_rays_per_step": (rays - W * H * args.steps) // args.steps},
        }
        if roof_gather:
            line["roofline_gather"] = roof_gather
        line["extra"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(flat, seed)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
