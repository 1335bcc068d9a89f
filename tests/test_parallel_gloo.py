"""World-size-2 test of the sharded emission / render protocol on CPU (gloo), with the CPU oracle as
each rank's engine: the merged result must be exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_io import load_scene
from oracle import oracle as ora
from raytracer_amd import parallel

ND, NC, W, H, SEED = 300, 200, 64, 48, 99


class OracleEngine:
    def __init__(self, maps=None):
        self.flat = load_scene(10, W, H)
        self.sc = ora.Scene(self.flat)
        self.p = ora.params(num_photons_diffuse=ND, num_photons_caustic=NC, seed=SEED)
        self.maps = maps

    def emit(self, phase, a0, a1):
        if a1 <= a0:
            return np.zeros((0, 11)), np.zeros(0, np.int64), np.zeros(0, np.uint8)
        flag, rec = ora.emit_attempts(self.sc, self.p, phase, a0, a1)
        keep = flag > 0
        return rec[keep], (np.nonzero(keep)[0] + a0).astype(np.int64), flag[keep].astype(np.uint8)

    def render(self, x0, y0, w, h):
        return ora.render(self.sc, self.p, self.maps, x0, y0, w, h)[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleEngine()
        res = parallel.emit_sharded(eng, ND, NC, batch=257)          # odd batch: ragged shards every round
        eng.maps = ora.Maps(res["diffuse"], res["caustic"])
        frame = parallel.render_sharded(eng, W, H, band=5)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), frame=frame if frame is not None else np.zeros(0), **res)
    finally:
        dist.destroy_process_group()


def test_emit_state_caps_match_sequential_loop():
    eng = OracleEngine()
    single = ora.emit_many(eng.sc, eng.p)
    for batch in (1, 7, 64, 100000):
        res = parallel.emit_sharded(eng, ND, NC, batch=batch)        # world size 1 path
        assert np.array_equal(res["diffuse"], single["diffuse"]) and np.array_equal(res["caustic"], single["caustic"])
        assert np.array_equal(res["diffuse_attempt"], single["diffuse_tag"][:, 1:])
        assert res["attempts1"] == single["attempts1"] and res["attempts2"] == single["attempts2"] and not res["capped"]


def test_band_plan_interleaves_rows():
    plan = parallel.band_plan(50, 3, band=16)
    assert plan == [(0, 16, 0), (16, 16, 1), (32, 16, 2), (48, 2, 0)]
    assert sum(h for _, h, _ in plan) == 50


@pytest.mark.timeout(300)
def test_world_size_2_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    eng = OracleEngine()
    single = ora.emit_many(eng.sc, eng.p)
    for r in (r0, r1):                                               # every rank holds the full, identical lists
        assert np.array_equal(r["diffuse"], single["diffuse"]) and np.array_equal(r["caustic"], single["caustic"])
        assert np.array_equal(r["caustic_attempt"], single["caustic_tag"][:, 1:])
        assert int(r["attempts1"]) == single["attempts1"] and int(r["attempts2"]) == single["attempts2"]
    maps = ora.Maps(single["diffuse"], single["caustic"])
    _, post, _, _ = ora.render(eng.sc, eng.p, maps, 0, 0, W, H)
    got = r0["frame"]
    assert got.shape == post.shape
    assert ((got == post) | (np.isnan(got) & np.isnan(post))).all()
    assert r1["frame"].size == 0
