"""World-size-2 test of the sharded emission / render protocol on CPU (gloo), with the CPU oracle as each rank's
engine behind the same tensor interface the GPU engine has: the merged result must be exactly the single-process
result."""
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
WORDS = 22                                               # full fp64 records: 11 doubles = 22 words, + 1 tag word


class OracleEngine:
    """the oracle behind the engine interface of raytracer_amd.parallel; shards are CPU int32 tensors in the layout of
    rt_emit_shard (RT_SHARD_FULL, fp64)"""

    def __init__(self):
        self.flat = load_scene(10, W, H)
        self.sc = ora.Scene(self.flat)
        self.p = ora.params(num_photons_diffuse=ND, num_photons_caustic=NC, seed=SEED)
        self.state = parallel.EmitState(ND, NC)
        self.maps = None

    def caps(self):
        return ND, NC

    def counts(self):
        return self.state.nd, self.state.nc

    def phase_done(self, phase):
        return self.state.nd >= ND if phase == 1 else self.state.nc >= NC

    def sync(self):
        pass

    def emit_shard(self, phase, a0, a1, tag_base):
        cap = max(a1 - a0, 1)
        shard = np.zeros((cap, WORDS + 1), np.int32)
        n = 0
        if a1 > a0:
            flag, rec = ora.emit_attempts(self.sc, self.p, phase, a0, a1)
            keep = np.nonzero(flag > 0)[0]
            n = keep.size
            shard[:n, :WORDS] = np.ascontiguousarray(rec[keep]).view(np.int32).reshape(n, WORDS)
            tag = (keep + (a0 - tag_base)).astype(np.uint32) | np.where(flag[keep] == 2, np.uint32(1 << 31), np.uint32(0))
            shard[:n, WORDS] = tag.view(np.int32)
        return torch.from_numpy(shard), torch.tensor([n], dtype=torch.int32)

    def apply(self, phase, tag_base, merged, counts, n_shards, stride):
        merged, counts = merged.numpy(), counts.numpy()
        rows = np.concatenate([merged[k, :counts[k]] for k in range(n_shards)])
        rec = np.ascontiguousarray(rows[:, :WORDS]).view(np.float64).reshape(-1, 11)
        tag = np.ascontiguousarray(rows[:, WORDS]).view(np.uint32)
        attempt = (tag & 0x7FFFFFFF).astype(np.int64) + tag_base
        assert (np.diff(attempt) > 0).all()                        # the merged batch is in attempt order
        kind = np.where(tag >> 31, 2, 1).astype(np.uint8)
        return self.state.apply(phase, tag_base, rec, attempt, kind)

    def render_bands(self, width, height, band, world, rank):
        rows = parallel.band_rows_of(height, band, world, rank)
        out = np.zeros((rows.size, width, 4))
        y = 0
        while y < rows.size:                                       # one oracle call per band
            h = 1
            while y + h < rows.size and rows[y + h] == rows[y] + h:
                h += 1
            out[y:y + h] = ora.render(self.sc, self.p, self.maps, 0, int(rows[y]), width, h)[1]
            y += h
        return torch.from_numpy(out)


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
        T = {}
        res = parallel.emit_sharded(eng, batch=257, timings=T)       # odd batch: ragged shards every round
        lists = eng.state.result()
        eng.maps = ora.Maps(lists["diffuse"], lists["caustic"])
        frame = parallel.render_sharded(eng, W, H, band=8)
        batch = parallel.render_sharded_batch(eng, W, H, 3, band=8)  # three frames' bands in one collective
        # two batches with the first one's gather left running while the second is traced
        h1 = parallel.gather_bands_begin(parallel.trace_bands_batch(eng, W, H, 2, band=8), H, band=8)
        parts2 = parallel.trace_bands_batch(eng, W, H, 1, band=8)
        over1 = parallel.gather_bands_end(h1, W, H, band=8)
        over2 = parallel.gather_bands_end(parallel.gather_bands_begin(parts2, H, band=8), W, H, band=8)
        over = torch.cat([over1, over2]) if over1 is not None else None
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), frame=frame.numpy() if frame is not None else np.zeros(0),
                 batch=batch.numpy() if batch is not None else np.zeros(0), over=over.numpy() if over is not None else np.zeros(0),
                 attempts1=res["attempts1"], attempts2=res["attempts2"], allgather_bytes=T["allgather_bytes"], **lists)
    finally:
        dist.destroy_process_group()


def test_emit_state_caps_match_sequential_loop():
    single = None
    for batch, sim in ((1, None), (7, 3), (64, 2), (100000, 5), (100000, None)):
        eng = OracleEngine()
        if single is None:
            single = ora.emit_many(eng.sc, eng.p)
        res = parallel.emit_sharded(eng, batch=batch, world_override=sim)     # world size 1; `sim` ranks played in turn
        lists = eng.state.result()
        assert np.array_equal(lists["diffuse"], single["diffuse"]) and np.array_equal(lists["caustic"], single["caustic"])
        assert np.array_equal(lists["diffuse_attempt"], single["diffuse_tag"][:, 1:])
        assert res["attempts1"] == single["attempts1"] and res["attempts2"] == single["attempts2"] and not res["capped"]


def test_band_rows():
    assert parallel.band_rows_of(50, 16, 3, 0).tolist() == list(range(0, 16)) + [48, 49]
    assert parallel.band_rows_of(50, 16, 3, 2).tolist() == list(range(32, 48))
    assert sum(parallel.band_rows_of(563, 8, 8, k).size for k in range(8)) == 563
    from raytracer_amd import capi
    lib = capi.load_library()
    for (h, b, w) in ((50, 16, 3), (563, 8, 8), (1080, 16, 8), (7, 8, 2)):
        for k in range(w):
            assert lib.rt_band_rows(h, b, w, k) == parallel.band_rows_of(h, b, w, k).size


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
        assert int(r["allgather_bytes"]) > 0
    maps = ora.Maps(single["diffuse"], single["caustic"])
    _, post, _, _ = ora.render(eng.sc, eng.p, maps, 0, 0, W, H)
    got = r0["frame"]
    assert got.shape == post.shape
    assert ((got == post) | (np.isnan(got) & np.isnan(post))).all()
    assert r1["frame"].size == 0
    # the batched gather: every frame of the batch is that same frame (the oracle engine renders one seed)
    assert r0["batch"].shape == (3,) + post.shape and r1["batch"].size == 0
    assert r0["over"].shape == (3,) + post.shape and r1["over"].size == 0
    for f in list(r0["batch"]) + list(r0["over"]):
        assert ((f == post) | (np.isnan(f) & np.isnan(post))).all()
