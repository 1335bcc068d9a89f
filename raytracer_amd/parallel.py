"""Multi-GPU sharding of the hot path (one process and one rt_ctx per GPU, torch.distributed for
the exchange: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by construction: every emission attempt and every pixel has its own RNG stream
keyed by its index, so ranks take disjoint index ranges and no collective sits inside a kernel.
Two exchange steps exist:

  * emission: each round, rank k runs attempts [A + k*B/R, A + (k+1)*B/R); the attempt-ordered
    records are all-gathered (padded to the longest shard) and every rank applies the reference's
    list caps (rayphoton.js:1346-1383) to the identical merged sequence -> identical photon lists
    on every rank, hence identical (deterministic) kd-trees with no tree broadcast;
  * render: row bands are interleaved over ranks (cost per pixel is uneven) and gathered to rank 0.

`engine` is anything with emit(phase, a0, a1) -> (rec[n,11] f64, attempt[n] i64, kind[n] u8) and
render(x0, y0, w, h) -> post[h,w,4] f64; on a GPU rank that is a thin adapter over capi.Scene, in the
CPU tests it is the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:                                       # pragma: no cover
    torch = None
    dist = None


@dataclass
class EmitState:
    """list caps of emitManyPhotons, applied batch by batch in attempt order"""
    nd_cap: int
    nc_cap: int
    diffuse: list = field(default_factory=list)
    caustic: list = field(default_factory=list)
    diffuse_attempt: list = field(default_factory=list)
    caustic_attempt: list = field(default_factory=list)
    nd: int = 0
    nc: int = 0
    attempts: dict = field(default_factory=lambda: {1: 0, 2: 0})
    capped: bool = False

    def phase_done(self, phase):
        return self.nd >= self.nd_cap if phase == 1 else self.nc >= self.nc_cap

    def apply(self, phase, a0, a1, rec, attempt, kind):
        """Consume the stored photons of attempts [a0, a1) (attempt-ordered).  Returns the next attempt
        index the sequential loop would run (== a1 unless a cap ended the phase inside this batch)."""
        rec, attempt, kind = np.asarray(rec), np.asarray(attempt), np.asarray(kind)
        d, c = kind == 1, kind == 2
        da, ca = attempt[d], attempt[c]
        last = a1 - 1
        take_d = take_c = 0
        if phase == 1:
            take_d = min(int(d.sum()), self.nd_cap - self.nd)
            if self.nd + take_d == self.nd_cap and take_d > 0:
                last = int(da[take_d - 1])              # the attempt that stores the N_D-th diffuse photon ends phase 1
            take_c = min(int(np.searchsorted(ca, last, side="right")), self.nc_cap - self.nc)
        else:
            take_c = min(int(c.sum()), self.nc_cap - self.nc)
            if self.nc + take_c == self.nc_cap and take_c > 0:
                last = int(ca[take_c - 1])
        if take_d:
            self.diffuse.append(rec[d][:take_d])
            self.diffuse_attempt.append(np.stack([np.full(take_d, phase), da[:take_d]], axis=1))
        if take_c:
            self.caustic.append(rec[c][:take_c])
            self.caustic_attempt.append(np.stack([np.full(take_c, phase), ca[:take_c]], axis=1))
        self.nd += take_d
        self.nc += take_c
        self.attempts[phase] = last + 1
        if self.attempts[phase] > 100 * (self.nd_cap if phase == 1 else self.nc_cap):   # rayphoton.js:1365, :1377
            self.capped = True
        return last + 1

    def result(self):
        cat = lambda xs, w: np.concatenate(xs) if xs else np.zeros((0, w))
        return dict(diffuse=cat(self.diffuse, 11), caustic=cat(self.caustic, 11),
                    diffuse_attempt=cat(self.diffuse_attempt, 2).astype(np.int64),
                    caustic_attempt=cat(self.caustic_attempt, 2).astype(np.int64),
                    attempts1=self.attempts[1], attempts2=self.attempts[2], capped=self.capped)


def _world():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_gather_ragged(arrays, device):
    """all-gather a tuple of per-rank arrays of equal leading length n_k (different per rank);
    returns per-array concatenation in rank order."""
    rank, world = _world()
    if world == 1:
        return arrays
    n = torch.tensor([arrays[0].shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    out = []
    for a in arrays:
        t = torch.zeros((m,) + a.shape[1:], dtype=torch.from_numpy(a[:0]).dtype, device=device)
        if a.shape[0]:
            t[:a.shape[0]] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)                       # padded shards, one collective per field
        out.append(np.concatenate([p[:counts[k]].cpu().numpy() for k, p in enumerate(parts)]))
    return tuple(out)


def emit_sharded(engine, nd_cap, nc_cap, batch=1 << 16, device="cpu"):
    """emitManyPhotons over all ranks.  Every rank returns the same lists."""
    rank, world = _world()
    st = EmitState(nd_cap, nc_cap)
    for phase in (1, 2):
        a = 0
        cap = 100 * (nd_cap if phase == 1 else nc_cap)
        while not st.phase_done(phase) and not st.capped:
            b = min(batch, cap + 1 - a)
            # contiguous sub-ranges in rank order, so rank-major concatenation is attempt order
            lo = a + (b * rank) // world
            hi = a + (b * (rank + 1)) // world
            rec, att, kind = engine.emit(phase, lo, hi)
            rec, att, kind = _all_gather_ragged((np.asarray(rec, np.float64).reshape(-1, 11), np.asarray(att, np.int64),
                                                 np.asarray(kind, np.uint8)), device)
            a = st.apply(phase, a, a + b, rec, att, kind)
        if st.capped:
            break
    return st.result()


def band_plan(height, world, band=16):
    """row bands interleaved over ranks: band i -> rank i % world"""
    return [(y0, min(band, height - y0), (y0 // band) % world) for y0 in range(0, height, band)]


def render_sharded(engine, width, height, band=16, device="cpu"):
    """Every rank renders its bands; rank 0 returns the assembled post-floor RGBA frame, others None."""
    rank, world = _world()
    plan = band_plan(height, world, band)
    mine = {y0: engine.render(0, y0, width, h) for (y0, h, r) in plan if r == rank}
    if world == 1:
        frame = np.zeros((height, width, 4))
        for y0, img in mine.items():
            frame[y0:y0 + img.shape[0]] = img
        return frame
    rows = max(1, sum(h for (_, h, r) in plan if r == 0))       # rank 0 owns the most rows
    buf = torch.zeros((rows, width, 4), dtype=torch.float64, device=device)
    o = 0
    for (y0, h, r) in plan:
        if r == rank:
            buf[o:o + h] = torch.from_numpy(mine[y0]).to(device)
            o += h
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)
    if rank != 0:
        return None
    frame = np.zeros((height, width, 4))
    offs = [0] * world
    for (y0, h, r) in plan:
        frame[y0:y0 + h] = parts[r][offs[r]:offs[r] + h].cpu().numpy()
        offs[r] += h
    return frame


class GpuEngine:
    """adapter: capi.Scene (+ photon maps) -> the engine interface above"""

    def __init__(self, scene, params, diffuse=None, caustic=None):
        self.scene, self.params, self.diffuse, self.caustic = scene, params, diffuse, caustic

    def emit(self, phase, a0, a1):
        rec, att, kind, _ = self.scene.emit_photons(self.params, phase, a0, a1)
        return rec, att, kind

    def render(self, x0, y0, w, h):
        return self.scene.trace_tile(self.params, self.diffuse, self.caustic, x0=x0, y0=y0, w=w, h=h, want=("post",))["post"]
