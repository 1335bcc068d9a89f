"""Multi-GPU sharding of the hot path: one process and one rt_ctx per GPU, torch.distributed for the two
exchange steps (backend "nccl" is RCCL over xGMI on ROCm; "gloo" with CPU tensors in the tests).

The path shards by construction: every emission attempt and every pixel has its own RNG stream keyed by its
index, so ranks take disjoint index ranges and no collective sits inside a kernel.

  * emission (emitManyPhotons, rayphoton.js:1354-1383): a batch of attempts [A, A + B) is cut into `world`
    contiguous sub-ranges in rank order.  Rank k emits its sub-range into a SHARD in its own device memory
    (rt_emit_shard: stored photons in attempt order, W + 1 words each, tag = attempt - A | caustic bit).  The
    shard lengths are all-gathered (one int per rank), then the shards themselves, cut to the longest, in ONE
    all-gather straight from and into device memory -- with the packed format that is the 36-byte photon plus
    4 bytes of tag.  Every rank then applies the reference's list caps to the identical merged batch on its GPU
    (rt_lists_apply): identical photon lists and therefore identical, deterministically built kd-trees on every
    rank, with no tree broadcast.
  * render: row bands of the frame are interleaved over the ranks (cost per pixel is uneven: glass against bare
    wall); each rank traces its bands in one call (rt_params.band_*) into device memory and the bands are
    gathered to rank 0.

An ENGINE hides who computes: DeviceEngine drives librt355 through raytracer_amd.capi; the CPU tests use the
oracle behind the same five methods with CPU tensors.
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


# ------------------------------------------------------------------ the caps, restated for engines without a GPU
@dataclass
class EmitState:
    """list caps of emitManyPhotons, applied batch by batch in attempt order (numpy; the GPU path applies the same
    rules on the device, csrc/rt_emit.h)"""
    nd_cap: int
    nc_cap: int
    diffuse: list = field(default_factory=list)
    caustic: list = field(default_factory=list)
    diffuse_attempt: list = field(default_factory=list)
    caustic_attempt: list = field(default_factory=list)
    nd: int = 0
    nc: int = 0

    def apply(self, phase, a0, rec, attempt, kind):
        """Consume the stored photons of one batch (attempt-ordered).  Returns (phase_ended, last_attempt)."""
        rec, attempt, kind = np.asarray(rec), np.asarray(attempt), np.asarray(kind)
        d, c = kind == 1, kind == 2
        da, ca = attempt[d], attempt[c]
        ended, last = False, None
        take_d = take_c = 0
        if phase == 1:
            take_d = min(int(d.sum()), self.nd_cap - self.nd)
            if self.nd + take_d == self.nd_cap and take_d > 0:
                ended, last = True, int(da[take_d - 1])     # the attempt that stores the N_D-th diffuse photon ends phase 1
            upto = int(np.searchsorted(ca, last, side="right")) if ended else int(c.sum())
            take_c = min(upto, self.nc_cap - self.nc)
        else:
            take_c = min(int(c.sum()), self.nc_cap - self.nc)
            if self.nc + take_c == self.nc_cap and take_c > 0:
                ended, last = True, int(ca[take_c - 1])
        if take_d:
            self.diffuse.append(rec[d][:take_d])
            self.diffuse_attempt.append(np.stack([np.full(take_d, phase), da[:take_d]], axis=1))
        if take_c:
            self.caustic.append(rec[c][:take_c])
            self.caustic_attempt.append(np.stack([np.full(take_c, phase), ca[:take_c]], axis=1))
        self.nd += take_d
        self.nc += take_c
        return ended, last

    def result(self):
        cat = lambda xs, w: np.concatenate(xs) if xs else np.zeros((0, w))
        return dict(diffuse=cat(self.diffuse, 11), caustic=cat(self.caustic, 11),
                    diffuse_attempt=cat(self.diffuse_attempt, 2).astype(np.int64),
                    caustic_attempt=cat(self.caustic_attempt, 2).astype(np.int64))


def _world():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_gather_rows(local, world):
    """local: (m, w) tensor, the same m on every rank -> (world, m, w) in rank order, one collective"""
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out.view(-1, *local.shape[1:]), local.contiguous())
    except (RuntimeError, NotImplementedError, AttributeError):
        dist.all_gather([out[k] for k in range(world)], local.contiguous())
    return out


def emit_sharded(engine, batch=1 << 22, world_override=None, timings=None):
    """emitManyPhotons over all ranks; every rank ends with the same lists inside its engine.

    batch: attempts per round over ALL ranks.  world_override=R makes this one process play R ranks in turn (a
    single-GPU rehearsal of the protocol: same shards, same merge, no collective).
    timings (optional dict) receives seconds per phase of the protocol and the bytes all-gathered."""
    import time
    rank, world = _world()
    sim = world_override if (world_override and world == 1) else None
    nranks = sim or world
    nd_cap, nc_cap = engine.caps()
    attempts = {1: 0, 2: 0}
    capped = False
    T = dict(emit_s=0.0, exchange_s=0.0, apply_s=0.0, allgather_bytes=0, rounds=0)
    for phase in (1, 2):
        a = 0
        limit = 100 * (nd_cap if phase == 1 else nc_cap)            # rayphoton.js:1365, :1377
        got0 = engine.counts()[phase - 1]                             # what the phase's list held when the phase began
        while not engine.phase_done(phase):
            # Every attempt of a round is traced, also those behind the attempt that fills the list, so a round should
            # end just behind it: the first round (at most 2^18 attempts per rank) measures the phase's yield, the next is
            # sized for what is still missing plus 1 %; while nothing has been stored the rounds run four times the need.
            # All within [65536 per rank, batch] and the reference's attempt limit.
            nd_now, nc_now = engine.counts()
            need = (nd_cap - nd_now) if phase == 1 else (nc_cap - nc_now)
            got = (nd_now if phase == 1 else nc_now) - got0
            b = max(4 * need, 65536 * nranks)
            if a == 0:
                b = min(b, (1 << 18) * nranks)
            elif got > 0:
                b = max(int(need * a / got * 1.01) + 4096, 65536 * nranks)
            b = int(min(batch, b, limit + 1 - a))
            bounds = [a + (b * k) // nranks for k in range(nranks + 1)]   # contiguous sub-ranges in rank order = attempt order
            t0 = time.perf_counter()
            mine = range(nranks) if sim else [rank]
            shards, counts = [], []
            for k in mine:
                sh, n = engine.emit_shard(phase, bounds[k], bounds[k + 1], a)
                shards.append(sh)
                counts.append(n)
            engine.sync()
            t1 = time.perf_counter()
            if sim:
                cnt = torch.cat(counts)
                m = max(int(cnt.max().item()), 1)
                merged = torch.stack([_first_rows(s, m) for s in shards])
            elif world > 1:
                cnt = _all_gather_rows(counts[0].view(1, 1), world).view(world)
                m = max(int(cnt.max().item()), 1)                     # (host sync: the longest shard sizes the exchange)
                merged = _all_gather_rows(_first_rows(shards[0], m), world)
                T["allgather_bytes"] += merged.numel() * merged.element_size()
            else:
                cnt = counts[0]
                m = max(int(cnt.item()), 1)
                merged = _first_rows(shards[0], m).unsqueeze(0)
            engine.sync()
            t2 = time.perf_counter()
            ended, last = engine.apply(phase, a, merged, cnt.to(torch.int32), nranks, m)
            t3 = time.perf_counter()
            T["emit_s"] += t1 - t0
            T["exchange_s"] += t2 - t1
            T["apply_s"] += t3 - t2
            T["rounds"] += 1
            a = (last if ended else a + b - 1) + 1
            attempts[phase] = a
            if a > limit:
                capped = True
                break
        attempts[phase] = a
        if capped:
            break
    if timings is not None:
        timings.update(T)
    nd, nc = engine.counts()
    return dict(attempts1=attempts[1], attempts2=attempts[2], capped=capped, n_diffuse=nd, n_caustic=nc)


def _first_rows(shard, m):
    """the first m records of a shard tensor, padded with zero rows if the shard's capacity is smaller"""
    if shard.shape[0] >= m:
        return shard[:m]
    pad = torch.zeros((m - shard.shape[0], shard.shape[1]), dtype=shard.dtype, device=shard.device)
    return torch.cat([shard, pad])


def band_rows_of(height, band, world, rank):
    """rows of the frame in rank's interleaved bands (band i -> rank i % world), top to bottom"""
    y = np.arange(height)
    return y[(y // band) % world == rank]


def render_sharded(engine, width, height, band=16, out=None):
    """Every rank traces its row bands; rank 0 returns the assembled (height, width, C) frame tensor, others None.
    The per-rank parts travel as device tensors (padded to the longest part)."""
    rank, world = _world()
    part = engine.render_bands(width, height, band, world, rank)      # (rows_k, width, C) tensor on the engine's device
    if world == 1:
        return part
    rows_max = max(len(band_rows_of(height, band, world, k)) for k in range(world))
    if part.shape[0] < rows_max:
        part = torch.cat([part, torch.zeros((rows_max - part.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)])
    parts = [torch.empty_like(part) for _ in range(world)] if rank == 0 else None
    dist.gather(part.contiguous(), parts, dst=0)
    if rank != 0:
        if part.is_cuda:
            torch.cuda.current_stream().synchronize()        # (the engine's buffer is free again once the bands have left)
        return None
    frame = out if out is not None else torch.empty((height, width) + tuple(part.shape[2:]), dtype=part.dtype, device=part.device)
    for k in range(world):
        key = (height, band, world, k, str(part.device))
        if key not in _ROWS:
            _ROWS[key] = torch.from_numpy(band_rows_of(height, band, world, k)).to(part.device)
        frame.index_copy_(0, _ROWS[key], parts[k][:_ROWS[key].numel()])
    if part.is_cuda:
        torch.cuda.current_stream().synchronize()
    return frame


_ROWS = {}


def trace_bands_batch(engine, width, height, n_frames, band=16):
    """this rank's row bands of n_frames frames, traced back to back and finished: (n_frames, rows, width, C)"""
    rank, world = _world()
    if hasattr(engine, "render_bands_batch"):
        return engine.render_bands_batch(width, height, band, world, rank, n_frames)
    return torch.stack([engine.render_bands(width, height, band, world, rank) for _ in range(n_frames)])


def gather_bands_begin(parts, height, band=16):
    """starts the ONE gather that brings every rank's (n_frames, rows_k, width, C) bands to rank 0 and returns a handle
    for gather_bands_end; the collective runs beside whatever the caller queues next (the next batch's tracing).  The
    caller leaves `parts` alone until gather_bands_end has returned."""
    rank, world = _world()
    if world == 1:
        return {"parts": parts, "work": None, "got": None}
    rows_max = max(len(band_rows_of(height, band, world, k)) for k in range(world))
    if parts.shape[1] < rows_max:
        pad = torch.zeros((parts.shape[0], rows_max - parts.shape[1]) + tuple(parts.shape[2:]), dtype=parts.dtype, device=parts.device)
        parts = torch.cat([parts, pad], dim=1)
    parts = parts.contiguous()
    got = [torch.empty_like(parts) for _ in range(world)] if rank == 0 else None
    work = dist.gather(parts, got, dst=0, async_op=True)
    return {"parts": parts, "work": work, "got": got}


def gather_bands_end(handle, width, height, band=16, out=None):
    """waits for a gather begun with gather_bands_begin; rank 0 returns the assembled (n_frames, height, width, C) frames
    (in `out` if given), the others None"""
    rank, world = _world()
    parts = handle["parts"]
    if world == 1:
        return parts
    handle["work"].wait()
    if rank != 0:
        # (a device collective's wait() orders torch's stream behind it, not the host and not the library's streams, which
        # will write these bands' buffer again: the host waits here until this rank's bands have left)
        if parts.is_cuda:
            torch.cuda.current_stream().synchronize()
        return None
    got, n_frames = handle["got"], parts.shape[0]
    frames = out if out is not None else torch.empty((n_frames, height, width) + tuple(parts.shape[3:]), dtype=parts.dtype, device=parts.device)
    for k in range(world):
        key = (height, band, world, k, str(parts.device))
        if key not in _ROWS:
            _ROWS[key] = torch.from_numpy(band_rows_of(height, band, world, k)).to(parts.device)
        frames.index_copy_(1, _ROWS[key], got[k][:, :_ROWS[key].numel()])
    if parts.is_cuda:
        torch.cuda.current_stream().synchronize()
    return frames


def render_sharded_batch(engine, width, height, n_frames, band=16, out=None):
    """n_frames frames at once: every rank traces its row bands of all of them back to back, ONE gather brings the
    (n_frames, rows_k, width, C) parts to rank 0, which returns the assembled (n_frames, height, width, C) tensor
    (others None).  One collective and one host wait per batch instead of per frame: a frame's bands take tens of
    microseconds to trace, less than a collective's latency.  (A caller with several batches overlaps a batch's gather
    with the next batch's tracing through the three functions above.)"""
    parts = trace_bands_batch(engine, width, height, n_frames, band)
    return gather_bands_end(gather_bands_begin(parts, height, band), width, height, band, out)


class DeviceEngine:
    """librt355 behind the engine interface: shards, lists and frames stay in this rank's GPU memory"""

    def __init__(self, scene, params, fmt=None, device="cuda", with_lists=True):
        from . import capi
        self.capi, self.scene, self.params, self.device = capi, scene, params, device
        self.fmt = capi.SHARD_PACKED if fmt is None else fmt
        self.lists = scene.ctx.lists(params, self.fmt) if with_lists else None
        self.words = self.lists.words if with_lists else 0
        self.out = None                                               # optional caller buffer for the next render_bands
        self.maps = [None, None]
        self._nd = self._nc = 0

    def caps(self):
        return self.params.num_photons_diffuse, self.params.num_photons_caustic

    def counts(self):
        return self._nd, self._nc

    def phase_done(self, phase):
        return self._nd >= self.params.num_photons_diffuse if phase == 1 else self._nc >= self.params.num_photons_caustic

    def sync(self):
        """host wait for the library's streams and for torch's (the two sides hand buffers to each other)"""
        self.scene.ctx.synchronize()
        if str(self.device).startswith("cuda"):
            torch.cuda.synchronize()

    def emit_shard(self, phase, a0, a1, tag_base):
        cap = max(a1 - a0, 1)                                         # every attempt stores at most one photon
        buf = torch.empty((cap, self.words + 1), dtype=torch.int32, device=self.device)     # (torch's allocator recycles these)
        n = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.scene.emit_shard(self.params, phase, a0, a1, self.fmt, tag_base, buf, n, cap=cap)
        return buf, n

    def apply(self, phase, tag_base, merged, counts, n_shards, stride):
        info = self.lists.apply(phase, tag_base, merged.contiguous(), counts.contiguous(), n_shards, stride)
        self._nd, self._nc = info["n_diffuse"], info["n_caustic"]
        return bool(info["phase_ended"]), int(info["last_attempt"])

    def build_maps(self):
        """the replicated, deterministic kd-tree builds (identical on every rank)"""
        self.maps = [self.lists.photon_map(w) if n else None for w, n in ((0, self._nd), (1, self._nc))]
        return self.maps

    def render_bands(self, width, height, band, world, rank):
        """this rank's bands of the frame, finished (the caller hands the tensor to a collective on torch's streams)"""
        capi = self.capi
        out = self.out
        q = capi.Params.from_buffer_copy(self.params)
        q.band_rows, q.band_count, q.band_index = (band, world, rank) if world > 1 else (0, 0, 0)
        rows = self.scene.ctx.band_rows(height, band, world, rank) if world > 1 else height
        if out is None:
            out = torch.empty((rows, width, 4), dtype=torch.uint8, device=self.device)
        if rows:
            self.scene.trace_tile(q, self.maps[0], self.maps[1], x0=0, y0=0, w=width, h=height, want=(), out_rgba8=out, stats=False)
            self.scene.ctx.synchronize()
        return out

    def render_bands_batch(self, width, height, band, world, rank, n_frames, in_flight=4):
        """this rank's bands of n_frames frames, (n_frames, rows, width, 4), queued back to back (up to in_flight of them
        overlap on the library's streams) and finished when this returns"""
        capi, ctx = self.capi, self.scene.ctx
        q = capi.Params.from_buffer_copy(self.params)
        q.band_rows, q.band_count, q.band_index = (band, world, rank) if world > 1 else (0, 0, 0)
        rows = ctx.band_rows(height, band, world, rank) if world > 1 else height
        out = self.out
        if out is None or tuple(out.shape) != (n_frames, rows, width, 4):
            out = torch.empty((n_frames, rows, width, 4), dtype=torch.uint8, device=self.device)
        if rows:
            ctx.set_frames_in_flight(max(1, min(in_flight, n_frames)))
            try:
                for j in range(n_frames):
                    self.scene.trace_tile(q, self.maps[0], self.maps[1], x0=0, y0=0, w=width, h=height, want=(), out_rgba8=out[j], stats=False)
                ctx.synchronize()
            finally:
                ctx.set_frames_in_flight(1)
        return out
