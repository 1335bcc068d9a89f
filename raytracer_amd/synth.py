"""Synthetic workloads for bench.py and the full-size property tests (seeded, no files).

  * box_photons / box_queries : the photon-map workload -- N photons on the inner faces of the box
    [-4,4] x [-4,4] x [0,8] and one query per pixel of a camera inside that box.
  * many_primitive_scene      : a scene of ~n_prims primitives (spheres, cans, boxes, cups) inside a
    room of six planes with two spotlights.

The composite helpers below follow the reference's constructors (Box rayphoton.js:184-210, Ball
:272-276, Can :278-284, Tube :286-293, Spotlight :316-328, all shrinking by LITTLE_SPACE = 1e-3).
"""
from __future__ import annotations

import math

import numpy as np

from . import scene as S

LITTLE_SPACE = 1e-3


# ------------------------------------------------------------------ photon-map workload
def box_photons(n: int, seed: int = 7) -> np.ndarray:
    """(n, 11) float64 photon records: positions uniform on the six inner faces (fp32-representable),
    directions cosine-weighted about the inward normal, power*colour uniform in [0,1]^3."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-4.0, -4.0, 0.0]), np.array([4.0, 4.0, 8.0])
    face = rng.integers(0, 6, n)
    pos = rng.uniform(lo, hi, (n, 3)).astype(np.float32).astype(np.float64)
    axis, side = face // 2, face % 2
    pos[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    nrm = np.zeros((n, 3))
    nrm[np.arange(n), axis] = np.where(side == 0, 1.0, -1.0)
    u1, u2 = rng.random(n), rng.random(n)
    ct, st, phi = np.sqrt(u1), np.sqrt(1 - u1), 2 * np.pi * u2
    t1 = np.zeros((n, 3))
    t1[np.arange(n), (axis + 1) % 3] = 1.0
    t2 = np.cross(nrm, t1)
    d = ct[:, None] * nrm + (st * np.cos(phi))[:, None] * t1 + (st * np.sin(phi))[:, None] * t2
    rec = np.zeros((n, 11))
    rec[:, 0:3] = pos
    rec[:, 3:6] = d
    rec[:, 6] = 1.0 / 255.0
    rec[:, 7:10] = rng.uniform(0, 255, (n, 3))
    rec[:, 10] = 1.0
    return rec


def box_queries(width: int, height: int, tile: int = 8) -> np.ndarray:
    """(width*height, 9) float32 hit records pos3 normal3 albedo3: where the primary ray of each pixel
    of a camera at (0.5, -3.5, 3) looking along +y leaves the box.  Pixels are emitted tile by tile
    (tile x tile blocks, row-major inside a block) so that one 64-lane wave covers an 8x8 patch."""
    lo, hi = np.array([-4.0, -4.0, 0.0]), np.array([4.0, 4.0, 8.0])
    org = np.array([0.5, -3.5, 3.0])
    fov_w = math.tan(math.radians(35))
    fov_h = fov_w * height / width
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    th, tw = (height + tile - 1) // tile, (width + tile - 1) // tile
    order = np.lexsort((xs.ravel() % tile, ys.ravel() % tile, xs.ravel() // tile, ys.ravel() // tile))
    assert th * tw * tile * tile >= order.size
    px, py = xs.ravel()[order], ys.ravel()[order]
    u = ((px + 0.5) * 2 / width - 1) * fov_w
    v = -((py + 0.5) * 2 / height - 1) * fov_h
    d = np.stack([u, np.ones_like(u), v], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_exit = np.where(d > 0, (hi - org) / d, np.where(d < 0, (lo - org) / d, np.inf))
    axis = np.argmin(t_exit, axis=1)
    t = t_exit[np.arange(d.shape[0]), axis]
    pos = org + t[:, None] * d
    nrm = np.zeros_like(pos)
    nrm[np.arange(d.shape[0]), axis] = -np.sign(d[np.arange(d.shape[0]), axis])
    pos[np.arange(d.shape[0]), axis] = np.where(nrm[np.arange(d.shape[0]), axis] > 0, lo[axis], hi[axis])
    hits = np.concatenate([pos, nrm, np.full_like(pos, 200.0)], axis=1).astype(np.float32)
    return np.ascontiguousarray(hits)


# ------------------------------------------------------------------ composite builders (Python mirror)
def _norm(v):
    v = np.asarray(v, float)
    s = float(v @ v)
    return np.array([0.0, 0.0, 1.0]) if s < 1e-6 else v / math.sqrt(s)


class SceneBuilder:
    def __init__(self):
        self.meta, self.geom, self.colour = [], [], []
        self.lights = []          # (shape index, wattage, area, colour)

    def _push(self, type_id, material, flags, g, colour):
        gg = np.zeros(12)
        gg[:len(g)] = g
        self.meta.append((type_id, material, flags, 0))
        self.geom.append(gg)
        self.colour.append(tuple(colour) + (1.0,) * (4 - len(colour)))
        return len(self.meta) - 1

    def plane(self, origin, normal, colour, material):
        return self._push(S.PLANE, material, 0, [*origin, *_norm(normal)], colour)

    def sphere(self, centre, radius, colour, material):
        return self._push(S.SPHERE, material, 0, [*centre, 0, 0, 0, radius], colour)

    def cylinder(self, centre, axis, radius, half_height, convex, colour, material):
        return self._push(S.CYLINDER, material, (S.FLAG_CONVEX if convex else 0) | S.FLAG_HALF_HEIGHT,
                          [*centre, *_norm(axis), radius, half_height], colour)

    def disc(self, centre, radius, normal, colour, material, flags=0):
        return self._push(S.DISC, material, flags, [*centre, *_norm(normal), radius], colour)

    def annulus(self, centre, outer, inner, normal, colour, material):
        return self._push(S.ANNULUS, material, 0, [*centre, *_norm(normal), outer, inner], colour)

    def hemisphere(self, centre, radius, normal, trunc_min, trunc_max, convex, colour, material):
        flags = (S.FLAG_CONVEX if convex else 0) | (S.FLAG_TRUNC_MAX if trunc_max is not None else 0)
        return self._push(S.HEMISPHERE, material, flags, [*centre, *_norm(normal), radius, trunc_min, trunc_max or 0.0], colour)

    def square(self, a, ab, ac, colour, material):
        n = _norm(np.cross(ab, ac))
        return self._push(S.SQUARE, material, 0, [*a, *ab, *ac, *n], colour)

    # composites
    def ball(self, centre, radius, colour, material):
        return self.sphere(centre, radius - LITTLE_SPACE, colour, material)

    def box(self, a, ab, ac, ad, colour, material):
        a, ab, ac, ad = (np.asarray(x, float) for x in (a, ab, ac, ad))
        adj_a = a + LITTLE_SPACE * _norm(ab) + LITTLE_SPACE * _norm(ac) + LITTLE_SPACE * _norm(ad)
        eb, ec, ed = (e - 2 * LITTLE_SPACE * _norm(e) for e in (ab, ac, ad))
        opp = adj_a + eb + ec + ed
        for (v, e1, e2) in ((adj_a, ec, eb), (adj_a, ed, ec), (adj_a, eb, ed), (opp, -eb, -ec), (opp, -ec, -ed), (opp, -ed, -eb)):
            self.square(v, e1, e2, colour, material)

    def can(self, centre, axis, radius, half_height, colour, material):
        centre, ax = np.asarray(centre, float), _norm(axis)
        self.cylinder(centre, axis, radius - LITTLE_SPACE, half_height - LITTLE_SPACE, True, colour, material)
        self.disc(centre + (half_height - LITTLE_SPACE) * ax, radius - LITTLE_SPACE, ax, colour, material)
        self.disc(centre - (half_height - LITTLE_SPACE) * ax, radius - LITTLE_SPACE, -ax, colour, material)

    def cup(self, centre, axis, outer, inner, half_height, colour, material):
        """a Tube (rayphoton.js:286-293) whose bottom annulus is replaced by a full disc"""
        centre, ax = np.asarray(centre, float), _norm(axis)
        self.cylinder(centre, axis, outer - LITTLE_SPACE, half_height - LITTLE_SPACE, True, colour, material)
        self.cylinder(centre, axis, inner + LITTLE_SPACE, half_height - LITTLE_SPACE, False, colour, material)
        self.annulus(centre + (half_height - LITTLE_SPACE) * ax, outer - LITTLE_SPACE, inner + LITTLE_SPACE, ax, colour, material)
        self.disc(centre - (half_height - LITTLE_SPACE) * ax, outer - LITTLE_SPACE, -ax, colour, material)

    def spotlight(self, centre, radius, direction, wattage, colour=(255, 255, 255, 1)):
        centre, dr = np.asarray(centre, float), _norm(direction)
        copper, umber = S.MAT_COPPER, (112, 68, 17)
        self.hemisphere(centre, 1.3 * radius - LITTLE_SPACE, -dr, 0, None, True, umber, copper)
        self.hemisphere(centre, 1.125 * radius + LITTLE_SPACE, -dr, 0, None, False, umber, copper)
        self.annulus(centre, 1.3 * radius - LITTLE_SPACE, 1.125 * radius + LITTLE_SPACE, dr, umber, copper)
        s = self.disc(centre - 0.5 * radius * dr, radius, dr, (255, 32, 144), S.MAT_NONE, flags=S.FLAG_LIGHT)
        self.lights.append((s, float(wattage), math.pi * radius ** 2, colour))

    def finish(self, cam_origin, gaze_theta, gaze_phi, width, height, fov=60.0) -> S.FlatScene:
        nl = len(self.lights)
        light_data = np.zeros((nl, 8))
        tot = 0.0
        for i, (s, w, a, c) in enumerate(self.lights):                       # initLights, rayphoton.js:1313-1324
            light_data[i, 0], light_data[i, 1] = w, a
            light_data[i, 3:7] = c
            tot += w * a
        for i, (s, w, a, c) in enumerate(self.lights):
            light_data[i, 2] = (w * a) / tot
        th, ph = math.radians(gaze_theta), math.radians(gaze_phi)            # Camera, rayphoton.js:697-706
        gaze = _norm([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)])
        w_ = -gaze
        u_ = _norm(np.cross([0, 0, 1.0], w_))
        v_ = np.cross(w_, u_)
        cam = np.zeros(16)
        cam[0:3], cam[3:6], cam[6:9], cam[9:12] = cam_origin, u_, v_, w_
        cam[12] = math.tan(math.radians(fov / 2))
        cam[13] = cam[12] * height / width
        sc = S.FlatScene(np.array(self.meta, np.uint8), np.array(self.geom, np.float64), np.array(self.colour, np.float64),
                         np.array([l[0] for l in self.lights], np.uint32), light_data, cam, width, height)
        sc.validate()
        return sc


def many_primitive_scene(n_prims: int = 10000, seed: int = 42, width: int = 3840, height: int = 2160) -> S.FlatScene:
    """~n_prims primitives: 40 % spheres r in [.05,.3], 20 % cans (cylinder + 2 discs), 30 % box faces
    (6 squares per box), 10 % cups (2 cylinders + annulus + disc); centres uniform in [-20,20]^2 x [0,10];
    materials 60 % plaster / 15 % copper / 10 % mirror / 10 % glass / 5 % water; six room planes; two spotlights."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    grey = (128, 128, 128)
    for o, n in (((0, 0, 0), (0, 0, 1)), ((0, 0, 14), (0, 0, -1)), ((-24, 0, 0), (1, 0, 0)), ((24, 0, 0), (-1, 0, 0)),
                 ((0, -24, 0), (0, 1, 0)), ((0, 24, 0), (0, -1, 0))):
        b.plane(o, n, grey, S.MAT_PLASTER)
    mats = [S.MAT_PLASTER, S.MAT_COPPER, S.MAT_MIRROR, S.MAT_GLASS, S.MAT_WATER]
    probs = [0.60, 0.15, 0.10, 0.10, 0.05]

    def pick():
        return int(rng.choice(mats, p=probs)), tuple(rng.integers(32, 256, 3).tolist())

    def centre():
        return np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(0.5, 10)])

    for _ in range(int(0.40 * n_prims)):
        m, c = pick()
        b.ball(centre(), rng.uniform(0.05, 0.3), c, m)
    for _ in range(int(0.20 * n_prims) // 3):
        m, c = pick()
        b.can(centre(), rng.normal(size=3), rng.uniform(0.1, 0.4), rng.uniform(0.1, 0.5), c, m)
    for _ in range(int(0.30 * n_prims) // 6):
        m, c = pick()
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 2] = -q[:, 2]
        l = rng.uniform(0.2, 0.8, 3)
        a = centre() - 0.5 * (q @ l)
        b.box(a, l[0] * q[:, 0], l[1] * q[:, 1], l[2] * q[:, 2], c, m)
    for _ in range(int(0.10 * n_prims) // 4):
        m, c = pick()
        ro = rng.uniform(0.15, 0.4)
        b.cup(centre(), rng.normal(size=3), ro, ro * rng.uniform(0.6, 0.9), rng.uniform(0.1, 0.4), c, m)
    b.spotlight((-6, 0, 13), 1.0, (0.2, 0.1, -1), 400)
    b.spotlight((8, -5, 13), 1.0, (-0.1, 0.2, -1), 400)
    return b.finish((-22, -22, 9), 45, -12, width, height)


def first_hit_queries(scene, params, width, height, tile=8):
    """Radiance-gather queries at the first-hit points of a frame's primary rays (pixel centres), ordered by
    tile x tile image tiles like box_queries: (n, 9) float32 -- position, normal, albedo.  The hits come from the
    library's closest-hit scan; the normal is the direction back to the camera and the albedo white (only the positions
    shape the cost of a gather).  Rays that miss or end on a light are left out."""
    from . import capi
    cam = np.asarray(scene.flat.camera, np.float64)
    o, cu, cv, cw = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    ys, xs = np.mgrid[0:height, 0:width]
    if height % tile == 0 and width % tile == 0:                      # tile order
        order = np.arange(width * height).reshape(height // tile, tile, width // tile, tile).transpose(0, 2, 1, 3).reshape(-1)
        xs, ys = xs.reshape(-1)[order], ys.reshape(-1)[order]
    else:
        xs, ys = xs.reshape(-1), ys.reshape(-1)
    u = ((xs + 0.5) * 2.0 / width - 1.0) * cam[12]
    v = -((ys + 0.5) * 2.0 / height - 1.0) * cam[13]
    d = u[:, None] * cu + v[:, None] * cv - cw
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.broadcast_to(o, d.shape).copy()
    t, idx = scene.intersect(params, capi.SCAN_CLOSEST, org, d)
    light = np.zeros(scene.flat.n_shapes, bool)
    light[np.asarray(scene.flat.light_shape, np.int64)] = True
    keep = (idx >= 0) & np.isfinite(t)
    keep[keep] &= ~light[idx[keep]]
    pos = org[keep] + t[keep, None] * d[keep]
    return np.concatenate([pos, -d[keep], np.full((pos.shape[0], 3), 255.0)], axis=1).astype(np.float32)
