"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/librt_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OraScene(C.Structure):
    _fields_ = [("n_shapes", C.c_uint32), ("shape_meta", C.c_void_p), ("shape_geom", C.c_void_p),
                ("shape_colour", C.c_void_p), ("n_lights", C.c_uint32), ("light_shape", C.c_void_p),
                ("light_data", C.c_void_p), ("camera", C.c_double * 16), ("width", C.c_uint32),
                ("height", C.c_uint32), ("materials", (C.c_double * 8) * 4), ("shape_surface", C.c_void_p)]


class OraParams(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("max_trace_dist", C.c_double), ("max_depth", C.c_int32),
                ("sub_sample", C.c_int32), ("num_photons_diffuse", C.c_uint32),
                ("num_photons_caustic", C.c_uint32), ("diffuse_radius", C.c_double),
                ("caustic_radius", C.c_double), ("standard_lamp_area", C.c_double), ("seed", C.c_uint64),
                ("use_direct_light", C.c_int32), ("use_photon_maps", C.c_int32),
                ("super_sample_block", C.c_int32), ("pad_", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.ora_intersect.restype = C.c_double
        _LIB.ora_uniform.restype = C.c_double
        _LIB.ora_find_median.restype = C.c_uint32
        _LIB.ora_kd_radius.restype = C.c_uint32
        _LIB.ora_maps_create.restype = C.c_void_p
        _LIB.ora_maps_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        _LIB.ora_maps_destroy.argtypes = [C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Scene:
    """Keeps the numpy arrays alive behind an ora_scene struct."""

    def __init__(self, flat):
        self.flat = flat
        self._keep = [_c(flat.shape_meta, np.uint8), _c(flat.shape_geom, np.float64),
                      _c(flat.shape_colour, np.float64), _c(flat.light_shape, np.uint32),
                      _c(flat.light_data, np.float64)]
        s = OraScene()
        s.n_shapes = flat.n_shapes
        s.shape_meta, s.shape_geom, s.shape_colour = (_p(a) for a in self._keep[:3])
        s.n_lights = flat.n_lights
        s.light_shape, s.light_data = _p(self._keep[3]), _p(self._keep[4])
        for i in range(16):
            s.camera[i] = float(flat.camera[i])
        s.width, s.height = flat.width, flat.height
        for r in range(4):
            for k in range(8):
                s.materials[r][k] = float(flat.materials[r, k])
        if flat.shape_surface is not None:
            self._keep.append(_c(flat.shape_surface, np.float64))
            s.shape_surface = _p(self._keep[-1])
        self.c = s


def params(**kw) -> OraParams:
    p = OraParams()
    lib().ora_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def uniform(seed, domain, a, b, draw):
    return lib().ora_uniform(C.c_uint64(seed), C.c_uint32(domain), C.c_uint32(a), C.c_uint32(b), C.c_uint32(draw))


def intersect_batch(o, d, meta, geom, eps=1e-6, normalize_dir=False):
    o, d, meta, geom = _c(o, np.float64), _c(d, np.float64), _c(meta, np.uint8), _c(geom, np.float64)
    n = o.shape[0]
    t = np.empty(n, np.float64)
    lib().ora_intersect_batch(C.c_int(n), _p(o), _p(d), _p(meta), _p(geom), C.c_double(eps),
                              C.c_int(int(normalize_dir)), _p(t))
    return t


def scan_batch(scene: Scene, mode, o, d, source_dist=None, eps=1e-6):
    o, d = _c(o, np.float64), _c(d, np.float64)
    n = o.shape[0]
    t = np.empty(n, np.float64)
    idx = np.empty(n, np.int32)
    sd = _c(source_dist, np.float64) if source_dist is not None else np.zeros(n, np.float64)
    lib().ora_scan_batch(C.byref(scene.c), C.c_double(eps), C.c_int(mode), C.c_int(n), _p(o), _p(d), _p(sd), _p(t), _p(idx))
    return t, idx


def create_photon(scene: Scene, seed, domain, a, b):
    out = np.empty(8, np.float64)
    l = lib().ora_create_photon_keyed(C.byref(scene.c), C.c_uint64(seed), C.c_uint32(domain), C.c_uint32(a), C.c_uint32(b), _p(out))
    return l, out


def vec_perturb(seed, domain, a, b, first_draw, v):
    v = _c(v, np.float64)
    out = np.empty(3, np.float64)
    lib().ora_vec_perturb_keyed(C.c_uint64(seed), C.c_uint32(domain), C.c_uint32(a), C.c_uint32(b), C.c_uint32(first_draw), _p(v), _p(out))
    return out


def disc_sample(seed, domain, a, b, first_draw, c, radius, n):
    c, n = _c(c, np.float64), _c(n, np.float64)
    out = np.empty(3, np.float64)
    lib().ora_disc_sample_keyed(C.c_uint64(seed), C.c_uint32(domain), C.c_uint32(a), C.c_uint32(b), C.c_uint32(first_draw),
                                _p(c), C.c_double(radius), _p(n), _p(out))
    return out


def direct_light(scene: Scene, p: OraParams, domain, a, b, first_draw, point, normal, local):
    """directLight at one point on a keyed stream: (rgb, stream position afterwards)"""
    point, normal, local = _c(point, np.float64), _c(normal, np.float64), _c(local, np.float64)
    out = np.empty(3, np.float64)
    lib().ora_direct_light_keyed.restype = C.c_uint32
    draws = lib().ora_direct_light_keyed(C.byref(scene.c), C.byref(p), C.c_uint32(domain), C.c_uint32(a), C.c_uint32(b), C.c_uint32(first_draw),
                                         _p(point), _p(normal), _p(local), _p(out))
    return out, draws


def emit_many(scene: Scene, p: OraParams):
    nd, nc = p.num_photons_diffuse, p.num_photons_caustic
    diffuse = np.zeros((nd, 11), np.float64)
    caustic = np.zeros((nc, 11), np.float64)
    dtag = np.zeros((nd, 3), np.int64)
    ctag = np.zeros((nc, 3), np.int64)
    n_d, n_c = C.c_uint32(), C.c_uint32()
    a1, a2, draws, segs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    capped = lib().ora_emit_many(C.byref(scene.c), C.byref(p), _p(diffuse), _p(dtag), C.byref(n_d),
                                 _p(caustic), _p(ctag), C.byref(n_c), C.byref(a1), C.byref(a2),
                                 C.byref(draws), C.byref(segs))
    return dict(diffuse=diffuse[:n_d.value], diffuse_tag=dtag[:n_d.value], caustic=caustic[:n_c.value],
                caustic_tag=ctag[:n_c.value], attempts1=a1.value, attempts2=a2.value, draws=draws.value,
                segments=segs.value, capped=bool(capped))


def emit_attempts(scene: Scene, p: OraParams, phase, a0, a1):
    n = a1 - a0
    flag = np.zeros(n, np.int32)
    rec = np.zeros((n, 11), np.float64)
    lib().ora_emit_attempts(C.byref(scene.c), C.byref(p), C.c_int(phase), C.c_uint64(a0), C.c_uint64(a1), _p(flag), _p(rec))
    return flag, rec


def find_median(n):
    return lib().ora_find_median(C.c_uint32(n))


def kd_build(pos):
    pos = _c(pos, np.float64)
    n = pos.shape[0]
    perm = np.empty(n, np.int32)
    dim = np.empty(n, np.uint8)
    rc = lib().ora_kd_build(_p(pos), C.c_uint32(n), _p(perm), _p(dim))
    if rc:
        raise ValueError("tried to build a k-d tree with no data!")
    return perm, dim


def kd_radius(pos, perm, dim, x, r, cap=4096):
    pos, x = _c(pos, np.float64), _c(x, np.float64)
    out = np.empty(cap, np.int32)
    n = lib().ora_kd_radius(_p(pos), C.c_uint32(pos.shape[0]), _p(perm), _p(dim), _p(x), C.c_double(r), _p(out), C.c_uint32(cap))
    if n > cap:
        return kd_radius(pos, perm, dim, x, r, cap=int(n))
    return out[:n].copy()


def kd_nearest(pos, perm, dim, x):
    """nearestNeighbourAndDistance(x) (kdtree.js:10-62): (input index, distance), or (-1, nan) for [undefined, undefined]"""
    pos, x = _c(pos, np.float64), _c(x, np.float64)
    idx, dist = C.c_int32(0), C.c_double(0.0)
    lib().ora_kd_nearest(_p(pos), C.c_uint32(pos.shape[0]), _p(perm), _p(dim), _p(x), C.byref(idx), C.byref(dist))
    return idx.value, dist.value


def knn_brute(pos, x, k):
    pos, x = _c(pos, np.float64), _c(x, np.float64)
    idx = np.empty(k, np.int32)
    d2 = np.empty(k, np.float64)
    lib().ora_knn_brute(_p(pos), C.c_uint32(pos.shape[0]), _p(x), C.c_uint32(k), _p(idx), _p(d2))
    return idx, d2


class Maps:
    def __init__(self, diffuse, caustic):
        self.diffuse, self.caustic = _c(diffuse, np.float64), _c(caustic, np.float64)
        self.h = lib().ora_maps_create(_p(self.diffuse), C.c_uint32(self.diffuse.shape[0]),
                                       _p(self.caustic), C.c_uint32(self.caustic.shape[0]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_maps_destroy(C.c_void_p(self.h))
            self.h = None


def render_whitted(scene: Scene, p: OraParams, x0, y0, w, h):
    """mode A (ray.js): raw traceRay colours and the putPixel values"""
    raw = np.zeros((h, w, 4), np.float64)
    post = np.zeros((h, w, 4), np.float64)
    rays = C.c_uint64()
    lib().ora_render_whitted(C.byref(scene.c), C.byref(p), C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), _p(raw), _p(post), C.byref(rays))
    return raw, post, rays.value


def render(scene: Scene, p: OraParams, maps: Maps | None, x0, y0, w, h):
    b = max(1, p.super_sample_block)
    pre = np.zeros((-(-h // b), -(-w // b), 4), np.float64)
    post = np.zeros((-(-h // b), -(-w // b), 4), np.float64)
    rays, tests = C.c_uint64(), C.c_uint64()
    lib().ora_render(C.byref(scene.c), C.byref(p), C.c_void_p(maps.h if maps else None), C.c_int(x0), C.c_int(y0),
                     C.c_int(w), C.c_int(h), _p(pre), _p(post), C.byref(rays), C.byref(tests))
    return pre, post, rays.value, tests.value
