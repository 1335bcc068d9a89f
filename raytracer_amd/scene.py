"""Flat scene description: the array layout that crosses the C ABI (include/rt355.h).

A `FlatScene` holds exactly what `rt_scene_upload` takes.  It is produced either from the JSON
the Node host's `flatten()` writes (raytracer_amd/host/scene.js), or by the synthetic-scene
builders in raytracer_amd/synth.py.  Geometry slot assignment g[0..11] per primitive type
(reference classes: rayphoton.js:353-481):

    plane       p=g0-2 n=g3-5
    sphere      c=g0-2                 r=g6
    hemisphere  c=g0-2 n=g3-5          r=g6 truncMin=g7 truncMax=g8 (flag TRUNC_MAX)
    cylinder    c=g0-2 axis=g3-5       r=g6 halfHeight=g7 (flag HALF_HEIGHT; absent = infinite)
    triangle    A=g0-2 AB=g3-5 AC=g6-8 n=g9-11   (square: same)
    disc        c=g0-2 n=g3-5          r=g6
    annulus     c=g0-2 n=g3-5          outer=g6 inner=g7
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

PLANE, SPHERE, HEMISPHERE, CYLINDER, TRIANGLE, SQUARE, DISC, ANNULUS = range(8)
TYPE_NAMES = ["plane", "sphere", "hemisphere", "cylinder", "triangle", "square", "disc", "annulus"]
FLAG_CONVEX, FLAG_LIGHT, FLAG_TRUNC_MAX, FLAG_HALF_HEIGHT, FLAG_FIRST_ROOT = 1, 2, 4, 8, 16
MAT_AIR, MAT_GLASS, MAT_WATER, MAT_MIRROR, MAT_COPPER, MAT_LINOLEUM, MAT_PLASTER, MAT_SPECTRALON = range(8)
MAT_NONE = 255

NAN = float("nan")
# rayphoton.js:64-100; NaN stands for `undefined`
DEFAULT_MATERIALS = np.array([
    [1, 1, 1, 0, 0, 0, 0, 0],                       # transparent
    [1.0, 1.5, 1.33, NAN, 0.64, NAN, NAN, NAN],     # refrIndex
    [NAN, NAN, NAN, 1, 0.7, 0.6, 0.2, 1],           # reflectance
    [NAN, NAN, NAN, 1, 0.5, 0.2, 0, 0],             # specular
], dtype=np.float64)


def _nan(x):
    return NAN if x is None else float(x)


@dataclass
class FlatScene:
    shape_meta: np.ndarray            # (n, 4) uint8: type, material, flags, proc colour id
    shape_geom: np.ndarray            # (n, 12) float64
    shape_colour: np.ndarray          # (n, 4) float64
    light_shape: np.ndarray           # (nl,) uint32
    light_data: np.ndarray            # (nl, 8) float64: wattage area prob r g b a reserved
    camera: np.ndarray                # (16,) float64: origin3 u3 v3 w3 fovScaleWidth fovScaleHeight
    width: int
    height: int
    materials: np.ndarray = field(default_factory=lambda: DEFAULT_MATERIALS.copy())
    shape_surface: np.ndarray | None = None   # (n, 2) per-shape reflectance, specular (ray.js scenes) or None

    @property
    def n_shapes(self) -> int:
        return int(self.shape_meta.shape[0])

    @property
    def n_lights(self) -> int:
        return int(self.light_shape.shape[0])

    def validate(self) -> None:
        n = self.n_shapes
        assert self.shape_meta.dtype == np.uint8 and self.shape_meta.shape == (n, 4)
        assert self.shape_geom.dtype == np.float64 and self.shape_geom.shape == (n, 12)
        assert self.shape_colour.dtype == np.float64 and self.shape_colour.shape == (n, 4)
        assert self.light_shape.dtype == np.uint32
        assert self.light_data.shape == (self.n_lights, 8)
        assert self.shape_surface is None or self.shape_surface.shape == (n, 2)
        assert self.camera.shape == (16,) and self.materials.shape == (4, 8)
        assert (self.shape_meta[:, 0] < 8).all(), "unknown primitive type"
        for l in self.light_shape:
            assert l < n and self.shape_meta[l, 0] == DISC and self.shape_meta[l, 2] & FLAG_LIGHT

    def with_size(self, width: int, height: int) -> "FlatScene":
        """Same scene on another canvas: fovScaleHeight follows height/width (rayphoton.js:694)."""
        cam = self.camera.copy()
        cam[13] = cam[12] * height / width
        return FlatScene(self.shape_meta, self.shape_geom, self.shape_colour, self.light_shape,
                         self.light_data, cam, width, height, self.materials, self.shape_surface)

    @classmethod
    def from_flat(cls, d: dict) -> "FlatScene":
        shapes = d["shapes"]
        n = len(shapes)
        meta = np.zeros((n, 4), np.uint8)
        geom = np.zeros((n, 12), np.float64)
        colour = np.zeros((n, 4), np.float64)
        for i, s in enumerate(shapes):
            meta[i] = (s["typeId"], s["material"], s["flags"], s.get("proc", 0))
            geom[i] = s["g"]
            colour[i] = [_nan(c) for c in s["colour"]]
        lights = d["lights"]
        light_shape = np.array([l["shape"] for l in lights], np.uint32)
        light_data = np.zeros((len(lights), 8), np.float64)
        for i, l in enumerate(lights):
            light_data[i, :3] = (l["wattage"], l["area"], l["prob"])
            light_data[i, 3:7] = [_nan(c) for c in l["colour"]]
        cam = d["camera"]
        camera = np.zeros(16, np.float64)
        camera[0:3], camera[3:6], camera[6:9], camera[9:12] = cam["origin"], cam["u"], cam["v"], cam["w"]
        camera[12], camera[13] = cam["fovScaleWidth"], cam["fovScaleHeight"]
        m = d.get("materials")
        if m is None:
            materials = DEFAULT_MATERIALS.copy()
        else:
            materials = np.array([[_nan(x) for x in m[k]] for k in
                                  ("transparent", "refrIndex", "reflectance", "specular")], np.float64)
        surface = None
        if shapes and "reflectance" in shapes[0]:
            surface = np.array([[_nan(s_.get("reflectance")), _nan(s_.get("specular"))] for s_ in shapes], np.float64)
        sc = cls(meta, geom, colour, light_shape, light_data, camera, int(d["width"]), int(d["height"]), materials, surface)
        sc.validate()
        return sc

    @classmethod
    def from_json(cls, path: str) -> "FlatScene":
        with open(path) as f:
            return cls.from_flat(json.load(f))
