"""Reader for the array bundles written by oracle/harness/bundle.js (<base>.json + <base>.bin)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Bundle(dict):
    meta: dict


def load_bundle(name: str) -> Bundle:
    base = os.path.join(GOLDEN, name)
    with open(base + ".json") as f:
        man = json.load(f)
    raw = np.fromfile(base + ".bin", dtype=np.uint8)
    out = Bundle()
    for k, a in man["arrays"].items():
        dt = np.dtype("<" + a["dtype"])
        arr = raw[a["offset"]:a["offset"] + a["nbytes"]].view(dt).reshape(a["shape"])
        out[k] = arr
    out.meta = man.get("meta", {})
    return out


def load_scene(preset, width=None, height=None):
    from raytracer_amd.scene import FlatScene
    sc = FlatScene.from_json(os.path.join(GOLDEN, f"scene_preset{preset}_flat.json"))
    if width is not None:
        sc = sc.with_size(width, height)
    return sc


def load_rayjs_scene(preset):
    """flattened ray.js preset (mode A): carries per-shape reflectance/specular"""
    from raytracer_amd.scene import FlatScene
    return FlatScene.from_json(os.path.join(GOLDEN, f"scene_rayjs{preset}_flat.json"))


RAYJS_PARAMS = dict(max_trace_dist=100.0, max_depth=20, sub_sample=1)      # ray.js:8-12


ALL_PRESETS = ["0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "rmt", "giacometti"]      # rayphoton.js:731-1305


def load_packaged_preset(name, width=None, height=None):
    """the reference's scene objects of one preset, flattened (the RNG-built presets as the instance the harness drew)"""
    from raytracer_amd.scene import FlatScene
    sc = FlatScene.from_json(os.path.join(GOLDEN, f"scene_preset{name}_flat.json"))
    return sc.with_size(width, height) if width is not None else sc
