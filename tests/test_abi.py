"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rt355.h declares,
and the ctypes struct mirrors have the sizes the header implies.  No compute calls."""
import ctypes as C
import os
import re

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(capi.EXPORTS)


def test_struct_layouts():
    assert C.sizeof(capi.SceneDesc) == 8 + 3 * 8 + 8 + 2 * 8 + 16 * 8 + 8 + 32 * 8 + 8
    assert C.sizeof(capi.Params) == 120 and C.sizeof(capi.ListsInfo) == 40
    assert C.sizeof(capi.TraceStats) == 72 and C.sizeof(capi.EmitStats) == 40
    p = capi.default_params()
    assert p.epsilon == 1e-6 and p.max_trace_dist == 20 and p.max_depth == 5          # rayphoton.js:6-9
    assert p.num_photons_diffuse == 20000 and p.num_photons_caustic == 50000          # rayphoton.js:16-17
    assert p.diffuse_radius == 0.40 and p.caustic_radius == 0.15 and p.standard_lamp_area == 100
    assert capi.load_library().rt_find_median(C.c_uint32(10)) == 6                    # kdtree.js:109-124


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    try:
        capi.Context(0)
    except capi.RtError as e:
        assert e.code == -3 and "no CPU path" in str(e)
    else:
        raise AssertionError("Context() must fail without a GPU")
