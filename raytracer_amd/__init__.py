"""MI355X-native ray-tracing + photon-mapping core (host-side Python mirror).

The compute path lives in raytracer_amd/csrc (hand-written HIP behind the C ABI of
include/rt355.h); this package is the thin ctypes host used by tests and bench.py.  The
reference-language host (Node + N-API) is raytracer_amd/host.
"""
from .scene import FlatScene  # noqa: F401
