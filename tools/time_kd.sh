#!/bin/bash
# photon-map build times and the per-kernel split at 1 M photons; run on the GPU box: tools/time_kd.sh [test-pattern]
R=$PWD; O=$R/gpurun_out/time_kd; mkdir -p $O
if [ -n "$1" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "$1" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
  tail -3 $O/pytest.txt
fi
timeout -k 10 300 python - > $O/kd_time.txt 2>&1 <<'PY'
import time, numpy as np, torch
from raytracer_amd import capi, synth
ctx = capi.Context(0)
for n in (100_000, 1_000_000, 1_400_000, 2_000_000, 4_000_000, 10_000_000):
    rec = torch.from_numpy(synth.box_photons(n, seed=7)).cuda()
    m = ctx.photon_map(rec); del m
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); m = ctx.photon_map(rec); ts.append(time.perf_counter() - t); del m
    print(n, "photon_map build ms:", [round(x * 1e3, 3) for x in ts])
ctx.profile_enable(True)
rec = torch.from_numpy(synth.box_photons(1_000_000, seed=7)).cuda()
ctx.profile_reset(); m = ctx.photon_map(rec)
for k in ("kd_init", "kd_bbox", "kd_keys", "kd_sort", "kd_split", "kd_lds", "map_extract_pos", "map_permute"):
    print(k, ctx.profile_get(k))
PY
cat $O/kd_time.txt
