#!/bin/bash
R=$PWD; O=$R/gpurun_out/${OUT:-r02f}; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -v --timeout 200 --timeout-method thread -k "kd or knn or radius or gather or lists or emission" > $O/pytest_kd.txt 2>&1 || { tail -40 $O/pytest_kd.txt; exit 1; }
tail -3 $O/pytest_kd.txt
timeout -k 10 300 python - > $O/kd_time.txt 2>&1 <<'PY'
import time, numpy as np, torch
from raytracer_amd import capi, synth
ctx = capi.Context(0)
for n in (1000, 100_000, 1_000_000, 10_000_000):
    rec = torch.from_numpy(synth.box_photons(n, seed=7)).cuda()
    m = ctx.photon_map(rec); del m
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); m = ctx.photon_map(rec); ts.append(time.perf_counter() - t); del m
    print(n, "photon_map build ms:", [round(x * 1e3, 3) for x in ts])
ctx.profile_enable(True)
rec = torch.from_numpy(synth.box_photons(1_000_000, seed=7)).cuda()
ctx.profile_reset(); m = ctx.photon_map(rec)
for k in ("kd_init", "kd_bbox_clear", "kd_bbox", "kd_keys", "kd_sort", "kd_split", "kd_lds", "map_extract_pos", "map_permute"):
    print(k, ctx.profile_get(k))
PY
cat $O/kd_time.txt
