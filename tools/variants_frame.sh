#!/bin/bash
# full-frame kernel times with each of the given variant libraries (gpurun_variants/librt355_<name>.so)
for v in "$@"; do
  echo "== $v"
  RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 200 python -u tools/full_frame_split.py 2>&1 | grep '"direct_light": 1, "photon_maps": 1' || exit 1
done
