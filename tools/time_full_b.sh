#!/bin/bash
# times the full mode-B frame leg for each library variant named (default: the in-tree build); run on the GPU box
for v in "${@:-}"; do
  RT355_LIB=$v timeout -k 10 300 python bench.py --legs full_b --no-cpu-baseline --steps 20 > gpurun_out/fb.json 2>/dev/null || exit 1
  echo "variant [$v]"; python tools/show.py gpurun_out/fb.json | grep -A3 "full_mode_b_frame"
done
