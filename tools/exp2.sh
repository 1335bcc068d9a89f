# bench (frame leg only) for each named variant library under gpurun_variants/, then for the in-tree build
set -e
for v in "$@"; do
  RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_v$v.json
done
timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_vbase.json
