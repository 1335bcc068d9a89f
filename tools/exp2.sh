# bench (frame leg only, one frame at a time and two in flight) for each named variant library under gpurun_variants/, then for the in-tree build
set -e
for v in "$@"; do
  RT355_BENCH_SERIAL_FRAMES=1 RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_v${v}_serial.json
  RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_v$v.json
done
RT355_BENCH_SERIAL_FRAMES=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_vbase_serial.json
timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_vbase.json
