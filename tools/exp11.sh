# first-pass time with and without per-workgroup culling (experiment)
set -e
RT355_BENCH_SERIAL_FRAMES=1 RT355_NO_CULL=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_nocull.json
RT355_BENCH_SERIAL_FRAMES=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_cull.json
