# GPU tests, then chunks per first-pass workgroup (experiment)
set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
for c in 1 2 4 8; do
  RT355_PRIMARY_CHUNKS=$c RT355_BENCH_SERIAL_FRAMES=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 40 > gpurun_out/exp_c${c}_serial.json
  RT355_PRIMARY_CHUNKS=$c timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 40 > gpurun_out/exp_c$c.json
done
