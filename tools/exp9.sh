# GPU tests, then the frame leg of the bench with 1..4 frames in flight (experiment)
set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
for n in 1 2 3 4; do
  RT355_BENCH_IN_FLIGHT=$n RT355_BENCH_NO_EVENTS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-gather --steps 40 > gpurun_out/exp_fl$n.json 2> gpurun_out/exp_fl$n.err
done
