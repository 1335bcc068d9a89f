set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-gather > gpurun_out/exp_full.json 2> gpurun_out/exp_full.err
RT355_BENCH_NO_EVENTS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_noev.json 2> gpurun_out/exp_noev.err
